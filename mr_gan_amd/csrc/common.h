// Shared device/host definitions for libmrgan_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <atomic>

namespace mrgan {

// Every kernel of the step is launched through MRGAN_LAUNCH.  Normally that is hipLaunchKernelGGL; while the library's
// profiling pass has armed the timer, the launch goes through hipExtLaunchKernelGGL with a (start, stop) event pair,
// whose timestamps are taken at the kernel's own begin and end on the device -- the same interval rocprofv3's
// kernel trace reports, with no launch gap or event-handling time in it.
struct LaunchTimer { hipEvent_t start, stop; int armed, fired; };
extern thread_local LaunchTimer g_launch_timer;
#define MRGAN_LAUNCH(kern, grid, block, lds, stream, ...)                                                             \
    do {                                                                                                               \
        ::mrgan::LaunchTimer& lt_ = ::mrgan::g_launch_timer;                                                           \
        if (lt_.armed) {                                                                                               \
            lt_.armed = 0; lt_.fired = 1;                                                                              \
            hipExtLaunchKernelGGL(kern, grid, block, lds, stream, lt_.start, lt_.stop, 0, __VA_ARGS__);                \
        } else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                        \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: each launcher keeps one bit per device id, so a process
// that builds handles on several GPUs sets it on each of them (thread-safe: the set is idempotent, the bit is an atomic or)
struct DeviceOnce {
    std::atomic<unsigned long long> done{0};
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return true;
        const unsigned long long bit = 1ull << (dev & 63);
        return (done.load(std::memory_order_relaxed) & bit) == 0;
    }
    void mark() {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) done.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
    }
};

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// activation codes shared by host and device
enum { ACT_LINEAR = 0, ACT_RELU = 1, ACT_SOFTPLUS = 2 };
// column-sum modes of the GEMM epilogues
enum { CS_NONE = 0, CS_SUM = 1, CS_SUM_SQ = 2, CS_SUM_XHAT = 3 };
// noise sites: 0..4 = GaussianNoise before discriminator dense 1..5 (mr_gan.py:118-126), 16 = z
enum { SITE_Z = 16 };

// Per-sub-step device state.  Two slots ping-pong: the first kernel of a sub-step reads slot
// `cur` and writes slot `cur^1`, every other kernel of that sub-step reads slot `cur`.  Kernel
// arguments therefore never change between sub-steps, which is what makes a captured
// (D-step, G-step) hipGraph replayable.
struct DevState {
    uint32_t iter;      // Keras Adam `iterations` (shared by the D and G update lists, mr_gan.py:165-167)
    uint32_t batch;     // batch index inside the epoch streams (stream mode)
    float lr_t;         // lr * sqrt(1-b2^t)/(1-b1^t), t = iter+1
    uint32_t pad;
};

// ---------------------------------------------------------------------------------------
// bf16 <-> f32
// ---------------------------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float to_f32(float v) { return v; }
    static __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Elem<__bf16> {
    static __device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
    static __device__ __forceinline__ __bf16 from_f32(float v) { return (__bf16)v; }   // RNE, NaN-safe (v_cvt_pk_bf16_f32)
};

// ---------------------------------------------------------------------------------------
// Layer-noise / z generator: counter hash -> random bytes -> +-1 Hadamard mix ON THE MATRIX CORE.
// Restated bit-for-bit in oracle/mrgan_oracle.py (device_normal); every step is integer arithmetic, so
// the device and the restatement agree exactly.
//
// The normal at (global row R, column C) of one noise site in one sub-step:
//     key     = mix(seed, site*256+seg, sub-step)                          (wave-uniform)
//     rowhash = mix32(key + R * 0x9E3779B1)
//     w[m]    = mix32(rowhash ^ ((C>>5)*8 + m) * 0x85EBCA77) | 0x01010101,  m = 0..7
//               -> 32 signed ODD bytes a[k] in {-127, -125, .., 127} (symmetric: mean exactly 0), k = 4m + t
//     s       = sum_k a[k] * H[k][C&31],   H[k][j] = (-1)^popcount(k & j)  (32 x 32 Sylvester-Hadamard)
//     n       = s / sqrt(32 * (128^2 - 1) / 3)
// i.e. each normal is a signed sum of 32 independent uniform bytes (Irwin-Hall: excess kurtosis -0.0375,
// support +-9.72 sigma).  The 32 columns of a block share their row's bytes through orthogonal sign patterns:
// exactly uncorrelated, unit variance.
// One v_mfma_i32_32x32x32_i8 turns 4 hash words per lane into the 16 normals a lane needs for a 32x32
// accumulator tile, already in the accumulator's own lane layout: ~4.5 VALU issue slots per normal
// (the Box-Muller generator of round 1 spent ~17) and the matrix pipe is otherwise idle in an epilogue.
// ---------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t noise_key(uint64_t seed, uint32_t site_seg, uint32_t step) {
    return mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) ^ mix32(step ^ mix32(site_seg))));
}
constexpr float NOISE_SCALE = 2.3921528308e-03f;      // 1 / sqrt(32 * (128^2 - 1) / 3)

// B operand of the noise MFMA: lane (j = lane&31, h = lane>>5) holds H[k][j] for k = 16h .. 16h+15 as bytes +1 / -1
__device__ __forceinline__ i32x4 hadamard_frag(int lane) {
    const uint32_t j = (uint32_t)lane & 31u, h = (uint32_t)lane >> 5;
    i32x4 f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        uint32_t w = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t k = 16u * h + 4u * m + t;
            w |= ((__builtin_popcount(k & j) & 1) ? 0xFFu : 0x01u) << (8 * t);
        }
        f[m] = (int)w;
    }
    return f;
}
__device__ __forceinline__ uint32_t noise_rowhash(uint32_t key, uint32_t row) { return mix32(key + row * 0x9E3779B1u); }
// raw sums s of the 32 x 32 block (rows: the 32 rows whose hashes the lanes l&31 carry; columns 32*cblk ..) in the MFMA
// accumulator layout: register r of lane l = (row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31)
__device__ __forceinline__ i32x16 noise_block(uint32_t rowhash, uint32_t cblk, int lane, const i32x4 hfrag) {
    const uint32_t m0 = cblk * 8u + 4u * ((uint32_t)lane >> 5);
    i32x4 a;
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = (int)(mix32(rowhash ^ ((m0 + (uint32_t)m) * 0x85EBCA77u)) | 0x01010101u);
    const i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, hfrag, z, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------
// numerics helpers (fp32)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_f(float x) {
    // log1p(exp(x)) evaluated as max(x,0) + log1p(exp(-|x|))
    return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x)));
}
// bf16 mode: hardware exp2/log2 (absolute error ~1e-7, far below bf16 resolution)
__device__ __forceinline__ float softplus_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));          // exp(-|x|)
    return fmaxf(x, 0.f) + 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + e);     // + ln(1 + e)
}
// 1 - exp(-h)
__device__ __forceinline__ float one_minus_exp_neg_fast(float h) {
    return 1.0f - __builtin_amdgcn_exp2f(-1.4426950408889634f * h);
}
__device__ __forceinline__ float sigmoid_f(float x) {
    return 1.0f / (1.0f + __expf(-x));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline long round_up(long a, long b) { return (a + b - 1) / b * b; }

// one fp8 tensor's scaling state.  Delayed scaling: a pass stores with `scale` (from the amax of the previous pass) and
// records its own max |v| (before scaling); fp8_update_scales_kernel (gemm_fp8.hip) turns that into the next pass's scale.
struct Fp8Slot { uint32_t amax_bits; float scale, inv_scale; float target; };
// ---- fp8 (OCP e4m3 / e5m2) packing: four floats -> four bytes of v * qs, round to nearest even, saturating ----
enum { FP8_E4M3 = 0, FP8_E5M2 = 1 };
template <int FMT>
__device__ __forceinline__ uint32_t fp8_pack4(float a, float b, float c, float d, float qs) {
    constexpr float LIM = FMT == FP8_E5M2 ? 57344.f : 448.f;
    a = fminf(fmaxf(a * qs, -LIM), LIM); b = fminf(fmaxf(b * qs, -LIM), LIM);
    c = fminf(fmaxf(c * qs, -LIM), LIM); d = fminf(fmaxf(d * qs, -LIM), LIM);
    int w;
    if constexpr (FMT == FP8_E5M2) { w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false); w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true); }
    else { w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false); w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true); }
    return (uint32_t)w;
}
// 4 x 4 byte transpose inside each lane quad: lane k of the quad gets byte k of the four lanes' words (DPP broadcasts + v_perm)
__device__ __forceinline__ uint32_t quad_byte_transpose(uint32_t w) {
    const uint32_t kq = threadIdx.x & 3;
    const uint32_t q0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x00, 0xF, 0xF, true);
    const uint32_t q1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x55, 0xF, 0xF, true);
    const uint32_t q2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xAA, 0xF, 0xF, true);
    const uint32_t q3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xFF, 0xF, 0xF, true);
    const uint32_t sel = kq * 0x01010101u + 0x04000400u;                 // bytes (k, 4 + k, k, 4 + k)
    const uint32_t lo = __builtin_amdgcn_perm(q1, q0, sel), hi = __builtin_amdgcn_perm(q3, q2, sel);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}
// wave-wide max of a non-negative value -> the slot's amax (float bits compare as integers)
__device__ __forceinline__ void fp8_amax_commit(Fp8Slot* slot, float amax) {
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    // thousands of waves target one address, and same-address atomics serialise in L2 (measured: +0.3 ms per launch at
    // 10 k waves): a wave whose maximum is not above the value it can already see skips the atomic (a stale read only costs
    // an extra atomic)
    if ((threadIdx.x & 63) == 0 && amax > 0.f) {
        const uint32_t bits = __float_as_uint(amax);
        if (bits > *(volatile uint32_t*)&slot->amax_bits) atomicMax(&slot->amax_bits, bits);
    }
}

}  // namespace mrgan
