// Shared device/host definitions for libmrgan_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

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
// Layer-noise / z generator: a counter hash + Box-Muller.  Restated bit-for-bit in
// oracle/mrgan_oracle.py (device_normal).  One call yields the four normals of rows 4q..4q+3 at one
// column of one noise site in one sub-step:
//     key = mix(seed, site*256+seg, sub-step)                       (wave-uniform)
//     a   = mix32(key ^ q*0x9E3779B1)                               (per row group)
//     x0  = mix32(a ^ col*0x85EBCA77),  x1 = mix32(rotl16(a) + col*0xC2B2AE3D + 1)
// and the four 16-bit halves of (x0, x1) are the uniforms of two Box-Muller pairs.  mix32 is the
// "lowbias32" integer finaliser; ~5 VALU ops per normal for the bits (Philox4x32-10: ~25), which
// matters because the forward GEMM epilogues draw one normal per output element.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t noise_key(uint64_t seed, uint32_t site_seg, uint32_t step) {
    return mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) ^ mix32(step ^ mix32(site_seg))));
}
__device__ __forceinline__ float u16_01(uint32_t v) { return fmaf((float)v, 1.0f / 65536.0f, 0.5f / 65536.0f); }   // (v+0.5)/2^16, inside (0,1)

// n[0..3]: standard normals for rows 4q..4q+3 at column col
__device__ __forceinline__ void normal4(uint32_t key, uint32_t q, uint32_t col, float n[4]) {
    const uint32_t a = mix32(key ^ (q * 0x9E3779B1u));
    const uint32_t x0 = mix32(a ^ (col * 0x85EBCA77u));
    const uint32_t x1 = mix32(((a << 16) | (a >> 16)) + col * 0xC2B2AE3Du + 1u);
    // r = sqrt(-2 ln u) = sqrt(-2 ln2 * log2 u); v_sin/v_cos take revolutions
    // raw v_sqrt_f32 (1 ulp): the IEEE-exact sqrtf expands to ~15 instructions
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u16_01(x0 & 0xFFFFu)));
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u16_01(x1 & 0xFFFFu)));
    const float t0 = u16_01(x0 >> 16), t1 = u16_01(x1 >> 16);
    n[0] = r0 * __builtin_amdgcn_cosf(t0);
    n[1] = r0 * __builtin_amdgcn_sinf(t0);
    n[2] = r1 * __builtin_amdgcn_cosf(t1);
    n[3] = r1 * __builtin_amdgcn_sinf(t1);
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

}  // namespace mrgan
