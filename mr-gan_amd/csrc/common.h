// Shared device/host definitions for libmrgan_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrgan {

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
// Philox4x32-10 + Box-Muller.  Restated bit-for-bit in oracle/mrgan_oracle.py (device_normal).
// counter = (col, row>>2, site*256+seg, sub-step), key = seed; the four outputs are the normals of
// rows 4q..4q+3 at that column.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                              uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ float u01(uint32_t x) {
    return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f);     // exact in fp32, inside (0,1)
}

// n[0..3]: standard normals for rows 4q..4q+3 at column col
__device__ __forceinline__ void normal4(uint64_t seed, uint32_t site_seg, uint32_t step, uint32_t q,
                                        uint32_t col, float n[4]) {
    uint32_t c0 = col, c1 = q, c2 = site_seg, c3 = step;
    philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
    // r = sqrt(-2 ln u) = sqrt(-2 ln2 * log2 u); v_sin/v_cos take revolutions
    const float r0 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(c0)));
    const float r1 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(c2)));
    const float t0 = u01(c1), t1 = u01(c3);
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
