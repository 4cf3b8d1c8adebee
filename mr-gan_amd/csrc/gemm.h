// GEMM argument block + the epilogues shared by the fp32 and bf16 MFMA kernels.
//
// Every dense layer of the generator / discriminator (mr_gan.py:110-128) and every backward
// contraction implied by adam.get_updates (mr_gan.py:166-167) is one of three products:
//   FWD   Y[M,N]  = act(X[M,K] W[K,N] + b) (+ sigma*noise)          fused bias/act/noise/mask/col-sums
//   DX    dX[M,K] = (dY[M,N] W[K,N]^T) * act'(prev layer)           fused activation-grad + bias-grad sums
//   SLAB  dW[K,N] = X[M,K]^T dY[M,N], split over M into fp32 slabs  summed later inside the Adam kernel
// All use 32x32 MFMA accumulators (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)),
// so the epilogue code below is common to v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16.
#pragma once
#include "common.h"

namespace mrgan {

enum { EPI_FWD = 0, EPI_DX = 1, EPI_SLAB = 2 };

struct Epi {
    int act;                 // FWD: activation; DX: derivative applied (RELU mask / SOFTPLUS from h / LINEAR)
    int n_valid;             // logical number of output columns; columns beyond are forced to zero
    const float* bias;       // FWD
    void* out; long out_bs; int ldo;            // T output [batch][rows][ldo]
    float sigma; uint32_t site; uint32_t seg0;  // FWD: out += sigma * N(0,1) drawn at (site, seg0+batch)
    uint32_t row0;                              // global row offset of this rank inside a segment
    uint64_t seed;
    uint32_t* mask; long mask_bs; int ldm;      // FWD relu: written; DX relu: read. bit (row, col) at word col>>5
    const void* h; long h_bs; int ldh;          // DX softplus: previous-layer output h (T); CS_SUM_XHAT: BN input h1
    int cs_mode; float* cs1; float* cs2; int ldcs;   // per-row-tile column partial sums [batch*tiles_m + tile][ldcs]
    const float* bn_mu; const float* bn_rstd;   // CS_SUM_XHAT
    float* slab; long slab_stride;              // SLAB: fp32 [split][rows][ldo]
    const DevState* st;
    int ablate;              // timing experiments only: 1 no noise, 2 no epilogue, 4 no main loop, 8 linear activation
};

struct GemmArgs {
    int M, N, K;             // output rows / cols, reduction length (per batch)
    int nbatch, splits;      // grid.z = nbatch * splits ; splits > 1 only for SLAB
    int kchunk;              // reduction elements per split (multiple of the kernel's BK)
    int tiles_m;             // ceil(M / BM)
    int seg_stride, seg_rows;  // SLAB: reduction index v is a row of [nseg][seg_stride] with only v % seg_stride < seg_rows valid
    const void* A; long a_bs, a_si, a_sk;   // A(i,k) at A + b*a_bs + i*a_si + k*a_sk
    const void* B; long b_bs, b_sk, b_sj;   // B(k,j) at B + b*b_bs + k*b_sk + j*b_sj
    Epi e;
};

// one wave's share of the block tile: MR x NR accumulators of 32x32.
// STAGED: the block's output tile is first assembled in LDS (`tile`, [BM][bn] of T, the dead staging
// buffers) and then written with coalesced 16-byte stores -- the accumulator layout holds one column
// per lane, so direct stores would be 2-byte pieces at a row stride (store-issue bound).
template <typename T, int EPI, int MR, int NR, int WM, bool STAGED = false>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[MR][NR], const GemmArgs& g, int batch, int split,
                                         int tile_m, int row_blk, int col_blk, int wm, int wn, int lane,
                                         float* lds /* >= 2*WM*bn floats of scratch, disjoint from `tile` */,
                                         int bn /* block tile width */, T* tile = nullptr) {
    const Epi& e = g.e;
    const int lc = lane & 31, lh = lane >> 5;
    const int M = g.M;
    if (e.ablate & 2) { if (acc[0][0][0] == 12345.678f) ((float*)e.out)[0] = 1.f; return; }

    if constexpr (EPI == EPI_SLAB) {
        float* dst = e.slab + (long)(batch * g.splits + split) * e.slab_stride;
#pragma unroll
        for (int mi = 0; mi < MR; ++mi)
#pragma unroll
            for (int ni = 0; ni < NR; ++ni) {
                const int col = col_blk + (wn * NR + ni) * 32 + lc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_blk + (wm * MR + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < M && col < g.N) dst[(long)row * e.ldo + col] = acc[mi][ni][r];
                }
            }
        return;
    } else {
        T* out = (T*)e.out + (long)batch * e.out_bs;
        uint32_t* mask = e.mask ? e.mask + (long)batch * e.mask_bs : nullptr;
        const T* hprev = e.h ? (const T*)e.h + (long)batch * e.h_bs : nullptr;
        const uint32_t step = e.st ? e.st->iter : 0u;
        const uint32_t nkey = noise_key(e.seed, e.site * 256u + e.seg0 + (uint32_t)batch, step);
        float cs1[NR], cs2[NR];
#pragma unroll
        for (int ni = 0; ni < NR; ++ni) { cs1[ni] = 0.f; cs2[ni] = 0.f; }

#pragma unroll
        for (int ni = 0; ni < NR; ++ni) {
            const int col = col_blk + (wn * NR + ni) * 32 + lc;
            const bool colvalid = col < e.n_valid;
            const bool colin = col < g.N;              // N is a multiple of 64, the block tile is 128 wide
            float bias = 0.f, mu = 0.f, rstd = 0.f;
            if constexpr (EPI == EPI_FWD) { if (colvalid && e.bias) bias = e.bias[col]; }
            if constexpr (EPI == EPI_DX) {
                if (e.cs_mode == CS_SUM_XHAT && colvalid) { mu = e.bn_mu[col]; rstd = e.bn_rstd[col]; }
            }
#pragma unroll
            for (int mi = 0; mi < MR; ++mi) {
                const int rbase = row_blk + (wm * MR + mi) * 32 + 4 * lh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r4 = rbase + 8 * q;
                    float nz[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (EPI == EPI_FWD) {
                        if (e.sigma > 0.f && !(e.ablate & 1)) normal4(nkey, (e.row0 + (uint32_t)r4) >> 2, (uint32_t)col, nz);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = r4 + j;
                        const bool rowvalid = row < M;
                        float v = acc[mi][ni][4 * q + j];
                        if constexpr (EPI == EPI_FWD) {
                            v += bias;
                            if (e.act == ACT_RELU) v = fmaxf(v, 0.f);
                            else if (e.act == ACT_SOFTPLUS && !(e.ablate & 8)) v = softplus_f(v);
                            if (!colvalid) v = 0.f;
                            if (e.act == ACT_RELU && mask) {
                                const unsigned long long bal = __ballot(v > 0.f);
                                if (lc == 0 && rowvalid && colin)
                                    mask[(long)row * e.ldm + (col >> 5)] = lh ? (uint32_t)(bal >> 32) : (uint32_t)bal;
                            }
                            if (rowvalid) { cs1[ni] += v; cs2[ni] += v * v; }
                            if (colvalid) v += e.sigma * nz[j];
                        } else {
                            if (e.act == ACT_RELU) {
                                const uint32_t w = (rowvalid && colin) ? mask[(long)row * e.ldm + (col >> 5)] : 0u;
                                v = ((w >> (col & 31)) & 1u) ? v : 0.f;
                            } else if (e.act == ACT_SOFTPLUS) {
                                const float hv = (rowvalid && colin) ? Elem<T>::to_f32(hprev[(long)row * e.ldh + col]) : 0.f;
                                if (!(e.ablate & 8)) v *= -expm1f(-hv);   // softplus'(pre) = sigmoid(pre) = 1 - exp(-h)
                            }
                            if (!colvalid || !rowvalid) v = 0.f;
                            cs1[ni] += v;
                            if (e.cs_mode == CS_SUM_XHAT && rowvalid && colin) {
                                const float h1 = Elem<T>::to_f32(hprev[(long)row * e.ldh + col]);
                                cs2[ni] += v * (h1 - mu) * rstd;
                            }
                        }
                        if constexpr (STAGED) tile[(row - row_blk) * bn + (col - col_blk)] = Elem<T>::from_f32(v);
                        else if (rowvalid && colin) out[(long)row * e.ldo + col] = Elem<T>::from_f32(v);
                    }
                }
            }
        }

        if constexpr (STAGED) {
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
            constexpr int EPV = 16 / (int)sizeof(T);               // elements per 16-byte chunk
            const int chunks_per_row = bn / EPV, bm = WM * MR * 32;
            __syncthreads();
            for (int cidx = threadIdx.x; cidx < bm * chunks_per_row; cidx += blockDim.x) {
                const int r = cidx / chunks_per_row, c = cidx - r * chunks_per_row;
                if (row_blk + r < M && col_blk + c * EPV < g.N)
                    *(u32x4_t*)(out + (long)(row_blk + r) * e.ldo + col_blk + c * EPV) = *(const u32x4_t*)(tile + r * bn + c * EPV);
            }
        }

        if (e.cs_mode != CS_NONE) {
            // lanes l and l^32 hold the same column; then the WM waves stacked along M combine through LDS
            const int wcols = NR * 32;
#pragma unroll
            for (int ni = 0; ni < NR; ++ni) {
                cs1[ni] += __shfl_xor(cs1[ni], 32, 64);
                cs2[ni] += __shfl_xor(cs2[ni], 32, 64);
            }
            __syncthreads();                       // staging LDS is dead from here on
            if (lh == 0) {
#pragma unroll
                for (int ni = 0; ni < NR; ++ni) {
                    const int c = wn * wcols + ni * 32 + lc;
                    lds[(wm * 2 + 0) * bn + c] = cs1[ni];
                    lds[(wm * 2 + 1) * bn + c] = cs2[ni];
                }
            }
            __syncthreads();
            const int t = threadIdx.x;
            if (t < bn) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) { s1 += lds[(w * 2 + 0) * bn + t]; s2 += lds[(w * 2 + 1) * bn + t]; }
                const long prow = (long)(batch * g.tiles_m + tile_m) * e.ldcs;
                if (col_blk + t < e.ldcs) {
                    e.cs1[prow + col_blk + t] = s1;
                    if (e.cs_mode != CS_SUM) e.cs2[prow + col_blk + t] = s2;
                }
            }
        }
    }
}

// host-side launchers (gemm_f32.hip / gemm_bf16.hip)
int launch_gemm_f32(int epi, const GemmArgs& g, hipStream_t s);
int launch_gemm_bf16(int epi, const GemmArgs& g, hipStream_t s);

}  // namespace mrgan
