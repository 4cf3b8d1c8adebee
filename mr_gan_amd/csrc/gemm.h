// GEMM argument block + the epilogues shared by the fp32 and bf16 MFMA kernels.
//
// Every dense layer of the generator / discriminator (mr_gan.py:110-128) and every backward
// contraction implied by adam.get_updates (mr_gan.py:166-167) is one of three products:
//   FWD   Y[M,N]  = act(X[M,K] W[K,N] + b) (+ sigma*noise)          fused bias/act/noise/mask/col-sums
//   DX    dX[M,K] = (dY[M,N] W[K,N]^T) * act'(prev layer)           fused activation-grad + bias-grad sums
//   SLAB  dW[K,N] = X[M,K]^T dY[M,N], split over M into fp32 slabs  summed later inside the Adam kernel
// All use 32x32 MFMA accumulators (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)),
// so the epilogue code below is common to v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16.
//
// ReLU masks are stored in that same lane-native layout: one 16-bit word per (32-row block, column,
// lane half), bit r = accumulator register r.  The forward epilogue builds the word in a register
// (2 VALU ops per element) and stores it once per 32x32 sub-tile, fully coalesced; the backward
// epilogue of the matching DX product loads it back the same way.  relu_mask_bit() decodes it for
// kernels that walk rows and columns.
#pragma once
#include "common.h"

namespace mrgan {

enum { EPI_FWD = 0, EPI_DX = 1, EPI_SLAB = 2 };
// compile-time epilogue variant: activation in bits 0-1, then flags.  VAR_DYN keeps every decision at run time
// (the fp32 parity kernels, where epilogue speed is irrelevant).
enum { VAR_ACT_MASK = 3, VAR_NOISE = 4, VAR_MASK = 8, VAR_DYN = 64 };


struct Epi {
    int act;                 // FWD: activation; DX: derivative applied (RELU mask / SOFTPLUS from h / LINEAR)
    int n_valid;             // logical number of output columns; columns beyond are forced to zero
    const float* bias;       // FWD
    void* out; long out_bs; int ldo;            // T output [batch][rows][ldo]
    float sigma; uint32_t site; uint32_t seg0;  // FWD: out += sigma * N(0,1) drawn at (site, seg0 + batch*seg_step, iter + batch*iter_step)
    int seg_step; uint32_t iter_step;           // (1, 0) for the segments of one sub-step
    uint32_t row0;                              // global row offset of this rank inside a segment
    uint64_t seed;
    uint16_t* mask; long mask_bs; int ldm;      // FWD relu: written; DX relu: read. word ((row>>5)*ldm + col)*2 + half
    const void* h; long h_bs; int ldh;          // DX softplus: previous-layer output h (T); CS_SUM_XHAT: BN input h1
    int cs_mode; float* cs1; float* cs2; int ldcs;   // column partial sums per 64 rows: [batch*tiles_m + row/64][ldcs]
    const float* bn_mu; const float* bn_rstd;   // CS_SUM_XHAT
    float* slab; long slab_stride;              // SLAB: fp32 [split][rows][ldo]
    const DevState* st;
    int ablate;              // timing experiments only: 2 = skip the epilogue, 4 = skip the main loop (one branch each)
    float acc_scale;         // fp8 products: 1 / (scale of A * scale of B), applied to the accumulator (when qa is null)
    // fp8 path (gemm_fp8.hip).  Every fp8 tensor has a slot {amax of the last pass, power-of-two scale, 1 / scale}.
    const Fp8Slot* qa; const Fp8Slot* qb;     // operand slots: accumulator *= qa->inv_scale * qb->inv_scale
    Fp8Slot* qo;                              // output slot: stored byte = fp8(v * qo->scale), e4m3 after a forward product,
                                              // e5m2 after a dX product; max |v| -> qo->amax_bits
    void* q8; long q8_bs; int ldq8;           // fp8 copy of the output tile [batch][rows][ldq8]   (null: none)
    void* q8t; long q8t_bs; int ldq8t;        // transposed fp8 copy [cols][ldq8t], batch b at element offset b * q8t_bs
    int tune_kc_cfg;         // forward / dX tile config forced by mrgan_set_tuning (-1 = measured table)
    int tune_bits;           // TUNE_BIT_* of the handle
};

struct GemmArgs {
    int M, N, K;             // output rows / cols, reduction length (per batch)
    int nbatch, splits;      // grid.z = nbatch * splits ; splits > 1 only for SLAB
    int kchunk;              // reduction elements per split (multiple of the kernel's BK)
    int tiles_m;             // column-sum partial rows per batch = ceil(M / 64)
    int seg_stride, seg_rows;  // SLAB: reduction index v is a row of [nseg][seg_stride] with only v % seg_stride < seg_rows valid
    const void* A; long a_bs, a_si, a_sk;   // A(i,k) at A + b*a_bs + i*a_si + k*a_sk
    const void* B; long b_bs, b_sk, b_sj;   // B(k,j) at B + b*b_bs + k*b_sk + j*b_sj
    Epi e;
};

// bit of element (row, col) in the lane-native mask layout
__device__ __forceinline__ uint32_t relu_mask_bit(const uint16_t* mask, int ldm, int row, int col) {
    const int rr = row & 31;
    const uint32_t w = mask[((long)(row >> 5) * ldm + col) * 2 + ((rr >> 2) & 1)];
    return (w >> ((rr & 3) | ((rr >> 3) << 2))) & 1u;
}

// Values the epilogue needs from global memory, fetched BEFORE the main loop so that their latency hides under
// it (inside the epilogue each would be an exposed ~1 us round trip): the bias of the wave's columns (FWD) and
// the relu-mask words of its 32x32 sub-tiles (DX).
template <int MR, int NR>
struct EpiPrefetch {
    float bias[NR];
    float mu[NR], rstd[NR];      // DX with CS_SUM_XHAT: BatchNorm mean / 1/std of the wave's columns
    uint32_t mbits[MR][NR];
    uint32_t iter;               // DevState::iter for the noise key (FWD with noise)
};

// does this DX launch read e.h per element?  (softplus derivative, or the sum of dy * xhat for BatchNorm backward)
template <int VAR>
__device__ __host__ __forceinline__ bool dx_needs_h(const Epi& e) {
    if constexpr ((VAR & VAR_DYN) != 0) return e.h && (e.act == ACT_SOFTPLUS || e.cs_mode == CS_SUM_XHAT);
    else if constexpr ((VAR & VAR_ACT_MASK) == ACT_SOFTPLUS) return true;
    else if constexpr ((VAR & VAR_ACT_MASK) == ACT_LINEAR) return e.cs_mode == CS_SUM_XHAT;
    else return false;
}

template <typename T, int EPI, int MR, int NR, int VAR>
__device__ __forceinline__ void epilogue_prefetch(EpiPrefetch<MR, NR>& pf, const GemmArgs& g, int batch, int row_blk, int col_blk,
                                                  int wm, int wn, int lane) {
    const Epi& e = g.e;
    const int lc = lane & 31, lh = lane >> 5;
    const int act = (VAR & 64) ? e.act : (VAR & 3);
    pf.iter = 0u;
    if constexpr (EPI == EPI_FWD && ((VAR & VAR_DYN) || (VAR & VAR_NOISE))) { if (e.st) pf.iter = e.st->iter; }
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
        const int col = col_blk + (wn * NR + ni) * 32 + lc;
        pf.bias[ni] = 0.f; pf.mu[ni] = 0.f; pf.rstd[ni] = 0.f;
        if constexpr (EPI == EPI_FWD) { if (col < e.n_valid && e.bias) pf.bias[ni] = e.bias[col]; }
        if constexpr (EPI == EPI_DX) {
            // (inside the column-sum pass these two loads were an exposed round trip per launch: 7.6 k cycles in the stamps of the
            // generator's d(BatchNorm output) product)
            if (e.cs_mode == CS_SUM_XHAT) {
                const bool cok = col < e.n_valid && col < g.N;
                const int cc = cok ? col : 0;
                const float m = e.bn_mu[cc], r = e.bn_rstd[cc];
                pf.mu[ni] = cok ? m : 0.f; pf.rstd[ni] = cok ? r : 0.f;
            }
        }
#pragma unroll
        for (int mi = 0; mi < MR; ++mi) {
            pf.mbits[mi][ni] = 0;
            if constexpr (EPI == EPI_DX) {
                const int rsub = row_blk + (wm * MR + mi) * 32;
                if (act == ACT_RELU && e.mask && col < g.N && rsub < g.M)
                    pf.mbits[mi][ni] = (e.mask + (long)batch * e.mask_bs)[((long)(rsub >> 5) * e.ldm + col) * 2 + lh];
            }
        }
    }
}

// one wave's share of the block tile: MR x NR accumulators of 32x32.
// STAGED: the block's output tile is first assembled in LDS (`tile`, [BM][bn] of T, the dead staging
// buffers) and then written with coalesced 16-byte stores -- the accumulator layout holds one column
// per lane, so direct stores would be 2-byte pieces at a row stride (store-issue bound).
// Q8 >= 0 (fp8 path, with STAGED): instead of the bf16 tile, `tile` receives the block's output as fp8 bytes of format Q8,
// twice: row-major [BM][bn + 16] and transposed [bn][BM + 16] (BM = 32 WM MR).  A lane holds four consecutive rows of one
// column per accumulator quad, i.e. exactly one dword of the transposed image; max |v| goes to e.qo.  Rows >= M are stored as
// zeros.  The caller copies the two images out (gemm_fp8.hip copy_tile).
template <typename T, int EPI, int MR, int NR, int WM, bool STAGED = false, int VAR = VAR_DYN, int Q8 = -1>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[MR][NR], const GemmArgs& g, int batch, int split,
                                         int tile_m, int row_blk, int col_blk, int wm, int wn, int lane,
                                         float* lds /* >= 2*WM*bn floats of scratch, disjoint from `tile` */,
                                         int bn /* block tile width */, T* tile = nullptr,
                                         const EpiPrefetch<MR, NR>* pf = nullptr,
                                         const T* htile = nullptr /* STAGED: LDS copy [BM][bn] of the block's tile of e.h */,
                                         unsigned long long* est = nullptr /* make STAMPS=1: [4] cycle sums + [4] = previous stamp */) {
#ifdef MRGAN_STAMPS
#define EPI_STAMP(i) do { if (est) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); est[i] += n_ - est[4]; est[4] = n_; } } while (0)
    if (est) est[4] = __builtin_amdgcn_s_memtime();
#else
#define EPI_STAMP(i)
#endif
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
        constexpr bool DYN = (VAR & VAR_DYN) != 0;
        const int act = DYN ? e.act : (VAR & VAR_ACT_MASK);
        constexpr bool fast_math = sizeof(T) == 2;
        T* out = (T*)e.out + (long)batch * e.out_bs;
        // specialised variants decide mask / noise at compile time (the launcher guarantees the pointers)
        constexpr bool MASKED = !DYN && (EPI == EPI_DX || (VAR & VAR_MASK));
        uint16_t* mask = nullptr;
        if constexpr (MASKED) mask = e.mask + (long)batch * e.mask_bs;
        else if (DYN && e.mask) mask = e.mask + (long)batch * e.mask_bs;
        const T* hprev = e.h ? (const T*)e.h + (long)batch * e.h_bs : nullptr;
        const bool noisy = EPI == EPI_FWD && (DYN ? e.sigma > 0.f : (VAR & VAR_NOISE) != 0);
        // GaussianNoise of the NEXT layer's input (mr_gan.py:120-126), drawn per 32x32 accumulator tile by one integer
        // MFMA (common.h): the lane l&31 carries the hash of row rsub + (l&31), the result arrives in accumulator layout
        uint32_t rowhash[MR];
        i32x4 hfrag = {0, 0, 0, 0};
        const float sigs = e.sigma * NOISE_SCALE;
        if constexpr (EPI == EPI_FWD) {
            if (noisy) {
                const uint32_t nkey = noise_key(e.seed, e.site * 256u + e.seg0 + (uint32_t)(batch * e.seg_step),
                                                (pf ? pf->iter : (e.st ? e.st->iter : 0u)) + (uint32_t)batch * e.iter_step);
                hfrag = hadamard_frag(lane);
#pragma unroll
                for (int mi = 0; mi < MR; ++mi)
                    rowhash[mi] = noise_rowhash(nkey, e.row0 + (uint32_t)(row_blk + (wm * MR + mi) * 32 + lc));
            }
        }

        float q8_scale = 1.f, q8_amax = 0.f;
        float q8_cs[MR][NR];           // Q8 mode: column sums (CS_SUM) accumulated on the fly -- no `keep` copy of the tile
        if constexpr (Q8 >= 0) {
            q8_scale = e.qo->scale;
#pragma unroll
            for (int mi = 0; mi < MR; ++mi)
#pragma unroll
                for (int ni = 0; ni < NR; ++ni) q8_cs[mi][ni] = 0.f;
        }
        // activated values for the column-sum pass below: ordinary registers, so a launch without column sums never
        // moves them back into the accumulator file
        float keep[MR][NR][16];
#pragma unroll
        for (int ni = 0; ni < NR; ++ni) {
            const int col = col_blk + (wn * NR + ni) * 32 + lc;
            const bool colvalid = col < e.n_valid;
            const bool colin = col < g.N;              // N is a multiple of 64, the block tile may be wider
            float bias = 0.f;
            if constexpr (EPI == EPI_FWD) {
                if (pf) bias = pf->bias[ni];
                else if (colvalid && e.bias) bias = e.bias[col];
            }
            const float sig = (noisy && colvalid) ? sigs : 0.f;
#pragma unroll
            for (int mi = 0; mi < MR; ++mi) {
                const int rsub = row_blk + (wm * MR + mi) * 32;            // first row of this 32x32 sub-tile
                const long mword = ((long)(rsub >> 5) * e.ldm + col) * 2 + lh;
                uint32_t mbits = 0;
                if constexpr (EPI == EPI_DX) {
                    if (pf) mbits = pf->mbits[mi][ni];
                    else if (act == ACT_RELU && colin && rsub < M) mbits = mask[mword];
                }
                i32x16 nzs = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if constexpr (EPI == EPI_FWD) {
                    if (noisy) nzs = noise_block(rowhash[mi], (uint32_t)col >> 5, lane, hfrag);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r4 = rsub + 4 * lh + 8 * q;
                    float o4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * q + j, row = r4 + j;
                        float v = acc[mi][ni][r];
                        float& o = o4[j];
                        if constexpr (EPI == EPI_FWD) {
                            v += bias;
                            if (act == ACT_RELU) {
                                // padding columns have zero weights and zero bias, so they come out exactly 0
                                v = fmaxf(v, 0.f);
                                if (MASKED || mask) mbits |= min(__builtin_bit_cast(uint32_t, v), 1u) << r;    // v >= +0: bit = (v != 0)
                            } else if (act == ACT_SOFTPLUS) {
                                v = colvalid ? (fast_math ? softplus_fast(v) : softplus_f(v)) : 0.f;
                            }
                            o = noisy ? fmaf(sig, (float)nzs[r], v) : v;
                        } else {
                            // rows >= M and padding columns arrive as exact zeros (zero-filled operands / zero weights)
                            if (act == ACT_RELU)        // all-ones / zero from the mask bit, applied to the float's bits
                                v = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & (uint32_t)(-(int)((mbits >> r) & 1u)));
                            else if (act == ACT_SOFTPLUS) {
                                float hv;
                                if (STAGED && htile) hv = (row < M && colin) ? Elem<T>::to_f32(htile[(row - row_blk) * bn + (col - col_blk)]) : 0.f;
                                else hv = (row < M && colin) ? Elem<T>::to_f32(hprev[(long)row * e.ldh + col]) : 0.f;
                                // softplus'(pre) = sigmoid(pre) = 1 - exp(-h)
                                v *= fast_math ? one_minus_exp_neg_fast(hv) : -expm1f(-hv);
                            }
                            o = v;
                        }
                        if constexpr (Q8 < 0) keep[mi][ni][r] = v;
                        if constexpr (!STAGED) { if (row < M && colin) out[(long)row * e.ldo + col] = Elem<T>::from_f32(o); }
                    }
                    if constexpr (Q8 >= 0) {
                        constexpr int BMT = WM * MR * 32;
                        if (rsub + 32 > M) {                    // ragged last row block of the batch: rows >= M become zeros
#pragma unroll
                            for (int j = 0; j < 4; ++j) o4[j] = (r4 + j < M) ? o4[j] : 0.f;
                        }
                        q8_amax = fmaxf(q8_amax, fmaxf(fmaxf(fabsf(o4[0]), fabsf(o4[1])), fmaxf(fabsf(o4[2]), fabsf(o4[3]))));
                        if constexpr (EPI == EPI_DX) q8_cs[mi][ni] += (o4[0] + o4[1]) + (o4[2] + o4[3]);      // bias-gradient sums (o == v)
                        const uint32_t w = fp8_pack4<Q8>(o4[0], o4[1], o4[2], o4[3], q8_scale);
                        unsigned char* q8r = (unsigned char*)tile;
                        unsigned char* q8c = q8r + BMT * (bn + 16);
                        const int rl = r4 - row_blk, cl = col - col_blk;
                        *(uint32_t*)(q8c + cl * (BMT + 16) + rl) = w;
                        // row-major image: 4 x 4 byte transpose inside each lane quad (4 neighbouring columns) by DPP
                        // broadcasts + v_perm, then one dword per lane = row r4 + (lane & 3), columns of the quad
                        const int kq = lane & 3;
                        *(uint32_t*)(q8r + (rl + kq) * (bn + 16) + (cl - kq)) = quad_byte_transpose(w);
                    } else if constexpr (STAGED) {
                        // rows r4 .. r4+3 of one column: convert in pairs (one v_cvt_pk per two values), store the halves
                        T* tp = tile + (r4 - row_blk) * bn + (col - col_blk);
                        if constexpr (sizeof(T) == 2) {
                            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
                            typedef __attribute__((ext_vector_type(2))) float f32x2_t;
                            const bf16x2_t p01 = __builtin_convertvector((f32x2_t){o4[0], o4[1]}, bf16x2_t);
                            const bf16x2_t p23 = __builtin_convertvector((f32x2_t){o4[2], o4[3]}, bf16x2_t);
                            tp[0] = p01[0]; tp[bn] = p01[1]; tp[2 * bn] = p23[0]; tp[3 * bn] = p23[1];
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) tp[j * bn] = Elem<T>::from_f32(o4[j]);
                        }
                    }
                }
                if constexpr (EPI == EPI_FWD) {
                    if (act == ACT_RELU && (MASKED || mask) && colin && rsub < M) mask[mword] = (uint16_t)mbits;
                }
            }
        }

        EPI_STAMP(0);            // per-element math + staging stores
        if constexpr (Q8 >= 0) {
            fp8_amax_commit(e.qo, q8_amax);
        } else if constexpr (STAGED) {
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
            constexpr int EPV = 16 / (int)sizeof(T);               // elements per 16-byte chunk
            const int chunks_per_row = bn / EPV, bm = WM * MR * 32;
            __syncthreads();
            EPI_STAMP(1);        // barrier: the staged tile is complete
            for (int cidx = threadIdx.x; cidx < bm * chunks_per_row; cidx += blockDim.x) {
                const int r = cidx / chunks_per_row, c = cidx - r * chunks_per_row;
                if (e.out && row_blk + r < M && col_blk + c * EPV < g.N)          // (fp8 path: out may be null, only q8 / q8t are kept)
                    *(u32x4_t*)(out + (long)(row_blk + r) * e.ldo + col_blk + c * EPV) = *(const u32x4_t*)(tile + r * bn + c * EPV);
            }
            EPI_STAMP(2);        // copy-out loop (LDS reads + 16-byte stores issued)
        }

        if constexpr (Q8 >= 0) {
            static_assert(MR >= 2 && (MR % 2) == 0, "Q8 epilogue: each wave covers whole 64-row groups");
            if (EPI == EPI_DX && e.cs_mode == CS_SUM) {
                const long prow0 = (long)batch * g.tiles_m + (row_blk >> 6);
#pragma unroll
                for (int hh = 0; hh < MR / 2; ++hh)
#pragma unroll
                    for (int ni = 0; ni < NR; ++ni) {
                        float s1 = q8_cs[2 * hh][ni] + q8_cs[2 * hh + 1][ni];
                        s1 += __shfl_xor(s1, 32, 64);                     // lanes l and l ^ 32 hold the same column
                        const int col = col_blk + (wn * NR + ni) * 32 + lc;
                        const long pr = prow0 + wm * (MR / 2) + hh;
                        if (lh == 0 && col < e.ldcs && row_blk + (wm * (MR / 2) + hh) * 64 < M) e.cs1[pr * e.ldcs + col] = s1;
                    }
            }
        } else if (e.cs_mode != CS_NONE) {
            // column sums per 64-row half of the wave tile (MR >= 2) or of the whole 32-row wave tile
            constexpr int NH = MR >= 2 ? MR / 2 : 1, MH = MR >= 2 ? 2 : 1;
            float cs1[NH][NR], cs2[NH][NR];
#pragma unroll
            for (int hh = 0; hh < NH; ++hh)
#pragma unroll
                for (int ni = 0; ni < NR; ++ni) {
                    const int col = col_blk + (wn * NR + ni) * 32 + lc;
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int m2 = 0; m2 < MH; ++m2) {
                        const int mi = hh * MH + m2;
                        const int rsub = row_blk + (wm * MR + mi) * 32 + 4 * lh;
                        if (e.cs_mode == CS_SUM_XHAT) {
                            const bool cok = col < e.n_valid && col < g.N;
                            const float mu = pf ? pf->mu[ni] : (cok ? e.bn_mu[col] : 0.f), rstd = pf ? pf->rstd[ni] : (cok ? e.bn_rstd[col] : 0.f);
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int row = rsub + (r & 3) + 8 * (r >> 2);
                                const bool ok = row < M;
                                const float v = ok ? keep[mi][ni][r] : 0.f;
                                float hv;
                                if (STAGED && htile) hv = (ok && cok) ? Elem<T>::to_f32(htile[(row - row_blk) * bn + (col - col_blk)]) : mu;
                                else hv = (ok && cok) ? Elem<T>::to_f32(hprev[(long)row * e.ldh + col]) : mu;
                                s1 += v; s2 = fmaf(v * (hv - mu), rstd, s2);
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = (rsub + (r & 3) + 8 * (r >> 2) < M) ? keep[mi][ni][r] : 0.f;
                                s1 += v; s2 = fmaf(v, v, s2);
                            }
                        }
                    }
                    // lanes l and l^32 hold the same column
                    cs1[hh][ni] = s1 + __shfl_xor(s1, 32, 64);
                    cs2[hh][ni] = s2 + __shfl_xor(s2, 32, 64);
                }
            // partial sums have a fixed granularity of 64 rows: partial row = batch*cs_tiles + row/64
            const long prow0 = (long)batch * g.tiles_m + (row_blk >> 6);
            if constexpr (MR >= 2) {
                // each wave covers whole 64-row groups by itself: write its sums straight from registers
                if (lh == 0) {
#pragma unroll
                    for (int hh = 0; hh < NH; ++hh)
#pragma unroll
                        for (int ni = 0; ni < NR; ++ni) {
                            const int col = col_blk + (wn * NR + ni) * 32 + lc;
                            const long pr = prow0 + wm * NH + hh;
                            if (col < e.ldcs && row_blk + (wm * NH + hh) * 64 < M) {
                                e.cs1[pr * e.ldcs + col] = cs1[hh][ni];
                                if (e.cs_mode != CS_SUM) e.cs2[pr * e.ldcs + col] = cs2[hh][ni];
                            }
                        }
                }
            } else {
                // 32-row waves: the WM (= 2) waves stacked along M combine through LDS
                const int wcols = NR * 32;
                __syncthreads();
                if (lh == 0) {
#pragma unroll
                    for (int ni = 0; ni < NR; ++ni) {
                        const int c = wn * wcols + ni * 32 + lc;
                        lds[(wm * 2 + 0) * bn + c] = cs1[0][ni];
                        lds[(wm * 2 + 1) * bn + c] = cs2[0][ni];
                    }
                }
                __syncthreads();
                const int t = threadIdx.x;
                if (t < bn) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) { s1 += lds[(w * 2 + 0) * bn + t]; s2 += lds[(w * 2 + 1) * bn + t]; }
                    if (col_blk + t < e.ldcs) {
                        e.cs1[prow0 * e.ldcs + col_blk + t] = s1;
                        if (e.cs_mode != CS_SUM) e.cs2[prow0 * e.ldcs + col_blk + t] = s2;
                    }
                }
            }
        }
        EPI_STAMP(3);            // column sums
    }
#undef EPI_STAMP
}

enum { TUNE_BIT_KC_PIPE = 1, TUNE_BIT_KS_W8 = 2, TUNE_BIT_NO_KS_GROUP = 4, TUNE_BIT_KS_W4 = 8 };
// up to KS_GROUP_MAX weight-gradient products launched as one grid (gemm_bf16.hip)
constexpr int KS_GROUP_MAX = 6;
// a fold of per-block partial rows that rides along in the grouped launch (the loss head's weight-gradient partials):
// dst[grp][i] = sum of src[p][i] over p = grp, grp + ngroups, ...   ; blocks_x * ngroups extra blocks of 256 threads
struct FoldJob { const float* src; float* dst; long stride; int nsrc, n, ngroups, blocks_x; };
struct KsGroup {
    int n;
    int blk_end[KS_GROUP_MAX];       // exclusive prefix sums of the problems' block counts
    FoldJob fold;                    // fold.blocks_x == 0: none
    GemmArgs g[KS_GROUP_MAX];
};

// host-side launchers (gemm_f32.hip / gemm_bf16.hip)
// kname (optional) receives the name of the kernel instantiation that was launched, spelled as rocprofv3 prints it
int launch_gemm_f32(int epi, const GemmArgs& g, hipStream_t s, const char** kname = nullptr);
int launch_gemm_bf16(int epi, const GemmArgs& g, hipStream_t s, const char** kname = nullptr);
// gemm_fp8.hip: fp8 operands (g.A [M][K] bytes, g.B = Bt [N][K] bytes, K % 128 == 0); formats fixed per product
int launch_gemm_fp8(int epi, const GemmArgs& g, hipStream_t s, const char** kname = nullptr);
int launch_to_fp8(const float* src, long ld_src, unsigned char* dst, long ld_dst, int rows, int cols, int prow, int pcol, float scale,
                  int transpose, hipStream_t s);
// bf16 [nb][rows][ld] -> fp8 copy and / or transposed fp8 copy, scaled by slot->scale; records max |v| in the slot
struct Quant8Args {
    const __bf16* src; long src_bs; int ld; int rows, cols, nb;      // cols: padded width (multiple of 64); rows >= `rows` store zeros
    int prow;                                                         // rows written per batch (multiple of 64, >= rows)
    unsigned char* dst; long dst_bs; int ldd;
    unsigned char* dstt; long dstt_bs; int lddt;
    Fp8Slot* slot; int fmt;
};
int launch_quant8(const Quant8Args& a, hipStream_t s);
int launch_fp8_update_scales(Fp8Slot* slots, int n, hipStream_t s);
int launch_fp8_init_slots(Fp8Slot* slots, int n, const float* targets_dev, hipStream_t s);
int launch_gemm_bf16_dw_group(const GemmArgs* gs, int n, hipStream_t s, const char** kname = nullptr,
                              const FoldJob* fold = nullptr);   // 1 = not applicable

}  // namespace mrgan
