// Non-GEMM kernels of the mr_gan training path: HBM-bound elementwise / reduction work.
// Access pattern everywhere: 16-byte loads/stores per lane along the contiguous (feature) dimension,
// 256-thread blocks laid out as 32 column-groups x 8 row-lanes, >= 128 blocks per launch.
#include "aux_kernels.h"

namespace mrgan {
namespace {

// 8 consecutive elements of T <-> 8 floats
template <typename T> __device__ __forceinline__ void load8(const T* p, float v[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float v[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <> __device__ __forceinline__ void load8<__bf16>(const __bf16* p, float v[8]) {
    const bf16x8 a = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float v[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float v[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *(f32x4*)p = a; *(f32x4*)(p + 4) = b;
}
template <> __device__ __forceinline__ void store8<__bf16>(__bf16* p, const float v[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)v[i];
    *(bf16x8*)p = a;
}

// =========================================================================================
// stage: gather rows (optionally by index) of the resident fp32 matrix, add GaussianNoise(sigma)
// (mr_gan.py:118), convert to T, zero the padding columns.  Also draws z when asked to.
// wave <-> 32 rows x 128 columns.  The noise generator (common.h) delivers a 32x32 block in the MFMA accumulator
// layout (lane = column, 16 rows in registers); the integer sums (|s| <= 4064) go through a per-wave int16 LDS image so
// that global loads and stores move 8 consecutive columns per lane (2 x 16 B in, 16 B out).
// =========================================================================================
template <typename T>
__global__ __launch_bounds__(256) void stage_kernel(const StageArgs a) {
    __shared__ __attribute__((aligned(16))) short nlds[4][32][128 + 8];          // +8: rows 272 B apart (bank spread for the 2-byte writes)
    const DevState st = *a.cur;
    const StageSeg& sg = a.s[blockIdx.z];
    const int lane = threadIdx.x & 63, lc = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rbase = (blockIdx.y * 4 + wave) * 32, cbase = (int)blockIdx.x * 128;
    if (rbase >= sg.rows || cbase >= sg.cols_pad) return;          // wave-uniform; no block-level barrier below
    const long o = sg.stream ? (long)st.batch * sg.rows : 0;
    T* out = (T*)sg.out;
    const bool noisy = sg.gen || sg.sigma > 0.f;
    const float sigs = (sg.gen ? 1.0f : sg.sigma) * NOISE_SCALE;
    const int cg = (lane & 15) * 8, rl = lane >> 4;                // lane <-> (8 columns, every 4th row)
    const int col = cbase + cg;
    const bool vec_ok = !sg.gen && (sg.ld & 3) == 0 && ((uintptr_t)sg.src & 15) == 0 && col + 7 < sg.cols;
    // source rows first, then every row's loads in flight together (a dependent index -> row chain per iteration is
    // latency-bound: this kernel is 40 MB of traffic and must not take longer than a GEMM)
    long srow[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = rbase + it * 4 + rl;
        srow[it] = (!sg.gen && row < sg.rows) ? (sg.idx ? (long)sg.idx[o + row] : (o + row)) : 0;
    }
    float v[8][8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[it][c] = 0.f;
        const int row = rbase + it * 4 + rl;
        if (!sg.gen && row < sg.rows) {
            const float* src = sg.src + srow[it] * sg.ld + col;
            if (vec_ok) load8<float>(src, v[it]);
            else {
#pragma unroll
                for (int c = 0; c < 8; ++c) if (col + c < sg.cols) v[it][c] = src[c];
            }
        }
    }
    // the noise of the wave's 32 x 128 block is generated while the row loads above are in flight (it was generated first in
    // round 2, with nothing in flight)
    if (noisy) {
        const uint32_t rowhash = noise_rowhash(noise_key(a.seed, sg.site * 256u + sg.seg, st.iter + sg.iter_off), a.row0 + (uint32_t)(rbase + lc));
        const i32x4 hfrag = hadamard_frag(lane);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int c0 = cbase + cb * 32;
            if (c0 >= sg.cols) break;                              // wave-uniform; columns beyond are never read back
            const i32x16 nz = noise_block(rowhash, (uint32_t)c0 >> 5, lane, hfrag);
#pragma unroll
            for (int r = 0; r < 16; ++r) nlds[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][cb * 32 + lc] = (short)nz[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                           // same wave, in-order LDS: the reads below see the writes
    }
    if (col >= sg.cols_pad) return;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int rr = it * 4 + rl, row = rbase + rr;
        if (row >= sg.rows) break;
        if (noisy && col < sg.cols) {
            const s16x8 nz = *(const s16x8*)&nlds[wave][rr][cg];
#pragma unroll
            for (int c = 0; c < 8; ++c) if (col + c < sg.cols) v[it][c] = fmaf(sigs, (float)nz[c], v[it][c]);
        }
        store8<T>(out + (long)row * sg.ldo + col, v[it]);
    }
}

// =========================================================================================
// Column-statistic kernels (BatchNorm forward / backward, feature matching).
// Block = CB (64) columns x RB (64) rows, 256 threads.  Prologue: the per-64-row partial sums of the block's
// columns are folded by 16 "partial lanes" (thread = 4 columns x every 16th partial row, 16-byte loads) and
// combined through LDS; body: thread = 8 columns (16 B of bf16) x every 32nd row.  ld/64 x rows/64 blocks keep
// >= 256 blocks in flight for the shapes of this model.
// =========================================================================================
constexpr int RB = 64, CB = 64;

__device__ __forceinline__ void fold_partials(const float* part, int npart, int ldcs, int col0, int ncols, float (*scr)[CB], float* out) {
    const int t = threadIdx.x, cq = (t & 15) * 4, pl = t >> 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (col0 + cq < ncols) {
        int p = pl;
        for (; p + 16 < npart; p += 32) {
            s0 += *(const f32x4*)(part + (long)p * ldcs + col0 + cq);
            s1 += *(const f32x4*)(part + (long)(p + 16) * ldcs + col0 + cq);
        }
        for (; p < npart; p += 16) s0 += *(const f32x4*)(part + (long)p * ldcs + col0 + cq);
    }
    s0 += s1;
    *(f32x4*)(&scr[pl][cq]) = s0;
    __syncthreads();
    if (t < CB) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += scr[k][t];
        out[t] = s;
    }
    __syncthreads();
}

// BatchNorm forward with batch statistics (biased variance, eps inside the sqrt; mr_gan.py:112)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyArgs a) {
    __shared__ float sc[CB], sh[CB];
    __shared__ float scr[16][CB], f1[CB], f2[CB];
    const int t = threadIdx.x, col0 = blockIdx.x * CB, seg = blockIdx.z;
    fold_partials(a.cs1 + (long)seg * a.cs_seg_stride, a.npart, a.ldcs, col0, a.ld, scr, f1);
    fold_partials(a.cs2 + (long)seg * a.cs_seg_stride, a.npart, a.ldcs, col0, a.ld, scr, f2);
    if (t < CB) {
        const int col = col0 + t;
        float scale = 0.f, shift = 0.f;
        if (col < a.ld) {
            const float mean = f1[t] / a.count;
            const float var = fmaxf(f2[t] / a.count - mean * mean, 0.f);
            const float rstd = 1.0f / sqrtf(var + a.eps);
            if (col < a.cols) { scale = a.gamma[col] * rstd; shift = a.beta[col] - mean * scale; }
            if (blockIdx.y == 0) { a.mu[(long)seg * a.ld + col] = mean; a.rstd[(long)seg * a.ld + col] = rstd; }
        }
        sc[t] = scale; sh[t] = shift;
    }
    __syncthreads();
    const int cg = t & 7, rl = t >> 3, c0 = col0 + cg * 8;
    if (c0 >= a.ld) return;
    const T* h = (const T*)a.h + (long)seg * a.seg_rows * a.ld;
    T* out = (T*)a.out + (long)seg * a.seg_rows * a.ld;
    const int r1 = min(a.rows, (int)(blockIdx.y + 1) * RB);
    for (int r = blockIdx.y * RB + rl; r < r1; r += 32) {
        float v[8];
        load8<T>(h + (long)r * a.ld + c0, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * sc[cg * 8 + i] + sh[cg * 8 + i];
        store8<T>(out + (long)r * a.ld + c0, v);
    }
}

// BatchNorm backward fused with the softplus derivative of the dense layer in front of it
// (h = softplus(pre)  =>  sigmoid(pre) = 1 - exp(-h)); emits the bias-gradient partial sums.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_kernel(const BnBwdArgs a) {
    __shared__ float cA[CB], cB[CB], cM[CB], cR[CB], cG[CB];   // gamma*rstd/count ; dbeta ; mu ; rstd ; dgamma
    __shared__ float red[32][CB];
    const int t = threadIdx.x, col0 = blockIdx.x * CB;
    fold_partials(a.cs1, a.npart, a.ldcs, col0, a.ld, red, cB);
    fold_partials(a.cs2, a.npart, a.ldcs, col0, a.ld, red, cG);
    if (t < CB) {
        const int col = col0 + t;
        float g = 0.f, mu = 0.f, rs = 0.f;
        if (col < a.cols) { g = a.gamma[col]; mu = a.mu[col]; rs = a.rstd[col]; }
        cA[t] = g * rs / a.count; cM[t] = mu; cR[t] = rs;
    }
    __syncthreads();
    const int cg = t & 7, rl = t >> 3, c0 = col0 + cg * 8;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    if (c0 < a.ld) {
        const T* dy = (const T*)a.dy;
        const T* h = (const T*)a.h;
        T* dpre = (T*)a.dpre;
        const int r1 = min(a.rows, (int)(blockIdx.y + 1) * RB);
        for (int r = blockIdx.y * RB + rl; r < r1; r += 32) {
            float hv[8], d[8], o[8];
            load8<T>(h + (long)r * a.ld + c0, hv);
            load8<T>(dy + (long)r * a.ld + c0, d);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = cg * 8 + i;
                const float xh = (hv[i] - cM[c]) * cR[c];
                const float dh = cA[c] * (a.count * d[i] - cB[c] - xh * cG[c]);
                o[i] = dh * (-expm1f(-hv[i]));
                acc[i] += o[i];
            }
            store8<T>(dpre + (long)r * a.ld + c0, o);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[rl][cg * 8 + i] = acc[i];
    __syncthreads();
    if (t < CB && col0 + t < a.ld) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += red[k][t];
        a.db_part[(long)blockIdx.y * a.ld + col0 + t] = s;
    }
}

// =========================================================================================
// loss head: logits = f W6 + b6 ; labeled / unlabeled / fake losses of mr_gan.py:146-149 ; train error
// :161 ; closed-form dlogits ; dW6, db6 ; and dL/d(pre5) = (dlogits W6^T) * [f > 0].
// One block = HEAD_ROWS rows of one segment.  LDS: f tile as fp32 [HR][feat+8], W6 [feat][8], dlogits [HR][8].
// Per-block partial gradients go to part[blk][...]; reduce_partials_kernel folds them to <= 8 slabs.
// =========================================================================================
constexpr int HR = HEAD_ROWS;
// Q8 (fp8 mode): dpre leaves as e5m2 copies (row-major + transposed) packed from the fp32 values; a separate instantiation,
// so the bf16 / fp32 kernels keep their rolled row loop and register count
template <typename T, bool Q8 = false>
__global__ __launch_bounds__(256) void head_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float hl[];
    // the feature dimension is walked in chunks of CH <= 256 columns (one chunk for the reference's 250-wide layer)
    const int CH = min(a.feat, HEAD_CHUNK), nch = (a.feat + CH - 1) / CH;     // a ragged last chunk is zero-filled
    const int LDF = CH + 8;
    float* f_lds = hl;                         // [HR][LDF]   current chunk of f
    float* w_lds = f_lds + HR * LDF;           // [CH][KMAX]  matching rows of W6
    float* dl_lds = w_lds + CH * KMAX;         // [HR][KMAX]
    float* red = dl_lds + HR * KMAX;           // [4 waves][4]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int seg = blockIdx.y, kind = a.seg_kind[seg];
    const int row_blk = blockIdx.x * HR;
    const int blk = seg * gridDim.x + blockIdx.x;
    const T* f = (const T*)a.f + (long)seg * a.f_bs;

    const int cpr = CH / 8;                    // 8-element chunks per row
    auto load_chunk = [&](int c0) {
        for (int ci = t; ci < HR * cpr; ci += 256) {
            const int r = ci / cpr, c = (ci - r * cpr) * 8;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (row_blk + r < a.rows && c0 + c < a.feat) load8<T>(f + (long)(row_blk + r) * a.ldf + c0 + c, v);
            *(f32x4*)(f_lds + r * LDF + c) = (f32x4){v[0], v[1], v[2], v[3]};
            *(f32x4*)(f_lds + r * LDF + c + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        }
        for (int k = t; k < CH; k += 256) {
            f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
            if (c0 + k < a.feat_valid) { w0 = *(const f32x4*)(a.w + (long)(c0 + k) * a.ldw); w1 = *(const f32x4*)(a.w + (long)(c0 + k) * a.ldw + 4); }
#pragma unroll
            for (int c = 0; c < 4; ++c) { if (c >= a.classes) w0[c] = 0.f; if (c + 4 >= a.classes) w1[c] = 0.f; }
            *(f32x4*)(w_lds + k * KMAX) = w0; *(f32x4*)(w_lds + k * KMAX + 4) = w1;
        }
    };

    // ---- logits: LPR lanes per row, each over an interleaved slice of the features ----
    constexpr int LPR = 256 / HR;
    const int r = t / LPR, part = t % LPR;
    float l[KMAX];
#pragma unroll
    for (int c = 0; c < KMAX; ++c) l[c] = 0.f;
    for (int ch = 0; ch < nch; ++ch) {
        if (ch) __syncthreads();                   // every thread is done with the previous chunk
        load_chunk(ch * CH);
        __syncthreads();
        for (int kk = 0; kk < CH / LPR; ++kk) {
            const int k = kk * LPR + part;
            const float fv = f_lds[r * LDF + k];
            const f32x4 w0 = *(const f32x4*)(w_lds + k * KMAX), w1 = *(const f32x4*)(w_lds + k * KMAX + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) { l[c] = fmaf(fv, w0[c], l[c]); l[4 + c] = fmaf(fv, w1[c], l[4 + c]); }
        }
    }
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) l[c] += __shfl_xor(l[c], m, 64);
    }
    const int row = row_blk + r;
    const bool rowvalid = row < a.rows;
    float mx = -3.0e38f;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        if (c < a.classes) { l[c] += a.b[c]; mx = fmaxf(mx, l[c]); }
    }
    int am = 0;
    float se = 0.f, p[KMAX];
#pragma unroll
    for (int c = KMAX - 1; c >= 0; --c) {
        p[c] = (c < a.classes) ? expf(l[c] - mx) : 0.f;
        se += p[c];
        if (c < a.classes && l[c] == mx) am = c;          // ties -> first index (theano argmax)
    }
    const float lse = mx + logf(se);
    const float inv_se = 1.0f / se;
    float loss0 = 0.f, loss1 = 0.f, err = 0.f;
    float dl[KMAX];
#pragma unroll
    for (int c = 0; c < KMAX; ++c) dl[c] = 0.f;
    if (rowvalid) {
        if (kind == HEAD_MSE) {
            // Keras 'mse' on one-hot targets (mr_nn.py:99, :112): mean over the classes, then over the batch
            const long lo = a.labels_stream ? (long)a.st->batch * a.rows : 0;
            const int y = a.labels[lo + row];
            err = (y >= 0 && am != y) ? 1.f : 0.f;
            const float invc = 1.0f / (float)a.classes;
#pragma unroll
            for (int c = 0; c < KMAX; ++c) {
                if (c < a.classes && y >= 0) {            // label -1: padding row of a short last batch, no contribution
                    const float d = l[c] - (c == y ? 1.f : 0.f);
                    loss0 = fmaf(d * d, invc, loss0);
                    dl[c] = 2.0f * d * invc * a.inv_count;
                }
            }
        } else if (kind == HEAD_LAB || kind == HEAD_EVAL) {
            const long lo = a.labels_stream ? (long)a.st->batch * a.rows : 0;
            const int y = a.labels[lo + row];
            err = (am != y) ? 1.f : 0.f;
            if (kind == HEAD_LAB) {
                float ly = 0.f;
#pragma unroll
                for (int c = 0; c < KMAX; ++c) {
                    if (c == y) ly = l[c];
                    dl[c] = (p[c] * inv_se - (c == y ? 1.f : 0.f)) * a.inv_count;
                }
                loss0 = lse - ly;
            }
        } else if (kind != HEAD_LOGITS) {
            const float sg = sigmoid_f(lse), sp = softplus_f(lse);
            const float k = 0.5f * a.inv_count * a.unl_weight * (kind == HEAD_UNL ? (sg - 1.0f) : sg);
            loss1 = (kind == HEAD_UNL) ? 0.5f * (sp - lse) : 0.5f * sp;
#pragma unroll
            for (int c = 0; c < KMAX; ++c) dl[c] = k * p[c] * inv_se;
        }
    }
    if (part == 0) {
        *(f32x4*)(dl_lds + r * KMAX) = (f32x4){dl[0], dl[1], dl[2], dl[3]};
        *(f32x4*)(dl_lds + r * KMAX + 4) = (f32x4){dl[4], dl[5], dl[6], dl[7]};
        if (a.logits && rowvalid) {
            float* lp = a.logits + (long)seg * a.logits_bs + (long)row * KMAX;
#pragma unroll
            for (int c = 0; c < KMAX; ++c) lp[c] = (c < a.classes) ? l[c] : 0.f;
        }
    } else { loss0 = 0.f; loss1 = 0.f; err = 0.f; }
    loss0 = wave_sum(loss0); loss1 = wave_sum(loss1); err = wave_sum(err);
    if (lane == 0) { red[wave * 4 + 0] = loss0; red[wave * 4 + 1] = loss1; red[wave * 4 + 2] = err; }
    __syncthreads();
    if (t == 0) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int w = 0; w < 4; ++w) { s0 += red[w * 4 + 0]; s1 += red[w * 4 + 1]; s2 += red[w * 4 + 2]; }
        if (kind == HEAD_EVAL) { if (a.err_count) atomicAdd(a.err_count, (int)(s2 + 0.5f)); }
        else if (kind != HEAD_LOGITS) {
            a.loss_part[blk * 4 + 0] = s0; a.loss_part[blk * 4 + 1] = s1;
            a.loss_part[blk * 4 + 2] = s2; a.loss_part[blk * 4 + 3] = 0.f;
        }
    }
    if (kind == HEAD_EVAL || kind == HEAD_LOGITS) return;

    // ---- backward of the last dense: thread <-> feature column j ----
    float* part_row = a.part + (long)blk * a.part_stride;
    if (t < KMAX) {
        float s = 0.f;
        for (int rr = 0; rr < HR; ++rr) s += dl_lds[rr * KMAX + t];
        part_row[a.off_db + t] = s;
    }
    T* dpre = a.dpre ? (T*)a.dpre + (long)seg * a.dpre_bs : nullptr;
    // fp8 mode: the e5m2 copies of dpre that the dX / dW products read (one column's HR rows = HR contiguous bytes of the
    // transposed copy); packed from the fp32 value
    const float q8s = a.q8_slot ? a.q8_slot->scale : 1.f;
    float q8_amax = 0.f;
    unsigned char* q8 = a.q8 ? a.q8 + (long)seg * a.q8_bs : nullptr;
    unsigned char* q8t = a.q8t ? a.q8t + (long)seg * a.q8t_bs : nullptr;
    // the last chunk is still in LDS: walk the chunks backwards and reload only the others
    for (int ch = nch - 1; ch >= 0; --ch) {
        const int c0 = ch * CH;
        if (ch != nch - 1) { __syncthreads(); load_chunk(c0); __syncthreads(); }
        for (int j = t; j < CH && c0 + j < a.feat; j += 256) {
            float wj[KMAX], dw[KMAX];
#pragma unroll
            for (int c = 0; c < KMAX; ++c) { wj[c] = w_lds[j * KMAX + c]; dw[c] = 0.f; }
            float dbf = 0.f;
            auto row_step = [&](int rr) -> float {
                const float fv = f_lds[rr * LDF + j];
                const f32x4 d0 = *(const f32x4*)(dl_lds + rr * KMAX), d1 = *(const f32x4*)(dl_lds + rr * KMAX + 4);
                float dfe = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    dfe = fmaf(d0[c], wj[c], dfe); dfe = fmaf(d1[c], wj[4 + c], dfe);
                    dw[c] = fmaf(fv, d0[c], dw[c]); dw[4 + c] = fmaf(fv, d1[c], dw[4 + c]);
                }
                const float dp = (fv > 0.f) ? dfe : 0.f;
                if (dpre && row_blk + rr < a.rows) dpre[(long)(row_blk + rr) * a.ldd + c0 + j] = Elem<T>::from_f32(dp);
                dbf += dp;
                return dp;
            };
            if constexpr (Q8) {
                // 16 rows at a time: four dwords = 16 contiguous bytes of the transposed copy; the row-major copy trades bytes
                // inside the lane quad (4 neighbouring columns; CH and the column stride are multiples of 4) and stores one
                // dword per four rows.  Rows >= a.rows have dl == 0, hence dp == 0: zero bytes.
                typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
                const int kq = t & 3;
#pragma unroll 1
                for (int k16 = 0; k16 < HR / 16; ++k16) {
                    uint32_t qw[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float q4[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) { q4[i] = row_step(16 * k16 + 4 * k + i); q8_amax = fmaxf(q8_amax, fabsf(q4[i])); }
                        qw[k] = fp8_pack4<FP8_E5M2>(q4[0], q4[1], q4[2], q4[3], q8s);
                        __builtin_amdgcn_sched_barrier(0);      // keep the LDS reads of the next four rows from being hoisted (registers)
                    }
                    if (q8t) *(u32x4_t*)(q8t + (long)(c0 + j) * a.ldq8t + row_blk + 16 * k16) = (u32x4_t){qw[0], qw[1], qw[2], qw[3]};
                    if (q8) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t w = quad_byte_transpose(qw[k]);
                            const int row = row_blk + 16 * k16 + 4 * k + kq;
                            if (row < a.rows) *(uint32_t*)(q8 + (long)row * a.ldq8 + c0 + j - kq) = w;
                        }
                    }
                }
            } else {
                for (int rr = 0; rr < HR; ++rr) row_step(rr);
            }
            *(f32x4*)(part_row + (long)(c0 + j) * KMAX) = (f32x4){dw[0], dw[1], dw[2], dw[3]};
            *(f32x4*)(part_row + (long)(c0 + j) * KMAX + 4) = (f32x4){dw[4], dw[5], dw[6], dw[7]};
            part_row[a.off_dbf + c0 + j] = dbf;
        }
    }
    if constexpr (Q8) fp8_amax_commit(a.q8_slot, q8_amax);
}

// dst[g][i] = sum of src[p][i] over the partial rows p of group g (p = g, g + ngroups, ...)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* src, int nsrc, long stride, int n, int ngroups, float* dst) {
    const int i = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = g;
    for (; p + 3 * ngroups < nsrc; p += 4 * ngroups) {
        s0 += src[(long)p * stride + i]; s1 += src[(long)(p + ngroups) * stride + i];
        s2 += src[(long)(p + 2 * ngroups) * stride + i]; s3 += src[(long)(p + 3 * ngroups) * stride + i];
    }
    for (; p < nsrc; p += ngroups) s0 += src[(long)p * stride + i];
    dst[(long)g * stride + i] = (s0 + s1) + (s2 + s3);
}

// =========================================================================================
// feature matching: loss = mean_j (mean_b f_fake - mean_b f_real)^2 ; dL/df_fake broadcast over rows,
// masked by the feature layer's ReLU.
// =========================================================================================
template <typename T>
__global__ __launch_bounds__(256) void fm_kernel(const FmArgs a) {
    __shared__ float gj_lds[CB];
    __shared__ float scr[16][CB], sf[CB], sr[CB];
    const int t = threadIdx.x, col0 = blockIdx.x * CB;
    const float* cs_real = a.cs + (long)a.npart_fake * a.ldcs;
    const bool dist_loss = a.lscratch != nullptr;
    if (!dist_loss && blockIdx.y == gridDim.y - 1) {
        // the extra block row: block x == 0 produces the loss scalar, which needs every column.  Thread = 4 columns x
        // every 4th partial row, all loads independent, so this block is no slower than the row blocks.
        if (blockIdx.x != 0) return;
        float* dsc = &scr[0][0];                          // [4][256] view of the 16 x 64 scratch
        float tot = 0.f;
        const int q = (t & 63) * 4, pl = t >> 6;
        for (int c0 = 0; c0 < a.feat; c0 += 256) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            if (c0 + q < a.feat) {
                // four independent partial sums per stream: the loads of a wide feature layer's 128 partial rows overlap
                // instead of forming one dependent chain (this single block was the kernel's critical path: 0.27 ms)
                f32x4 s0 = d, s1 = d, s2 = d, s3 = d;
                int p = pl;
                for (; p + 12 < a.npart_fake; p += 16) {
                    s0 += *(const f32x4*)(a.cs + (long)p * a.ldcs + c0 + q); s1 += *(const f32x4*)(a.cs + (long)(p + 4) * a.ldcs + c0 + q);
                    s2 += *(const f32x4*)(a.cs + (long)(p + 8) * a.ldcs + c0 + q); s3 += *(const f32x4*)(a.cs + (long)(p + 12) * a.ldcs + c0 + q);
                }
                for (; p < a.npart_fake; p += 4) s0 += *(const f32x4*)(a.cs + (long)p * a.ldcs + c0 + q);
                p = pl;
                for (; p + 12 < a.npart_real; p += 16) {
                    s0 -= *(const f32x4*)(cs_real + (long)p * a.ldcs + c0 + q); s1 -= *(const f32x4*)(cs_real + (long)(p + 4) * a.ldcs + c0 + q);
                    s2 -= *(const f32x4*)(cs_real + (long)(p + 8) * a.ldcs + c0 + q); s3 -= *(const f32x4*)(cs_real + (long)(p + 12) * a.ldcs + c0 + q);
                }
                for (; p < a.npart_real; p += 4) s0 -= *(const f32x4*)(cs_real + (long)p * a.ldcs + c0 + q);
                d = (s0 + s1) + (s2 + s3);
            }
            __syncthreads();
            *(f32x4*)(dsc + pl * 256 + q) = d;
            __syncthreads();
            float v = 0.f;
            if (c0 + t < a.feat_valid) { v = (dsc[t] + dsc[256 + t] + dsc[512 + t] + dsc[768 + t]) / a.count; v *= v; }
            v = wave_sum(v);
            if ((t & 63) == 0) sf[t >> 6] = v;
            __syncthreads();
            if (t == 0) tot += sf[0] + sf[1] + sf[2] + sf[3];
        }
        if (t == 0) {
            const float loss = tot / (float)a.feat_valid;
            if (a.loss_out) *a.loss_out = loss;
            if (a.accum) *a.accum += loss;
        }
        return;
    }
    fold_partials(a.cs, a.npart_fake, a.ldcs, col0, a.feat, scr, sf);
    fold_partials(cs_real, a.npart_real, a.ldcs, col0, a.feat, scr, sr);
    if (t < CB) {
        const float diff = (col0 + t < a.feat_valid) ? (sf[t] - sr[t]) / a.count : 0.f;
        gj_lds[t] = a.grad_scale * 2.0f / ((float)a.feat_valid * a.count) * diff;
        if (dist_loss && blockIdx.y == 0) {
            // this column block's share of sum_j diff_j^2 (one wave: CB == 64), then the ticket
            const float part = wave_sum(diff * diff);
            __shared__ unsigned int ticket;
            if (t == 0) {
                a.lscratch[blockIdx.x] = part;
                __threadfence();
                ticket = atomicAdd(a.lcount, 1u);
            }
            if (t == 0 && ticket == gridDim.x - 1) {
                __threadfence();
                float tot = 0.f;
                for (unsigned int i = 0; i < gridDim.x; ++i) tot += ((volatile float*)a.lscratch)[i];      // fixed order: reproducible
                const float loss = tot / (float)a.feat_valid;
                if (a.loss_out) *a.loss_out = loss;
                if (a.accum) *a.accum += loss;
                *a.lcount = 0u;
            }
        }
    }
    __syncthreads();
    const int cg = t & 7, rl = t >> 3, c0 = col0 + cg * 8;
    if (c0 >= a.feat) return;
    T* dpre = (T*)a.dpre;
    const float q8s = a.q8_slot ? a.q8_slot->scale : 1.f;
    float q8_amax = 0.f;
    const int r1 = min(a.rows, (int)(blockIdx.y + 1) * a.rb);
    for (int r = blockIdx.y * a.rb + rl; r < r1; r += 32) {
        // lane-native mask layout (gemm.h): per (32-row block, column) two u16 words, one per lane half
        const int rr = r & 31, half = (rr >> 2) & 1, bit = (rr & 3) | ((rr >> 3) << 2);
        const uint32_t* mp = (const uint32_t*)(a.mask + ((long)(r >> 5) * a.ldm + c0) * 2);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (((mp[i] >> (16 * half)) >> bit) & 1u) ? gj_lds[cg * 8 + i] : 0.f;
        if (dpre) store8<T>(dpre + (long)r * a.ldd + c0, v);
        if (a.q8_slot) {                          // fp8 mode: the e5m2 copy the dX product reads
#pragma unroll
            for (int i = 0; i < 8; ++i) q8_amax = fmaxf(q8_amax, fabsf(v[i]));
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
            *(u32x2_t*)(a.q8 + (long)r * a.ldq8 + c0) = (u32x2_t){fp8_pack4<FP8_E5M2>(v[0], v[1], v[2], v[3], q8s), fp8_pack4<FP8_E5M2>(v[4], v[5], v[6], v[7], q8s)};
        }
    }
    if (a.q8_slot) fp8_amax_commit(a.q8_slot, q8_amax);
}

// two sets of per-row-tile partial sums -> two rows of `out` (out[0..n) and out[n..2n)); blockIdx.y picks the set
// blockIdx.z: independent segments (their partial rows follow each other, `seg_part` floats apart; their results 2 n apart)
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* part1, const float* part2, int npart, int ld, int n, float* out, long seg_part) {
    __shared__ float scr[16][CB], res[CB];
    const int col0 = blockIdx.x * CB;
    fold_partials((blockIdx.y ? part2 : part1) + (long)blockIdx.z * seg_part, npart, ld, col0, n, scr, res);
    if (threadIdx.x < CB && col0 + threadIdx.x < n) out[((long)blockIdx.z * 2 + blockIdx.y) * n + col0 + threadIdx.x] = res[threadIdx.x];
}

// =========================================================================================
// Adam, Keras 2.0.9: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr_t m / (sqrt(v) + eps).
// One block = one 64x64 tile of one tensor.  The gradient is the sum of `nslab` fp32 slabs (split-K
// weight-gradient slabs, per-row-tile bias partial sums), so no separate reduction pass exists on
// one GPU.  Also refreshes the bf16 weight copies W[K][N] and W^T[N][K] the bf16 GEMMs read.
// Small tiles (<= 64 four-element groups: biases, BN affine) spread their slabs over the otherwise
// idle threads ("slab lanes") and combine through LDS.
// =========================================================================================
__device__ __forceinline__ f32x4 adam_sum_slabs(const float* g, long stride, int first, int step, int nslab) {
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, g2 = g0, g3 = g0;
    int sl = first;
    for (; sl + 3 * step < nslab; sl += 4 * step) {
        g0 += *(const f32x4*)(g + (long)sl * stride);
        g1 += *(const f32x4*)(g + (long)(sl + step) * stride);
        g2 += *(const f32x4*)(g + (long)(sl + 2 * step) * stride);
        g3 += *(const f32x4*)(g + (long)(sl + 3 * step) * stride);
    }
    for (; sl < nslab; sl += step) g0 += *(const f32x4*)(g + (long)sl * stride);
    return (g0 + g1) + (g2 + g3);
}

// the flat gradient of one four-element group: fp32, or bfloat16 when the gradients travel as bfloat16 (MRGAN_FLAG_GRAD_BF16)
__device__ __forceinline__ f32x4 flat_load(const AdamTile& tile, long off) {
    if (tile.flat16) { const bf16x4 v = *(const bf16x4*)(tile.flat16 + off); return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
    return *(const f32x4*)(tile.flat + off);
}
__device__ __forceinline__ void flat_store(const AdamTile& tile, long off, const f32x4 g) {
    if (tile.flat16) *(bf16x4*)(tile.flat16 + off) = (bf16x4){(__bf16)g[0], (__bf16)g[1], (__bf16)g[2], (__bf16)g[3]};
    else *(f32x4*)(tile.flat + off) = g;
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
    __shared__ float tl[64 * 65];
    const int t = threadIdx.x;
    // block 0 is the serial one (next DevState with two double-precision pow(), the loss partials folded in a fixed order): first
    // in dispatch order, so that it runs beside the tile blocks instead of behind them
    const int tile_id = (int)blockIdx.x - 1;
    if (tile_id >= 0) {
    const AdamTile tile = a.tiles[tile_id];
    const float lr_t = a.st->lr_t;
    const int qpr = tile.cols >> 2;                       // four-element groups per row
    const int nq = tile.rows * qpr;
    const bool small = nq <= 64 && !tile.wt16;            // slab-parallel mode
    const float w8_scale = tile.w8_slot ? tile.w8_slot->scale : 1.f;
    float w8_amax = 0.f;
    const int QL = nq <= 16 ? 16 : 64;                    // element-group slots; the other threads are slab lanes
    const int lanes = small ? 256 / QL : 1;
    // update of one four-element group: Keras-2.0.9 Adam (eps outside the square root) + the bf16 / fp8 weight copies
    auto update = [&](const f32x4 g, f32x4 m, f32x4 v, f32x4& pn, long off) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            m[i] = a.b1 * m[i] + (1.0f - a.b1) * g[i];
            v[i] = a.b2 * v[i] + (1.0f - a.b2) * g[i] * g[i];
            pn[i] = pn[i] - lr_t * m[i] / (sqrtf(v[i]) + a.eps);
        }
        *(f32x4*)(tile.m + off) = m; *(f32x4*)(tile.v + off) = v; *(f32x4*)(tile.p + off) = pn;
        if (tile.w16) {
            bf16x4 w = {(__bf16)pn[0], (__bf16)pn[1], (__bf16)pn[2], (__bf16)pn[3]};
            *(bf16x4*)(tile.w16 + off) = w;
            if (tile.w8_slot) {           // fp8 mode: e4m3 copy of the bf16 values (what quant8_kernel would produce from w16)
#pragma unroll
                for (int i = 0; i < 4; ++i) { pn[i] = (float)w[i]; w8_amax = fmaxf(w8_amax, fabsf(pn[i])); }
                *(uint32_t*)(tile.w8 + off) = fp8_pack4<FP8_E4M3>(pn[0], pn[1], pn[2], pn[3], w8_scale);
            }
        }
    };
    if (small) {
        int sl0 = t / QL;
        const int qi = t & (QL - 1);
        int r = qi / qpr;
        const int tc = (qi - r * qpr) * 4;
        if (qi >= nq) r = tile.rows;
        const bool valid = r < tile.rows && tc < tile.cols;
        const long off = (long)r * tile.ld + tc;
        f32x4 g = {0.f, 0.f, 0.f, 0.f}, pn = g;
        if (valid) {
            if (a.mode == ADAM_FROM_FLAT) { if (sl0 == 0) g = flat_load(tile, off); }
            else g = adam_sum_slabs(tile.g + off, tile.slab_stride, sl0, lanes, tile.nslab);
        }
        *(f32x4*)(tl + t * 4) = g;                        // combine the slab lanes
        __syncthreads();
        if (sl0 == 0)
            for (int k = 1; k < lanes; ++k) g += *(const f32x4*)(tl + (t + QL * k) * 4);
        if (valid && sl0 == 0) {
            if (a.mode == ADAM_REDUCE_ONLY) flat_store(tile, off, g);
            else { pn = *(const f32x4*)(tile.p + off); update(g, *(const f32x4*)(tile.m + off), *(const f32x4*)(tile.v + off), pn, off); }
        }
    } else {
        // four row groups per thread; all their loads (gradient slabs, m, v, p) are issued before the first store (a load behind
        // a store through another pointer is not hoisted by the compiler)
        const int tc = (t & 15) * 4;
        f32x4 g[4], m[4], v[4], pn[4];
        long off[4];
        bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = (t >> 4) + 16 * u;
            valid[u] = r < tile.rows && tc < tile.cols;
            off[u] = (long)r * tile.ld + tc;
            g[u] = m[u] = v[u] = pn[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (valid[u]) {
                g[u] = a.mode == ADAM_FROM_FLAT ? flat_load(tile, off[u]) : adam_sum_slabs(tile.g + off[u], tile.slab_stride, 0, 1, tile.nslab);
                if (a.mode != ADAM_REDUCE_ONLY) { m[u] = *(const f32x4*)(tile.m + off[u]); v[u] = *(const f32x4*)(tile.v + off[u]); pn[u] = *(const f32x4*)(tile.p + off[u]); }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = (t >> 4) + 16 * u;
            if (valid[u]) {
                if (a.mode == ADAM_REDUCE_ONLY) flat_store(tile, off[u], g[u]);
                else update(g[u], m[u], v[u], pn[u], off[u]);
            }
            if (tile.wt16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) tl[(tc + i) * 65 + r] = pn[u][i];
            }
        }
    }
    if (tile.wt16 && a.mode != ADAM_REDUCE_ONLY) {
        __syncthreads();
        const int cc = t >> 2, rr0 = (t & 3) * 16;
        if (cc < tile.cols && rr0 < tile.rows) {
            bf16x8 lo, hi;
#pragma unroll
            for (int i = 0; i < 8; ++i) { lo[i] = (__bf16)tl[cc * 65 + rr0 + i]; hi[i] = (__bf16)tl[cc * 65 + rr0 + 8 + i]; }
            *(bf16x8*)(tile.wt16 + (long)cc * tile.ldt + rr0) = lo;
            *(bf16x8*)(tile.wt16 + (long)cc * tile.ldt + rr0 + 8) = hi;
            if (tile.w8_slot) {
                typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
                float f[16];
#pragma unroll
                for (int i = 0; i < 8; ++i) { f[i] = (float)lo[i]; f[8 + i] = (float)hi[i]; }
                *(u32x4_t*)(tile.w8t + (long)cc * tile.ldt + rr0) =
                    (u32x4_t){fp8_pack4<FP8_E4M3>(f[0], f[1], f[2], f[3], w8_scale), fp8_pack4<FP8_E4M3>(f[4], f[5], f[6], f[7], w8_scale),
                              fp8_pack4<FP8_E4M3>(f[8], f[9], f[10], f[11], w8_scale), fp8_pack4<FP8_E4M3>(f[12], f[13], f[14], f[15], w8_scale)};
            }
        }
    }
    if (tile.w8_slot && a.mode != ADAM_REDUCE_ONLY) fp8_amax_commit(tile.w8_slot, w8_amax);
    }
    // ---- block 0: the next sub-step's DevState, and this sub-step's loss partials ----
    if (tile_id < 0 && t == 0 && a.next && a.mode != ADAM_REDUCE_ONLY) {
        const DevState st = *a.st;
        DevState nx;
        nx.iter = st.iter + 1;
        nx.batch = st.batch + (uint32_t)a.advance_batch;
        const double tt = (double)nx.iter + 1.0;
        nx.lr_t = (float)((double)a.lr * sqrt(1.0 - pow((double)a.b2, tt)) / (1.0 - pow((double)a.b1, tt)));
        nx.pad = 0;
        *a.next = nx;
    }
    if (tile_id < 0 && a.step_out) {
        __shared__ float ms[3][256];
        float s[3] = {0.f, 0.f, 0.f};
        if (a.mode == ADAM_FROM_FLAT) { if (t == 0) for (int i = 0; i < 3; ++i) s[i] = a.flat_tail[i]; }
        else {
            for (int b = t; b < a.nloss_part; b += 256)
                for (int i = 0; i < 3; ++i) s[i] += a.loss_part[b * 4 + i];
            for (int i = 0; i < 3; ++i) ms[i][t] = s[i];
            __syncthreads();
            if (t == 0) {
                for (int i = 0; i < 3; ++i) {
                    float acc = 0.f;
                    for (int k = 0; k < 256; ++k) acc += ms[i][k];       // fixed order: reproducible
                    s[i] = acc * a.inv_rows;
                }
            }
        }
        if (t == 0) {
            if (a.mode == ADAM_REDUCE_ONLY) { for (int i = 0; i < 3; ++i) a.flat_tail[i] = s[i]; a.flat_tail[3] = 0.f; }
            else { for (int i = 0; i < 3; ++i) { a.step_out[i] = s[i]; a.accum[i] += s[i]; } }
        }
    }
}

__global__ __launch_bounds__(64) void noise_debug_kernel(uint64_t seed, uint32_t site, uint32_t seg, uint32_t step, uint32_t row0,
                                                         int rows, int cols, float* out) {
    const int lane = threadIdx.x, lc = lane & 31, lh = lane >> 5;
    const int rbase = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const uint32_t rh = noise_rowhash(noise_key(seed, site * 256u + seg, step), row0 + (uint32_t)(rbase + lc));
    const i32x16 nz = noise_block(rh, (uint32_t)c0 >> 5, lane, hadamard_frag(lane));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < rows && c0 + lc < cols) out[(long)row * cols + c0 + lc] = NOISE_SCALE * (float)nz[r];
    }
}

}  // namespace

#define LAUNCH_T(kern, grid, block, smem, s, args)                                  \
    do {                                                                            \
        if (bf16) MRGAN_LAUNCH(kern<__bf16>, grid, block, smem, s, args);     \
        else MRGAN_LAUNCH(kern<float>, grid, block, smem, s, args);           \
    } while (0)
#define RET_LAUNCH return hipGetLastError() == hipSuccess ? 0 : -2

int launch_stage(int bf16, const StageArgs& a, hipStream_t s) {
    int maxc = 0, maxr = 0;
    for (int i = 0; i < a.nseg; ++i) { maxc = max(maxc, a.s[i].cols_pad); maxr = max(maxr, a.s[i].rows); }
    dim3 grid(ceil_div(maxc, 128), ceil_div(maxr, 128), a.nseg);
    LAUNCH_T(stage_kernel, grid, dim3(256), 0, s, a);
    RET_LAUNCH;
}

int launch_bn_apply(int bf16, const BnApplyArgs& a, hipStream_t s) {
    dim3 grid(ceil_div(a.ld, CB), ceil_div(a.rows, RB), std::max(1, a.nseg));
    LAUNCH_T(bn_apply_kernel, grid, dim3(256), 0, s, a);
    RET_LAUNCH;
}

int stat_row_blocks(int rows) { return ceil_div(rows, RB); }

int launch_bn_bwd(int bf16, const BnBwdArgs& a, hipStream_t s) {
    dim3 grid(ceil_div(a.ld, CB), ceil_div(a.rows, RB));
    LAUNCH_T(bn_bwd_kernel, grid, dim3(256), 0, s, a);
    RET_LAUNCH;
}

int init_kernel_attributes() {
    // must run outside stream capture; called from mrgan_create
    hipError_t e = hipFuncSetAttribute((const void*)head_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)head_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    return e == hipSuccess ? 0 : -2;
}

int launch_head(int bf16, const HeadArgs& a, hipStream_t s) {
    const int ch = std::min(a.feat, HEAD_CHUNK);
    if ((a.feat % 64) != 0 || a.classes > KMAX) return -3;
    const size_t smem = sizeof(float) * ((size_t)HR * (ch + 8) + (size_t)ch * KMAX + HR * KMAX + 16);
    dim3 grid(ceil_div(a.rows, HR), a.nseg);
    if (a.q8_slot) {
        if (!bf16 || !(a.q8 || a.q8t)) return -3;
        static DeviceOnce attr;
        if (attr.first()) {
            if (hipFuncSetAttribute((const void*)head_kernel<__bf16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return -2;
            attr.mark();
        }
        MRGAN_LAUNCH((head_kernel<__bf16, true>), grid, dim3(256), smem, s, a);
    } else {
        LAUNCH_T(head_kernel, grid, dim3(256), smem, s, a);
    }
    RET_LAUNCH;
}

int launch_reduce_partials(const float* src, int nsrc, long stride, int n, int ngroups, float* dst, hipStream_t s) {
    MRGAN_LAUNCH(reduce_partials_kernel, dim3(ceil_div(n, 256), ngroups), dim3(256), 0, s, src, nsrc, stride, n, ngroups, dst);
    RET_LAUNCH;
}

int launch_fm(int bf16, const FmArgs& a, hipStream_t s) {
    if ((a.feat % 8) != 0) return -3;
    // every block first folds the column partial sums of its 64 columns (all row tiles): with a wide feature layer that
    // prologue would be repeated by thousands of 64-row blocks (0.34 ms at 4096 features x 8192 rows), so blocks take more rows
    // once ~512 of them exist
    FmArgs b = a;
    const int gx = ceil_div(a.feat, CB);
    b.rb = std::max(RB, (int)round_up(ceil_div(a.rows, std::max(1, 512 / gx)), 32));
    const bool dist = gx >= 16 && a.lscratch && a.lcount;
    if (!dist) { b.lscratch = nullptr; b.lcount = nullptr; }
    dim3 grid(gx, ceil_div(a.rows, b.rb) + (dist ? 0 : 1));      // + the loss block row of the narrow form
    LAUNCH_T(fm_kernel, grid, dim3(256), 0, s, b);
    RET_LAUNCH;
}

int launch_colsum_finalize(const float* part1, const float* part2, int npart, int ld, int n, float* out, hipStream_t s, int nseg) {
    if (n % 4) return -3;
    MRGAN_LAUNCH(colsum_finalize_kernel, dim3(ceil_div(n, CB), 2, nseg), dim3(256), 0, s, part1, part2, npart, ld, n, out, (long)npart * ld);
    RET_LAUNCH;
}

int launch_adam(const AdamArgs& a, hipStream_t s) {
    MRGAN_LAUNCH(adam_kernel, dim3(a.ntiles + 1), dim3(256), 0, s, a);
    RET_LAUNCH;
}

int launch_noise_debug(uint64_t seed, uint32_t site, uint32_t seg, uint32_t step, uint32_t row0, int rows, int cols,
                       float* out, hipStream_t s) {
    MRGAN_LAUNCH(noise_debug_kernel, dim3(ceil_div(cols, 32), ceil_div(rows, 32)), dim3(64), 0, s, seed, site, seg,
                       step, row0, rows, cols, out);
    RET_LAUNCH;
}

}  // namespace mrgan
