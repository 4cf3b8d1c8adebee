// Non-GEMM kernels of the mr_gan training path (HBM-bound elementwise / reduction work).
#include "aux_kernels.h"

namespace mrgan {
namespace {

// =========================================================================================
// stage: gather rows (optionally by index) of the resident fp32 matrix, add GaussianNoise(sigma)
// (mr_gan.py:118), convert to T, zero the padding columns.  Also draws z when asked to, and --
// being the first kernel of every sub-step -- publishes the next DevState slot.
// thread <-> (4 row-groups of 4 rows, one column): a Philox call yields the 4 normals of one
// row-group at one column, and consecutive lanes touch consecutive columns (coalesced).
// =========================================================================================
template <typename T>
__global__ __launch_bounds__(256) void stage_kernel(const StageArgs a) {
    const DevState st = *a.cur;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && a.next) {
        DevState nx;
        nx.iter = st.iter + 1;
        nx.batch = st.batch + (uint32_t)a.advance_batch;
        const double t = (double)nx.iter + 1.0;
        nx.lr_t = (float)((double)a.lr * sqrt(1.0 - pow((double)a.b2, t)) / (1.0 - pow((double)a.b1, t)));
        nx.pad = 0;
        *a.next = nx;
    }
    const StageSeg& sg = a.s[blockIdx.z];
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= sg.cols_pad) return;
    const long o = sg.stream ? (long)st.batch * sg.rows : 0;
    T* out = (T*)sg.out;
    const bool colvalid = col < sg.cols;
    const bool draw = colvalid && (sg.gen || sg.sigma > 0.f);
#pragma unroll 1
    for (int qq = 0; qq < 4; ++qq) {
        const int r4 = (blockIdx.y * 4 + qq) * 4;
        if (r4 >= sg.rows) break;
        float nz[4] = {0.f, 0.f, 0.f, 0.f};
        if (draw) normal4(a.seed, sg.site * 256u + sg.seg, st.iter, (a.row0 + (uint32_t)r4) >> 2, (uint32_t)col, nz);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r4 + j;
            if (row >= sg.rows) break;
            float v = 0.f;
            if (colvalid) {
                if (sg.gen) v = nz[j];
                else {
                    const long sr = sg.idx ? (long)sg.idx[o + row] : (o + row);
                    v = sg.src[sr * sg.ld + col] + sg.sigma * nz[j];
                }
            }
            out[(long)row * sg.ldo + col] = Elem<T>::from_f32(v);
        }
    }
}

// =========================================================================================
// BatchNorm forward with batch statistics (biased variance, eps inside the sqrt)
// =========================================================================================
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyArgs a) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= a.ld) return;
    float s1 = 0.f, s2 = 0.f;
    for (int p = 0; p < a.npart; ++p) { s1 += a.cs1[(long)p * a.ldcs + col]; s2 += a.cs2[(long)p * a.ldcs + col]; }
    const float mean = s1 / a.count;
    const float var = fmaxf(s2 / a.count - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + a.eps);
    float scale = 0.f, shift = 0.f;
    if (col < a.cols) { scale = a.gamma[col] * rstd; shift = a.beta[col] - mean * scale; }
    if (blockIdx.y == 0) { a.mu[col] = mean; a.rstd[col] = rstd; }
    const T* h = (const T*)a.h;
    T* out = (T*)a.out;
    const int r0 = blockIdx.y * 32, r1 = min(a.rows, r0 + 32);
    for (int r = r0; r < r1; ++r)
        out[(long)r * a.ld + col] = Elem<T>::from_f32(Elem<T>::to_f32(h[(long)r * a.ld + col]) * scale + shift);
}

// BatchNorm backward fused with the softplus derivative of the dense layer in front of it
// (h = softplus(pre)  =>  sigmoid(pre) = 1 - exp(-h)); emits the bias-gradient partial sums.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_kernel(const BnBwdArgs a) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= a.ld) return;
    float dbeta = 0.f, dgamma = 0.f;
    for (int p = 0; p < a.npart; ++p) { dbeta += a.cs1[(long)p * a.ldcs + col]; dgamma += a.cs2[(long)p * a.ldcs + col]; }
    float g = 0.f, mu = 0.f, rs = 0.f;
    if (col < a.cols) { g = a.gamma[col]; mu = a.mu[col]; rs = a.rstd[col]; }
    const float k = g * rs / a.count;
    const T* dy = (const T*)a.dy;
    const T* h = (const T*)a.h;
    T* dpre = (T*)a.dpre;
    const int r0 = blockIdx.y * a.rows_per_block, r1 = min(a.rows, r0 + a.rows_per_block);
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) {
        const float hv = Elem<T>::to_f32(h[(long)r * a.ld + col]);
        const float d = Elem<T>::to_f32(dy[(long)r * a.ld + col]);
        const float xh = (hv - mu) * rs;
        const float dh = k * (a.count * d - dbeta - xh * dgamma);
        const float dp = dh * (-expm1f(-hv));
        dpre[(long)r * a.ld + col] = Elem<T>::from_f32(dp);
        acc += dp;
    }
    a.db_part[(long)blockIdx.y * a.ld + col] = acc;
}

// =========================================================================================
// loss head: logits = f W6 + b6 ; labeled / unlabeled / fake losses of mr_gan.py:146-149 ; train error
// :161 ; closed-form dlogits ; dW6, db6 ; and dL/d(pre5) = (dlogits W6^T) * [f > 0].
// One block = 64 rows of one segment.  LDS: f tile as fp32 [64][feat+4], W6 [feat][8], dlogits [64][8].
// =========================================================================================
constexpr int HR = HEAD_ROWS;
template <typename T>
__global__ __launch_bounds__(256) void head_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float hl[];
    const int LDF = a.feat + 4;
    float* f_lds = hl;                         // [HR][LDF]
    float* w_lds = f_lds + HR * LDF;           // [feat][KMAX]
    float* dl_lds = w_lds + a.feat * KMAX;     // [HR][KMAX]
    float* red = dl_lds + HR * KMAX;           // [4 waves][4]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int seg = blockIdx.y, kind = a.seg_kind[seg];
    const int row_blk = blockIdx.x * HR;
    const int blk = seg * gridDim.x + blockIdx.x;
    const T* f = (const T*)a.f + (long)seg * a.f_bs;

    for (int i = t; i < HR * a.feat; i += 256) {
        const int r = i / a.feat, c = i - r * a.feat;
        const int row = row_blk + r;
        f_lds[r * LDF + c] = (row < a.rows) ? Elem<T>::to_f32(f[(long)row * a.ldf + c]) : 0.f;
    }
    for (int i = t; i < a.feat * KMAX; i += 256) {
        const int k = i / KMAX, c = i - k * KMAX;
        w_lds[i] = (c < a.classes && k < a.feat_valid) ? a.w[(long)k * a.ldw + c] : 0.f;
    }
    __syncthreads();

    // ---- logits: 4 lanes per row, each over an interleaved quarter of the features ----
    const int r = t >> 2, part = t & 3;
    float l[KMAX];
#pragma unroll
    for (int c = 0; c < KMAX; ++c) l[c] = 0.f;
    for (int kk = 0; kk < a.feat / 4; ++kk) {
        const int k = kk * 4 + part;
        const float fv = f_lds[r * LDF + k];
#pragma unroll
        for (int c = 0; c < KMAX; ++c) l[c] = fmaf(fv, w_lds[k * KMAX + c], l[c]);
    }
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        l[c] += __shfl_xor(l[c], 1, 64);
        l[c] += __shfl_xor(l[c], 2, 64);
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
        if (kind == HEAD_LAB || kind == HEAD_EVAL) {
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
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
            dl_lds[r * KMAX + c] = dl[c];
            if (a.logits && rowvalid) a.logits[(long)seg * a.logits_bs + (long)row * KMAX + c] = (c < a.classes) ? l[c] : 0.f;
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
    if (t < KMAX) {
        float s = 0.f;
        for (int rr = 0; rr < HR; ++rr) s += dl_lds[rr * KMAX + t];
        a.db_part[blk * KMAX + t] = s;
    }
    for (int j = t; j < a.feat; j += 256) {
        float wj[KMAX], dw[KMAX];
#pragma unroll
        for (int c = 0; c < KMAX; ++c) { wj[c] = w_lds[j * KMAX + c]; dw[c] = 0.f; }
        float dbf = 0.f;
        T* dpre = (T*)a.dpre + (long)seg * a.dpre_bs;
        for (int rr = 0; rr < HR; ++rr) {
            const float fv = f_lds[rr * LDF + j];
            float dfe = 0.f;
#pragma unroll
            for (int c = 0; c < KMAX; ++c) {
                const float d = dl_lds[rr * KMAX + c];
                dfe = fmaf(d, wj[c], dfe);
                dw[c] = fmaf(fv, d, dw[c]);
            }
            const float dp = (fv > 0.f) ? dfe : 0.f;
            if (row_blk + rr < a.rows) dpre[(long)(row_blk + rr) * a.ldd + j] = Elem<T>::from_f32(dp);
            dbf += dp;
        }
#pragma unroll
        for (int c = 0; c < KMAX; ++c) a.dw_part[((long)blk * a.feat + j) * KMAX + c] = dw[c];
        a.dbf_part[(long)blk * a.ldbf + j] = dbf;
    }
}

// =========================================================================================
// feature matching: loss = mean_j (mean_b f_fake - mean_b f_real)^2 ; dL/df_fake broadcast over rows,
// masked by the feature layer's ReLU.
// =========================================================================================
template <typename T>
__global__ __launch_bounds__(256) void fm_kernel(const FmArgs a) {
    __shared__ float red[4];
    const int j = blockIdx.x * 256 + threadIdx.x;
    float diff = 0.f;
    if (j < a.feat_valid) {
        float sf = 0.f, sr = 0.f;
        for (int p = 0; p < a.npart_fake; ++p) sf += a.cs[(long)p * a.ldcs + j];
        for (int p = 0; p < a.npart_real; ++p) sr += a.cs[(long)(a.npart_fake + p) * a.ldcs + j];
        diff = (sf - sr) / a.count;
    }
    if (blockIdx.y == 0 && gridDim.x == 1) {
        float s = wave_sum(diff * diff);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float loss = (red[0] + red[1] + red[2] + red[3]) / (float)a.feat_valid;
            if (a.loss_out) *a.loss_out = loss;
            if (a.accum) *a.accum += loss;
        }
    }
    if (j >= a.feat) return;
    const float gj = 2.0f / ((float)a.feat_valid * a.count) * diff;
    T* dpre = (T*)a.dpre;
    const int r0 = blockIdx.y * a.rows_per_block, r1 = min(a.rows, r0 + a.rows_per_block);
    for (int r = r0; r < r1; ++r) {
        const uint32_t w = a.mask[(long)r * a.ldm + (j >> 5)];
        dpre[(long)r * a.ldd + j] = Elem<T>::from_f32(((w >> (j & 31)) & 1u) ? gj : 0.f);
    }
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* part, int npart, int ld, int n, float* out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    float s = 0.f;
    for (int p = 0; p < npart; ++p) s += part[(long)p * ld + j];
    out[j] = s;
}

// =========================================================================================
// Adam, Keras 2.0.9: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr_t m / (sqrt(v) + eps).
// One block = one 64x64 tile of one tensor.  The gradient is the sum of `nslab` fp32 slabs (split-K
// weight-gradient slabs, per-row-tile bias partial sums), so no separate reduction pass exists on
// one GPU.  Also refreshes the bf16 weight copies W[K][N] and W^T[N][K] the bf16 GEMMs read.
// =========================================================================================
__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
    __shared__ float tl[64 * 65];
    const int t = threadIdx.x;
    if (blockIdx.x < a.ntiles) {
    const AdamTile tile = a.tiles[blockIdx.x];
    const int tr = t >> 4, tc = (t & 15) * 4;
    const float lr_t = a.st->lr_t;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = tr + 16 * u;
        f32x4 pn = {0.f, 0.f, 0.f, 0.f};
        if (r < tile.rows && tc < tile.cols) {
            const long off = (long)r * tile.ld + tc;
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (a.mode == ADAM_FROM_FLAT) g = *(const f32x4*)(tile.flat + off);
            else
                for (int s = 0; s < tile.nslab; ++s) g += *(const f32x4*)(tile.g + (long)s * tile.slab_stride + off);
            if (a.mode == ADAM_REDUCE_ONLY) { *(f32x4*)(tile.flat + off) = g; continue; }
            f32x4 m = *(const f32x4*)(tile.m + off), v = *(const f32x4*)(tile.v + off);
            pn = *(const f32x4*)(tile.p + off);
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
            }
        }
        if (tile.wt16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) tl[(tc + i) * 65 + r] = pn[i];
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
        }
    }
    }
    // ---- metrics: the extra last block folds this sub-step's loss partials ----
    if (blockIdx.x == a.ntiles && t == 0 && a.step_out) {
        float s[3] = {0.f, 0.f, 0.f};
        if (a.mode == ADAM_FROM_FLAT) { for (int i = 0; i < 3; ++i) s[i] = a.flat_tail[i]; }
        else {
            for (int b = 0; b < a.nloss_part; ++b)
                for (int i = 0; i < 3; ++i) s[i] += a.loss_part[b * 4 + i];
            for (int i = 0; i < 3; ++i) s[i] *= a.inv_rows;
        }
        if (a.mode == ADAM_REDUCE_ONLY) { for (int i = 0; i < 3; ++i) a.flat_tail[i] = s[i]; a.flat_tail[3] = 0.f; }
        else { for (int i = 0; i < 3; ++i) { a.step_out[i] = s[i]; a.accum[i] += s[i]; } }
    }
}

__global__ void noise_debug_kernel(uint64_t seed, uint32_t site, uint32_t seg, uint32_t step, uint32_t row0,
                                   int rows, int cols, float* out) {
    const int col = blockIdx.x * 64 + threadIdx.x;
    const int r4 = blockIdx.y * 4;
    if (col >= cols) return;
    float n[4];
    normal4(seed, site * 256u + seg, step, (row0 + (uint32_t)r4) >> 2, (uint32_t)col, n);
    for (int j = 0; j < 4; ++j) if (r4 + j < rows) out[(long)(r4 + j) * cols + col] = n[j];
}

}  // namespace

#define LAUNCH_T(kern, grid, block, smem, s, args)                                  \
    do {                                                                            \
        if (bf16) hipLaunchKernelGGL(kern<__bf16>, grid, block, smem, s, args);     \
        else hipLaunchKernelGGL(kern<float>, grid, block, smem, s, args);           \
    } while (0)
#define RET_LAUNCH return hipGetLastError() == hipSuccess ? 0 : -2

int launch_stage(int bf16, const StageArgs& a, hipStream_t s) {
    int maxc = 0, maxr = 0;
    for (int i = 0; i < a.nseg; ++i) { maxc = max(maxc, a.s[i].cols_pad); maxr = max(maxr, a.s[i].rows); }
    dim3 grid(ceil_div(maxc, 256), ceil_div(maxr, 16), a.nseg);
    LAUNCH_T(stage_kernel, grid, dim3(256), 0, s, a);
    RET_LAUNCH;
}

int launch_bn_apply(int bf16, const BnApplyArgs& a, hipStream_t s) {
    dim3 grid(ceil_div(a.ld, 256), ceil_div(a.rows, 32));
    LAUNCH_T(bn_apply_kernel, grid, dim3(256), 0, s, a);
    RET_LAUNCH;
}

int launch_bn_bwd(int bf16, const BnBwdArgs& a, hipStream_t s) {
    dim3 grid(ceil_div(a.ld, 256), ceil_div(a.rows, a.rows_per_block));
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
    if (a.feat > 256 || (a.feat % 4) != 0 || a.classes > KMAX) return -3;
    const size_t smem = sizeof(float) * ((size_t)HR * (a.feat + 4) + (size_t)a.feat * KMAX + HR * KMAX + 16);
    dim3 grid(ceil_div(a.rows, HR), a.nseg);
    LAUNCH_T(head_kernel, grid, dim3(256), smem, s, a);
    RET_LAUNCH;
}

int launch_fm(int bf16, const FmArgs& a, hipStream_t s) {
    if (a.feat > 256) return -3;
    dim3 grid(1, ceil_div(a.rows, a.rows_per_block));
    LAUNCH_T(fm_kernel, grid, dim3(256), 0, s, a);
    RET_LAUNCH;
}

int launch_colsum_finalize(const float* part, int npart, int ld, int n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, part, npart, ld, n, out);
    RET_LAUNCH;
}

int launch_adam(const AdamArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(adam_kernel, dim3(a.ntiles + 1), dim3(256), 0, s, a);
    RET_LAUNCH;
}

int launch_noise_debug(uint64_t seed, uint32_t site, uint32_t seg, uint32_t step, uint32_t row0, int rows, int cols,
                       float* out, hipStream_t s) {
    hipLaunchKernelGGL(noise_debug_kernel, dim3(ceil_div(cols, 64), ceil_div(rows, 4)), dim3(64), 0, s, seed, site, seg,
                       step, row0, rows, cols, out);
    RET_LAUNCH;
}

}  // namespace mrgan
