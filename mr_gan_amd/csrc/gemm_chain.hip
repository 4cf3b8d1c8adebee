// Row-block chain kernel (see chain.h): consecutive dense products of one 64-row block inside one launch.
//
// One workgroup = 8 waves = one block of 64 rows of one segment.  The block's current activation is an LDS image
// [K/64 k-tiles][64 rows][64 k] bf16 with the same XOR swizzle as the stand-alone forward kernel (gemm_bf16.hip), so the
// MFMA A fragments are conflict-free ds_read_b128; the weights stream through a 2-stage ring of [256 columns][64 k] tiles
// filled by LDS-DMA (buffer_load ... lds) from the XCD's L2.  Wave w owns output columns [32 w, 32 w + 32) of a 256-column
// pass over all 64 rows (two 32x32 accumulators): its column sums need no cross-wave step, and the epilogue writes the
// bf16 result straight into the LDS image that is the next product's A operand; a copy of it leaves for HBM in 16-byte
// stores because the weight-gradient launch needs every layer's input and output gradient.
// The ring never drains between products: the weight tile stream is one flat sequence over (product, pass, k-tile), and
// the first tile of the next product is already in flight while the current epilogue runs.
#include <algorithm>

#include "chain.h"
#include "gemm.h"

namespace mrgan {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// byte offset of element (row, col) inside an activation image
template <int ROWS = CH_ROWS>
__device__ __forceinline__ int act_off(int row, int col) {
    return (col >> 6) * (ROWS * 128) + kc_off(row, (col & 63) >> 3) + (col & 7) * 2;
}
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void lds_barrier() {          // LDS writes of every wave visible to every wave; VMEM left in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// wait until at most n of this wave's vector-memory operations are outstanding (they retire in issue order)
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}

#ifdef MRGAN_STAMPS
#define CH_STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_prev; st_prev = n_; } while (0)
#else
#define CH_STAMP(i)
#endif

// The weight tiles of a chain form one flat sequence over (product, pass, k-tile): the tile issued while tile g is consumed is
// tile g + 1 -- of this product, or the first tile of the next one, whose three scalars (Bt, K, N) the caller passes along.
// Everything the issue needs lives in registers for the whole product (a descriptor in SGPRs, four per-lane row offsets): no
// load from the argument block inside the k-loops.
struct BTile {
    __amdgpu_buffer_rsrc_t rs; int voff[4]; int pass_bytes;       // byte offset of 256 more columns of Bt
};
__device__ __forceinline__ void btile_setup(BTile& b, const __bf16* W, int K, int N, int wave, int lane) {
    b.rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)((long)N * K * 2), 0x00020000);      // rows >= N of Bt read as zeros
    const int lrow = lane >> 3, lp = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = (wave * 4 + i) * 8 + lrow;                   // column of the pass = row of Bt
        b.voff[i] = (R * K + ((lp ^ ((R >> 1) & 7)) << 3)) * 2;
    }
    b.pass_bytes = CH_PW * K * 2;
}
// one weight tile [256 columns][64 k] into a ring stage: 4 wave-instructions per wave, each wave its own 32 columns
__device__ __forceinline__ void issue_btile(const BTile& b, int pass, int kt, char* stage, int wave) {
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(b.rs, stage + (wave * 4 + i) * 1024, b.voff[i] + pass * b.pass_bytes, kt * 128);
}

// copy a [64 rows][256 columns] bf16 LDS image (columns col0 .. of the global tensor) out with 16-byte stores
template <int ROWS = CH_ROWS>
__device__ __forceinline__ void copy_out(const char* img, __bf16* out, int ldo, int col0, int ncols, int rows_valid, int t) {
#pragma unroll
    for (int u = 0; u < ROWS * CH_PW / 8 / CH_THREADS; ++u) {
        const int q = t + CH_THREADS * u, r = q >> 5, cch = q & 31;
        if (r < rows_valid && col0 + cch * 8 < ncols)
            *(u32x4*)(out + (long)r * ldo + col0 + cch * 8) = *(const u32x4*)(img + (cch >> 3) * (ROWS * 128) + kc_off(r, cch & 7));
    }
}

// Shared state of the weight-tile stream of one block (all wave-uniform)
struct Stream {
    int gtile;                 // tiles consumed so far: tile g lives in ring stage g % (number of stages)
    int inflight;              // tiles issued and not yet consumed (tile gtile is the oldest)
    // The copy of a finished output image to HBM is DEFERRED to the end of the next pass's k-loop.  Stores count in vmcnt in issue
    // order with the weight-tile DMAs: issued right behind the image barrier (round 2) they sat in front of the next tile's
    // DMA, and the wait for that tile -- one k-tile of MFMAs later -- also waited for the stores' acknowledgement (~1 us per
    // pass, measured by ablation).  At the end of a k-loop the only DMA in flight is OLDER than the stores (the next product's
    // first tile), and the following wait comes a whole epilogue later.  The image is read-only until the product after next.
    const char* cp_img; __bf16* cp_out; int cp_ldo, cp_col0, cp_ncols;
};
template <int ROWS = CH_ROWS>
__device__ __forceinline__ void flush_copy(Stream& sm, int rows_valid, int t) {
    if (sm.cp_img) copy_out<ROWS>(sm.cp_img, sm.cp_out, sm.cp_ldo, sm.cp_col0, sm.cp_ncols, rows_valid, t);
    sm.cp_img = nullptr;
}

// ------------------------------------------------------------------------------------------------------------------
// loss head on the block's 64 rows (mr_gan.py:128, :146-149, :161), on the matrix cores.
// The three small products of the head -- logits = F W6, dL/d(pre5) = (dlogits W6^T) * relu', dW6 = F^T dlogits -- were scalar
// fmaf loops (24 k cycles per block, 22 % of the launch for 0.1 % of its FLOPs).  They are MFMA products now, at fp32
// accuracy: the features are exact bf16 values already, and every fp32 factor (W6, dlogits) enters as THREE bf16 addends
// hi + mid + lo that reproduce it exactly (8 + 8 + 8 significant bits), so each bf16 x bf16 product is exact in the fp32
// accumulator and only the summation order differs from head_kernel's fmaf chain (aux_kernels.hip).
//   1. W6 -> LDS as [3 addends][class][feature] bf16 (the B operand of the logits product, k = feature contiguous)
//   2. logits: every wave takes 32 of the features as its share of the reduction (2 k-steps x 2 row tiles x 3 addends), the
//      eight partial [64][8] tiles meet in LDS
//   3. wave 0, one lane per row: softmax / losses / closed-form dlogits, dlogits -> LDS as bf16 addends, row-major
//      (A operand of 4) and class-major (A operand of 5)
//   4. dL/d(pre5) for the wave's 32 feature columns: one 16-deep k-step (8 classes + 8 zeros) x 6 addend pairs, masked with
//      the relu bits the D5 forward epilogue left in registers, written as the next product's A image (+ bias-gradient sums)
//   5. dW6^T [class][feature] = dlogits^T F for the wave's 32 features: F enters through the transposing LDS read
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split3(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)v;
    float r = v - (float)hi;            // exact: the remainder of a round-to-nearest has at most 16 significant bits
    mid = (__bf16)r;
    r -= (float)mid;                    // exact: at most 8 significant bits remain
    lo = (__bf16)r;
}
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

// what the head reads from global memory, fetched in the kernel's prologue: inside the head each of these would be an exposed
// L2 / HBM round trip with the whole block waiting at the next barrier (the labels even two dependent ones)
struct HeadInputs { f32x4 w0, w1, bw0, bw1, b0, b1; int label; };
__device__ __forceinline__ void head_prefetch(const ChainArgs& a, HeadInputs& hi, int seg, int row_blk, int rows_valid, int t) {
    const HeadArgs& h = a.head;
    const int lane = t & 63, lc = lane & 31, wave = t >> 6;
    const int k = min(t & (CH_PW - 1), h.feat_valid - 1), j = min(wave * 32 + lc, h.feat_valid - 1);
    hi.w0 = *(const f32x4*)(h.w + (long)k * h.ldw); hi.w1 = *(const f32x4*)(h.w + (long)k * h.ldw + 4);
    hi.bw0 = *(const f32x4*)(h.w + (long)j * h.ldw); hi.bw1 = *(const f32x4*)(h.w + (long)j * h.ldw + 4);
    hi.b0 = (f32x4){0.f, 0.f, 0.f, 0.f}; hi.b1 = hi.b0;
#pragma unroll
    for (int c = 0; c < 4; ++c) { hi.b0[c] = h.b[min(c, h.classes - 1)]; hi.b1[c] = h.b[min(4 + c, h.classes - 1)]; }
    hi.label = 0;
    if (h.seg_kind[seg] == HEAD_LAB) {
        const long lo = h.labels_stream ? (long)h.st->batch * h.rows : 0;
        hi.label = h.labels[lo + row_blk + min(lane, rows_valid - 1)];
    }
}

__device__ __forceinline__ void chain_head(const ChainArgs& a, char* lds, Stream& sm, const HeadInputs& hi, const uint32_t (&mw)[2][2], int seg,
                                           int rb, int nrb, int row_blk, int rows_valid, int t) {
    const HeadArgs& h = a.head;
    const char* fimg = lds + a.head_f_off;
    char* oimg = lds + a.head_o_off;
    __bf16* w6t = (__bf16*)(lds + a.head_scratch_off);        // [3][KMAX][CH_PW]
    __bf16* dl_rc = w6t + 3 * KMAX * CH_PW;                   // [3][CH_ROWS][KMAX]   dlogits addends, row-major
    __bf16* dl_t = dl_rc + 3 * CH_ROWS * KMAX;                // [3][KMAX][CH_ROWS]   ... class-major
    float* red = (float*)(dl_t + 3 * KMAX * CH_ROWS);         // [3 + KMAX][CH_ROWS]  per-row loss terms and dlogits (fp32)
    float* lpart = (float*)oimg;                              // [8 waves][CH_ROWS][KMAX]: dead before the dpre image is written
    const int lane = t & 63, lc = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kind = h.seg_kind[seg];
    const int blk = seg * nrb + rb;
    const bf16x8 zero8 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};

    if (!(a.ablate & CH_ABL_COPY)) flush_copy(sm, rows_valid, t);      // the feature image -> HBM (no DMA wait follows inside the head)
    // ---- 1. W6: thread <-> feature ----
    if (t < CH_PW) {
        const f32x4 w0 = hi.w0, w1 = hi.w1;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
            const float wv = (t < h.feat_valid && c < h.classes) ? (c < 4 ? w0[c & 3] : w1[c & 3]) : 0.f;
            __bf16 p0, p1, p2;
            split3(wv, p0, p1, p2);
            w6t[(0 * KMAX + c) * CH_PW + t] = p0; w6t[(1 * KMAX + c) * CH_PW + t] = p1; w6t[(2 * KMAX + c) * CH_PW + t] = p2;
        }
    }
    // B operand of product 4 (k = class, column = feature 32 wave + lc): the eight class weights of this lane's feature
    bf16x8 bw[3] = {zero8, zero8, zero8};
    {
        const int j = wave * 32 + lc;
        const f32x4 w0 = hi.bw0, w1 = hi.bw1;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
            const float wv = (lh == 0 && j < h.feat_valid && c < h.classes) ? (c < 4 ? w0[c & 3] : w1[c & 3]) : 0.f;      // lh = 1: k = 8 .. 15, zeros
            __bf16 p0, p1, p2;
            split3(wv, p0, p1, p2);
            bw[0][c] = p0; bw[1][c] = p1; bw[2][c] = p2;
        }
    }
    lds_barrier();

    // ---- 2. logits: this wave's 32 features of the reduction ----
    {
        f32x16 acc[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kg = 2 * wave + u;                      // k-step: features 16 kg .. 16 kg + 15
            if (16 * kg < h.feat) {                           // (wave-uniform)
                const char* As = fimg + (kg >> 2) * (CH_ROWS * 128);
                const int ch = (kg & 3) * 2 + lh;
                const bf16x8 fa0 = *(const bf16x8*)(As + kc_off(lc, ch)), fa1 = *(const bf16x8*)(As + kc_off(32 + lc, ch));
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    bf16x8 fb = *(const bf16x8*)(w6t + (p * KMAX + (lc & (KMAX - 1))) * CH_PW + 16 * kg + 8 * lh);
                    if (lc >= KMAX) fb = zero8;               // columns 8 .. 31 of the product are padding
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb, acc[1], 0, 0, 0);
                }
            }
        }
        if (lc < KMAX) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) lpart[(wave * CH_ROWS + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * KMAX + lc] = acc[mi][r];
        }
    }
    lds_barrier();

    // ---- 3. per row (wave 0: lane <-> row): losses, error, dlogits ----
    float* part_row = h.part + (long)blk * h.part_stride;
    if (wave == 0) {
        const int r = lane;
        float l[KMAX];
#pragma unroll
        for (int c = 0; c < KMAX; ++c) l[c] = 0.f;
#pragma unroll
        for (int w = 0; w < CH_THREADS / 64; ++w) {
            const f32x4 p0 = *(const f32x4*)(lpart + (w * CH_ROWS + r) * KMAX), p1 = *(const f32x4*)(lpart + (w * CH_ROWS + r) * KMAX + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) { l[c] += p0[c]; l[4 + c] += p1[c]; }
        }
        const bool rowvalid = r < rows_valid;
        float mx = -3.0e38f;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
            if (c < h.classes) { l[c] += (c < 4 ? hi.b0[c & 3] : hi.b1[c & 3]); mx = fmaxf(mx, l[c]); }
        }
        int am = 0;
        float se = 0.f, p[KMAX];
#pragma unroll
        for (int c = KMAX - 1; c >= 0; --c) {
            p[c] = (c < h.classes) ? expf(l[c] - mx) : 0.f;
            se += p[c];
            if (c < h.classes && l[c] == mx) am = c;          // ties -> first index (theano argmax)
        }
        const float lse = mx + logf(se);
        const float inv_se = 1.0f / se;
        float loss0 = 0.f, loss1 = 0.f, err = 0.f;
        float dl[KMAX];
#pragma unroll
        for (int c = 0; c < KMAX; ++c) dl[c] = 0.f;
        if (rowvalid) {
            if (kind == HEAD_LAB) {
                const int y = hi.label;
                err = (am != y) ? 1.f : 0.f;
                float ly = 0.f;
#pragma unroll
                for (int c = 0; c < KMAX; ++c) {
                    if (c == y) ly = l[c];
                    dl[c] = (p[c] * inv_se - (c == y ? 1.f : 0.f)) * h.inv_count;
                }
                loss0 = lse - ly;
            } else {
                const float sg = sigmoid_f(lse), sp = softplus_f(lse);
                const float k = 0.5f * h.inv_count * h.unl_weight * (kind == HEAD_UNL ? (sg - 1.0f) : sg);
                loss1 = (kind == HEAD_UNL) ? 0.5f * (sp - lse) : 0.5f * sp;
#pragma unroll
                for (int c = 0; c < KMAX; ++c) dl[c] = k * p[c] * inv_se;
            }
        }
        bf16x8 d3[3];
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
            __bf16 p0, p1, p2;
            split3(dl[c], p0, p1, p2);
            d3[0][c] = p0; d3[1][c] = p1; d3[2][c] = p2;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            *(bf16x8*)(dl_rc + (q * CH_ROWS + r) * KMAX) = d3[q];
#pragma unroll
            for (int c = 0; c < KMAX; ++c) dl_t[(q * KMAX + c) * CH_ROWS + r] = d3[q][c];
        }
        // the eleven per-row quantities whose sums over the 64 rows leave the block (three loss terms, db6 = column sums of
        // dlogits): to LDS, row-contiguous; eleven lanes of the last wave add them up behind the barrier (as wave-wide shuffle
        // reductions -- eleven six-step ds_bpermute chains on this one wave -- they cost ~3 us with the other seven waves waiting)
        red[0 * CH_ROWS + r] = loss0; red[1 * CH_ROWS + r] = loss1; red[2 * CH_ROWS + r] = err;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) red[(3 + c) * CH_ROWS + r] = dl[c];
    }
    lds_barrier();

    // ---- 4. dL/d(pre5) = (dlogits W6^T) * relu'(pre5) for columns 32 wave .. + 31: the next product's A image ----
    {
        f32x16 acc[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
        bf16x8 da[2][3];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                da[mi][q] = *(const bf16x8*)(dl_rc + (q * CH_ROWS + mi * 32 + lc) * KMAX);
                if (lh) da[mi][q] = zero8;                    // k = 8 .. 15: padding
            }
        // addend pairs down to 2^-24 of the product: (hi, hi) (hi, mid) (mid, hi) (hi, lo) (lo, hi) (mid, mid)
        constexpr int PA[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
        for (int i = 5; i >= 0; --i)                          // smallest terms first
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[mi][PA[i]], bw[PB[i]], acc[mi], 0, 0, 0);
        const int cip = wave * 32 + lc;
        int obase[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            obase[i] = (cip >> 6) * (CH_ROWS * 128) + lh * 512 + (((((cip & 63) >> 3) ^ (lh << 1)) ^ ((i & 1) | ((i >> 1) << 2))) << 4) + (cip & 7) * 2;
        float s1 = 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float av = acc[mi][r];
                const float v = ((mw[0][mi] >> r) & 1u) ? av : 0.f;         // (a select: see chain_gemm)
                s1 += v;
                *(__bf16*)(oimg + obase[((r >> 1) & 1) | (((r >> 2) & 1) << 1)] + (mi * 32 + (r & 3) + 8 * (r >> 2)) * 128) = (__bf16)v;
            }
        s1 += __shfl_xor(s1, 32, 64);
        if (lh == 0 && cip < h.feat) part_row[h.off_dbf + cip] = s1;       // bias gradient of the feature layer
    }
    // ---- 5. dW6^T [class][feature] = dlogits^T F, features 32 wave .. + 31 ----
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
        const int f0 = wave * 32 + (g4 & 1) * 16 + 4 * pp;
#pragma unroll
        for (int ks = 0; ks < CH_ROWS / 16; ++ks) {
            // B fragment: eight consecutive rows (k) of this lane's feature column, by the transposing read
            const int m0 = ks * 16 + (g4 >> 1) * 8 + q;
            const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(fimg + act_off(m0, f0)));
            const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(fimg + act_off(m0 + 4, f0)));
            const bf16x8 fb = __builtin_bit_cast(bf16x8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int p = 2; p >= 0; --p) {
                bf16x8 fa = *(const bf16x8*)(dl_t + (p * KMAX + (lc & (KMAX - 1))) * CH_ROWS + 16 * ks + 8 * lh);
                if (lc >= KMAX) fa = zero8;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
            }
        }
        const int j = wave * 32 + lc;                         // registers 0 .. 3 = classes 4 lh .. 4 lh + 3 of feature j
        if (j < h.feat) *(f32x4*)(part_row + (long)j * KMAX + 4 * lh) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
    }
    if (wave == CH_THREADS / 64 - 1 && lane < 3 + KMAX) {
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CH_ROWS / 4; ++i) {
            const f32x4 v = *(const f32x4*)(red + lane * CH_ROWS + 4 * i);
            s4[i & 3] += (v[0] + v[1]) + (v[2] + v[3]);
        }
        const float tot = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        if (lane < 3) h.loss_part[blk * 4 + lane] = tot;
        else part_row[h.off_db + lane - 3] = tot;
        if (lane == 0) h.loss_part[blk * 4 + 3] = 0.f;
    }
    lds_barrier();
    // dL/d(pre5): the next product's A image is complete; its copy for the weight-gradient launch leaves at the end of that
    // product's k-loop
    sm.cp_img = oimg; sm.cp_out = (__bf16*)h.dpre + (long)seg * h.dpre_bs + (long)row_blk * h.ldd;
    sm.cp_ldo = h.ldd; sm.cp_col0 = 0; sm.cp_ncols = h.feat;
}

// ------------------------------------------------------------------------------------------------------------------
// feature-matching gradient as the first A image (mr_gan.py:152-154): dL/d(pre5) = relu-mask ? 2/(J B) (m_gen - m_real) : 0
// ------------------------------------------------------------------------------------------------------------------
template <int ROWS>
__device__ __forceinline__ void chain_fmgrad(const ChainArgs& a, char* lds, int a_off, int rb, int row_blk, int rows_valid, int t
#ifdef MRGAN_STAMPS
                                             , unsigned long long (&st_acc)[8], unsigned long long& st_prev
#endif
) {
    const FmArgs& f = a.fm;
    // the stored features of this block's rows (their sign is relu'(pre5), used at the end): requested first, so that their round
    // trip runs beside the fold of the partial sums instead of behind it
    constexpr int NCH = ROWS * CH_PW / 8 / CH_THREADS;
    bf16x8 fv[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        const int q = t + CH_THREADS * u, r = q >> 5, c0 = (q & 31) * 8;
        const bool ok = r < rows_valid && c0 < f.feat;
        fv[u] = *(const bf16x8*)(a.fm_feat + (long)(row_blk + (ok ? r : 0)) * a.fm_ldf + (ok ? c0 : 0));
    }
    float* gj = (float*)(lds + a.op[0].o_off);               // 9 KiB in the first product's output image: idle until its epilogue (both ring stages are in flight)
    float* scr = gj + CH_PW;                                  // [8][256]
    const float* cs_real = f.cs + (long)f.npart_fake * f.ldcs;
    // fold the per-row-block partial sums: thread <-> (4 columns, every 8th partial row), 64 partial rows of both streams per
    // round trip (16 loads of 16 bytes in flight per thread), then an 8-way combine through LDS
    {
        const int cq = (t & 63) * 4, pg = t >> 6;
        f32x4 u = {0.f, 0.f, 0.f, 0.f};
        if (cq < f.feat) {
            // unconditional loads from clamped rows, zero weight beyond the end: a load under a runtime condition makes
            // hipcc branch around it and wait for each one (16 serialized round trips, ~30 k cycles measured here); and a
            // remainder loop `for (p = pg + 64; p < npart; p += 8) u += load` is one round trip per iteration (the 32-row
            // blocks of the G sub-step leave 128 partial rows per stream: 16 of them, 20 k cycles of a 43 k-cycle launch)
            const int nmax = max(f.npart_fake, f.npart_real);
            for (int base = 0; base < nmax; base += 64) {
                f32x4 vf[8], vr[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) vf[i] = *(const f32x4*)(f.cs + (long)min(base + pg + 8 * i, f.npart_fake - 1) * f.ldcs + cq);
#pragma unroll
                for (int i = 0; i < 8; ++i) vr[i] = *(const f32x4*)(cs_real + (long)min(base + pg + 8 * i, f.npart_real - 1) * f.ldcs + cq);
#pragma unroll
                for (int i = 0; i < 8; ++i) u += vf[i] * ((base + pg + 8 * i < f.npart_fake) ? 1.0f : 0.0f);
#pragma unroll
                for (int i = 0; i < 8; ++i) u -= vr[i] * ((base + pg + 8 * i < f.npart_real) ? 1.0f : 0.0f);
            }
        }
        *(f32x4*)(scr + pg * 256 + cq) = u;                   // scr: [8][256]
    }
    __syncthreads();
    CH_STAMP(1);
    const int c = t & 255, hf = t >> 8;
    float sq = 0.f;
    if (hf == 0) {
        float sd = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) sd += scr[g * 256 + c];
        const float diff = (c < f.feat_valid) ? sd / f.count : 0.f;
        gj[c] = f.grad_scale * 2.0f / ((float)f.feat_valid * f.count) * diff;
        sq = diff * diff;
    }
    if (rb == 0 && blockIdx.x == 0) {                         // the loss scalar, once
        sq = wave_sum(sq);
        __syncthreads();
        if ((t & 63) == 0) scr[t >> 6] = sq;
        __syncthreads();
        if (t == 0) {
            const float loss = (scr[0] + scr[1] + scr[2] + scr[3]) / (float)f.feat_valid;
            if (f.loss_out) *f.loss_out = loss;
            if (f.accum) *f.accum += loss;
        }
    }
    __syncthreads();
    CH_STAMP(2);
    char* img = lds + a_off;
    // thread <-> (row, 8 columns): one 16-byte chunk of the image.  relu'(pre5) comes from the stored features of the
    // generated rows (f > 0 <=> pre5 > 0; bf16 keeps every positive value positive): one unconditional 16-byte load per chunk
    // from a clamped row, all four in flight together.  (Decoding the lane-native mask words here instead costs 8 scattered
    // loads per chunk: ~18 k cycles per block, measured.)
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        const int q = t + CH_THREADS * u, r = q >> 5, cch = q & 31, c0 = cch * 8;
        const bool ok = r < rows_valid && c0 < f.feat;
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (__bf16)((ok && (float)fv[u][i] > 0.f) ? gj[c0 + i] : 0.f);
        *(bf16x8*)(img + (cch >> 3) * (ROWS * 128) + kc_off(r, cch & 7)) = v;
    }
    __syncthreads();
    CH_STAMP(6);
    CH_STAMP(7);        // (the image's copy to HBM is deferred to the end of the first product's k-loop: see Stream)
}

// one dense product of the chain on the block's rows.  MODE is compile-time; `bias` (forward) and `mw` (the relu-mask words
// of the output tile: read by dX, returned by forward) live in registers, loaded or produced before this call.
template <int MODE, int MI>
__device__ __forceinline__ void chain_gemm(const ChainArgs& a, const ChainOp& op, const __bf16* nextW, const int nextK, const int nextN,
                                           char* lds, Stream& sm, const float bias,
                                           uint32_t (&mw)[2][MI], const int seg, const int nrb, const int rb, const int row_blk,
                                           const int rows_valid, const uint32_t iter, const i32x4 hfrag, const int t
#ifdef MRGAN_STAMPS
                                           , unsigned long long (&st_acc)[8], unsigned long long& st_prev
#endif
) {
    const int lane = t & 63, lc = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr bool fwd = MODE == CH_FWD_RELU;
    constexpr int ROWS = 32 * MI, NS = chain_stages(ROWS);
    const int K = op.K, N = op.N, a_off = op.a_off, o_off = op.o_off;
    const int npass = (N + CH_PW - 1) / CH_PW, nk = K / 64;
    const bool noisy = fwd && op.sigma > 0.f;
    uint32_t rowhash[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) rowhash[mi] = 0u;
    if (noisy) {
        const uint32_t nkey = noise_key(a.seed, op.site * 256u + (uint32_t)(a.seg0 + seg), iter);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) rowhash[mi] = noise_rowhash(nkey, a.row0 + (uint32_t)(row_blk + mi * 32 + lc));
    }
    uint16_t* mask = op.mask ? op.mask + (long)seg * op.mask_bs : nullptr;
    BTile bt;
    btile_setup(bt, op.W, K, N, wave, lane);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass >= npass) break;
        const int col = pass * CH_PW + wave * 32 + lc;
        const bool colin = col < N, colvalid = col < op.n_valid;
        f32x16 acc[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
        CH_STAMP(1);               // pass setup
        for (int kt = 0; kt < nk; ++kt) {
            // No workgroup barrier inside the k-loop: a wave reads only ITS OWN 32 columns of a weight tile, and those are
            // exactly the pieces it loads itself (issue_btile), so its own counted vmcnt orders the LDS-DMA before its reads;
            // the stage refilled below was last read by this wave during tile gtile - 1, whose fragments its MFMAs have
            // already consumed.  The A image is read-only for the whole product (completed behind the barrier that ended the
            // previous product / the prologue).  The waves of a block drift apart inside a product -- one wave's MFMAs beside
            // another's epilogue -- and meet again at the image barrier that ends the pass.
            // Two tiles of the flat sequence are in flight while tile gtile is consumed (at one tile the wait below was the L2 round
            // trip of every tile).  With the 3-stage ring (32-row blocks) tile gtile + 2 is issued before the reads of tile gtile;
            // with the 2-stage ring (64-row blocks: the images leave 64 KiB) it goes into tile gtile's OWN stage as soon as this wave
            // has the tile's four B fragments in registers.  vmcnt counts in issue order: all but the youngest 4 operations done =
            // tile gtile has landed (stores issued since then only make the wait stricter).
            // (sm.inflight tiles are issued and not yet consumed, tile gtile the oldest of them: at the tail of the last product no
            //  younger tile exists and the wait must cover everything)
            if (!(a.ablate & CH_ABL_STREAM)) {
                if (sm.inflight > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            --sm.inflight;
            CH_STAMP(3);                                   // wait for the weight tile
            auto issue_ahead = [&](char* stage) {
                const int kk = kt + 2;                     // (every product of a chain has at least 2 k-tiles: launch_chain)
                if (kk < nk) { issue_btile(bt, pass, kk, stage, wave); ++sm.inflight; }
                else if (pass + 1 < npass) { issue_btile(bt, pass + 1, kk - nk, stage, wave); ++sm.inflight; }
                else if (nextW) { BTile nb; btile_setup(nb, nextW, nextK, nextN, wave, lane); issue_btile(nb, 0, kk - nk, stage, wave); ++sm.inflight; }
            };
            const char* As = lds + a_off + kt * (ROWS * 128);
            char* Bs = lds + chain_ring(ROWS) + (sm.gtile % NS) * CH_STAGE_BYTES;
            if constexpr (NS == 3) { if (!(a.ablate & CH_ABL_STREAM)) issue_ahead(lds + chain_ring(ROWS) + ((sm.gtile + 2) % NS) * CH_STAGE_BYTES); }
            ++sm.gtile;
            if (a.ablate & CH_ABL_MFMA) { if constexpr (NS == 2) { if (!(a.ablate & CH_ABL_STREAM)) issue_ahead(Bs); } continue; }
            // fragments of two k-steps per batch: their LDS latency is paid once per batch (the other wave of the SIMD
            // covers the rest); a deeper batch costs registers the epilogue needs
            if constexpr (NS == 3) {
#pragma unroll
                for (int kg = 0; kg < 4; kg += 2) {
                    bf16x8 fa[2][MI], fb[2];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) fa[ks][mi] = *(const bf16x8*)(As + kc_off(mi * 32 + lc, (kg + ks) * 2 + lh));
                        fb[ks] = *(const bf16x8*)(Bs + kc_off(wave * 32 + lc, (kg + ks) * 2 + lh));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mi], fb[ks], acc[mi], 0, 0, 0);
                }
            } else {
                // (named registers, not an array: hipcc puts an array it indexes in a loop into scratch memory)
                const bf16x8 fb0 = *(const bf16x8*)(Bs + kc_off(wave * 32 + lc, 0 + lh)), fb1 = *(const bf16x8*)(Bs + kc_off(wave * 32 + lc, 2 + lh));
                const bf16x8 fb2 = *(const bf16x8*)(Bs + kc_off(wave * 32 + lc, 4 + lh)), fb3 = *(const bf16x8*)(Bs + kc_off(wave * 32 + lc, 6 + lh));
                bf16x8 fa0[MI], fa1[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    fa0[mi] = *(const bf16x8*)(As + kc_off(mi * 32 + lc, 0 + lh));
                    fa1[mi] = *(const bf16x8*)(As + kc_off(mi * 32 + lc, 2 + lh));
                }
                // the stage may be refilled once its fragments are in registers (the LDS-DMA write must not overtake the reads)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!(a.ablate & CH_ABL_STREAM)) issue_ahead(Bs);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[mi], fb0, acc[mi], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[mi], fb1, acc[mi], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    fa0[mi] = *(const bf16x8*)(As + kc_off(mi * 32 + lc, 4 + lh));
                    fa1[mi] = *(const bf16x8*)(As + kc_off(mi * 32 + lc, 6 + lh));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[mi], fb2, acc[mi], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[mi], fb3, acc[mi], 0, 0, 0);
            }
            CH_STAMP(5);                                   // tile issue + fragment reads + MFMAs
        }

        if (!(a.ablate & CH_ABL_COPY)) flush_copy<ROWS>(sm, rows_valid, t);      // the previous pass's image -> HBM (see Stream)
        // every pass of a product assembles its 256 columns in the SAME image: a second pass may only overwrite them when every
        // wave has copied the first pass's out (the only such product, dX through D3, ends the chain)
        if (pass > 0) lds_barrier();
        // ---- epilogue: bias / relu / mask / noise, bf16 into the output image, column sums ----
        char* oimg = lds + o_off;
        const int cip = wave * 32 + lc;                    // column inside the pass = column of the output image
        // element (row, cip) of the image sits at obase[sel(r)] + a compile-time offset: the swizzle term (row>>1)&7 of
        // row = 32 mi + (r&3) + 8 (r>>2) + 4 lh is  ((r>>1)&1) | lh<<1 | ((r>>2)&1)<<2, i.e. a lane part and 4 register cases
        int obase[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            obase[i] = (cip >> 6) * (ROWS * 128) + lh * 512 + (((((cip & 63) >> 3) ^ (lh << 1)) ^ ((i & 1) | ((i >> 1) << 2))) << 4) + (cip & 7) * 2;
        const float sig = (noisy && colvalid) ? op.sigma * NOISE_SCALE : 0.f;
        float s1 = 0.f;
        auto ostore = [&](int mi, int r, float o) {
#ifdef MRGAN_CH_NO_OSTORE       // timing experiment (compile-time: a run-time test per element perturbs the epilogue it measures)
            asm volatile("" :: "v"(o)); return;
#endif
            *(__bf16*)(oimg + obase[((r >> 1) & 1) | (((r >> 2) & 1) << 1)] + (mi * 32 + (r & 3) + 8 * (r >> 2)) * 128) = (__bf16)o;
        };
        if (a.ablate & CH_ABL_EPI) {
            if (acc[0][0] == 12345.678f) ostore(0, 0, s1);
        } else if constexpr (fwd) {
            const float bv = colvalid ? bias : 0.f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                i32x16 nzs = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (noisy) nzs = noise_block(rowhash[mi], (uint32_t)col >> 5, lane, hfrag);
                uint32_t mbits = 0u;
                // Rows >= rows_valid of a ragged block carry relu(bias) + noise instead of zeros.  That is harmless: every
                // product is row-local, copy_out never stores those rows, the head gives them zero dlogits -- only a
                // column sum must leave them out (below).  Masking them here would cost a compare per element, whose 32
                // lane masks hipcc keeps in SGPR pairs for the whole kernel (250 spilled SGPRs, ~2 k cycles per pass).
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = fmaxf(acc[mi][r] + bv, 0.f);
                    mbits |= min(__builtin_bit_cast(uint32_t, v), 1u) << r;
                    s1 += v;
                    acc[mi][r] = v;
                    ostore(mi, r, fmaf(sig, (float)nzs[r], v));       // sig = 0 without noise
                }
                mw[pass][mi] = mbits;
                if (mask && colin && row_blk + mi * 32 < a.rows)
                    mask[((long)((row_blk + mi * 32) >> 5) * op.ldm + col) * 2 + lh] = (uint16_t)mbits;
            }
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // rows >= rows_valid and padding columns arrive as exact zeros (zero A rows / zero weights).
                    // (a select, not a bit-AND on the accumulator element: hipcc 7.2 mis-folds that form)
                    const float av = acc[mi][r];
                    const float v = ((mw[pass][mi] >> r) & 1u) ? av : 0.f;
                    s1 += v;
                    ostore(mi, r, v);
                }
            }
        }
        if (op.cs) {
            if (fwd && rows_valid < ROWS) {                 // ragged block: the column sum again, without the padding rows
                s1 = 0.f;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s1 += (mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh < rows_valid) ? acc[mi][r] : 0.f;
            }
            s1 += __shfl_xor(s1, 32, 64);
            if (lh == 0 && col < op.ldcs) op.cs[((long)seg * nrb + rb) * op.ldcs + col] = s1;
        }
        CH_STAMP(6);                                       // epilogue math + image writes
        lds_barrier();                                     // the output image is complete
        if (op.out) {                                      // this pass's columns: copied out at the end of the next k-loop
            sm.cp_img = oimg; sm.cp_out = op.out + (long)seg * op.out_bs + (long)row_blk * op.ldo;
            sm.cp_ldo = op.ldo; sm.cp_col0 = pass * CH_PW; sm.cp_ncols = N;
        }
        CH_STAMP(7);                                       // image barrier + copy-out issue
    }
}

// relu-mask words of a dX product's output tile, from HBM (written by an earlier launch)
template <int MI>
__device__ __forceinline__ void load_mask_words(const ChainArgs& a, const ChainOp& op, uint32_t (&mw)[2][MI], int seg, int row_blk, int wave, int lc, int lh) {
    const uint16_t* mask = op.mask + (long)seg * op.mask_bs;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int col = pass * CH_PW + wave * 32 + lc;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            // branch-free (see chain_fmgrad): clamped address, result zeroed when out of range
            const bool ok = col < op.N && row_blk + mi * 32 < a.rows;
            const uint32_t w = mask[((long)((ok ? row_blk + mi * 32 : 0) >> 5) * op.ldm + (ok ? col : 0)) * 2 + lh];
            mw[pass][mi] = ok ? w : 0u;
        }
    }
}

#ifdef MRGAN_STAMPS
#define CH_ST_ARGS , st_acc, st_prev
#else
#define CH_ST_ARGS
#endif

// VARIANT: the three chains of one training step (chain.h).  The op list is fixed per variant, so the loop over products is
// unrolled at compile time: epilogue inputs are loaded once at the top (before the weight stream loads the memory
// pipeline), relu masks of products whose forward ran in this launch never leave registers, and no per-op descriptor
// reload sits between two products.
// MI: 32-row groups per block (2: 64 rows, the D sub-step's launch; 1: 32 rows, for launches that would leave most CUs idle)
template <int VARIANT, int MI>
__global__ __launch_bounds__(CH_THREADS) void chain_kernel(const ChainArgs a) {
    constexpr int ROWS = 32 * MI;
    static_assert(VARIANT != CH_V_DTAIL || MI == 2, "the loss head works on 64-row blocks");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63, lc = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nrb = (a.rows + ROWS - 1) / ROWS;
    const int seg = blockIdx.x / nrb, rb = blockIdx.x - seg * nrb;
    const int row_blk = rb * ROWS, rows_valid = min(ROWS, a.rows - row_blk);

#ifdef MRGAN_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
#endif
    // Keras iteration of this sub-step (noise key): loaded before the weight stream starts and pinned in an SGPR -- sunk to its
    // first use, the load would sit behind a vmcnt(0) that also drains the weight-tile DMA
    uint32_t iter = a.st ? __builtin_amdgcn_readfirstlane((int)a.st->iter) : 0u;
    asm volatile("" : "+s"(iter));
    // ---- epilogue inputs of every product: biases (forward) and the relu masks that come from HBM ----
    const int col0 = wave * 32 + lc;
    float bias[3] = {0.f, 0.f, 0.f};
    uint32_t mwA[2][MI], mwB[2][MI], mwC[2][MI];
#pragma unroll
    for (int p_ = 0; p_ < 2; ++p_)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) { mwA[p_][mi] = 0u; mwB[p_][mi] = 0u; mwC[p_][mi] = 0u; }
    if constexpr (VARIANT != CH_V_GBWD) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { const float bv = a.op[i].bias[min(col0, a.op[i].n_valid - 1)]; bias[i] = col0 < a.op[i].n_valid ? bv : 0.f; }
    }
    if constexpr (VARIANT == CH_V_DTAIL) load_mask_words<MI>(a, a.op[6], mwC, seg, row_blk, wave, lc, lh);      // dX through D3 needs D2's mask
    HeadInputs hin;
    if constexpr (VARIANT == CH_V_DTAIL) head_prefetch(a, hin, seg, row_blk, rows_valid, t);
    if constexpr (VARIANT == CH_V_GBWD) {
        load_mask_words<MI>(a, a.op[0], mwA, seg, row_blk, wave, lc, lh);
        load_mask_words<MI>(a, a.op[1], mwB, seg, row_blk, wave, lc, lh);
        load_mask_words<MI>(a, a.op[2], mwC, seg, row_blk, wave, lc, lh);
    }

    // ---- first A image ----
    const int a0_off = a.op[0].a_off;
    if constexpr (VARIANT != CH_V_GBWD) {
        const __bf16* src = a.a + (long)seg * a.a_bs;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)((long)a.rows * a.lda * 2), 0x00020000);
        const int lrow = lane >> 3, lp = lane & 7, nkt = a.a_cols / 64;
        // ROWS / 8 pieces of [8 rows][128 B] per k-tile, spread over the waves: rows >= a.rows arrive as zeros
        constexpr int PPT = ROWS / 8;
        for (int pce = wave; pce < nkt * PPT; pce += CH_THREADS / 64) {
            const int kt = pce / PPT, pr = pce - kt * PPT, R = pr * 8 + lrow;
            const int voff = (int)(((long)(row_blk + R) * a.lda + ((lp ^ ((R >> 1) & 7)) << 3)) * 2);
            glds16(rsA, lds + a0_off + kt * (ROWS * 128) + pr * 1024, voff, kt * 128);
        }
    }
    // ---- the weight-tile stream (per wave: its own 32 columns of every tile) ----
    Stream sm;
    constexpr int AHEAD = 2;       // tiles in flight (see chain_gemm)
    sm.gtile = 0; sm.cp_img = nullptr; sm.inflight = AHEAD;
    {
        BTile b0;
        btile_setup(b0, a.op[0].W, a.op[0].K, a.op[0].N, wave, lane);
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) issue_btile(b0, 0, i, lds + chain_ring(ROWS) + i * CH_STAGE_BYTES, wave);
    }
    if constexpr (VARIANT != CH_V_GBWD) {
        wait_vm(4 * AHEAD);        // this wave's pieces of the A image have landed (the weight-tile pieces are younger) ...
        __builtin_amdgcn_s_barrier();      // ... everyone's: the k-loops below run without workgroup barriers
        asm volatile("" ::: "memory");
    } else {
        chain_fmgrad<ROWS>(a, lds, a0_off, rb, row_blk, rows_valid, t CH_ST_ARGS);      // (ends with a workgroup barrier)
        sm.cp_img = lds + a0_off; sm.cp_out = (__bf16*)a.fm.dpre + (long)row_blk * a.fm.ldd;
        sm.cp_ldo = a.fm.ldd; sm.cp_col0 = 0; sm.cp_ncols = a.fm.feat;
    }
    CH_STAMP(0);                   // prologue (epilogue inputs, first tile issue, A image)

    const i32x4 hfrag = hadamard_frag(lane);
    // the op index goes through an opaque asm so that the descriptor's scalar loads happen at the product's start: hoisted to
    // the top of the kernel (what hipcc does with a constant index) seven descriptors overflow the SGPR file and every
    // pass pays ~2 k cycles of spill traffic
    auto opq = [](int i) { asm volatile("" : "+s"(i)); return i; };
    // ... and the descriptor is copied as a whole (wide scalar loads, one wait) instead of field by field at the points of use
    // NEXT: index of the product that follows (-1: none) -- its first weight tile is issued during this product's last k-tile
#define CH_GEMM(MODE, I, NEXT, BIAS, MW) do { const ChainOp op_ = a.op[opq(I)];                                                          \
        const int nx_ = opq(NEXT < 0 ? 0 : NEXT);                                                                                      \
        chain_gemm<MODE, MI>(a, op_, NEXT < 0 ? nullptr : a.op[nx_].W, a.op[nx_].K, a.op[nx_].N, lds, sm, BIAS, MW, seg, nrb, rb, row_blk,  \
                         rows_valid, iter, hfrag, t CH_ST_ARGS); } while (0)
    if constexpr (VARIANT == CH_V_DTAIL) {
        // D3 D4 D5 forward: the masks of D3 / D4 stay in registers for the way back
        uint32_t mw4[2][MI];
        CH_GEMM(CH_FWD_RELU, 0, 1, bias[0], mwB);
        CH_GEMM(CH_FWD_RELU, 1, 2, bias[1], mwA);
        CH_GEMM(CH_FWD_RELU, 2, 4, bias[2], mw4);
        CH_STAMP(1);               // (the feature image is complete: chain_gemm ended with the image barrier; the weight tile in
                                   //  flight lands in the ring, which the head does not touch)
        if constexpr (MI == 2) { if (!(a.ablate & CH_ABL_HEAD)) chain_head(a, lds, sm, hin, mw4, seg, rb, nrb, row_blk, rows_valid, t); }
        CH_STAMP(2);               // loss head
        CH_GEMM(CH_DX_RELU, 4, 5, 0.f, mwA);       // dX through D5 * relu'(D4)
        CH_GEMM(CH_DX_RELU, 5, 6, 0.f, mwB);       // dX through D4 * relu'(D3)
        CH_GEMM(CH_DX_RELU, 6, -1, 0.f, mwC);       // dX through D3 * relu'(D2)
    } else if constexpr (VARIANT == CH_V_GFWD) {
        uint32_t mwx[2][MI];
        CH_GEMM(CH_FWD_RELU, 0, 1, bias[0], mwx);
        CH_GEMM(CH_FWD_RELU, 1, 2, bias[1], mwx);
        CH_GEMM(CH_FWD_RELU, 2, -1, bias[2], mwx);
    } else {
        CH_GEMM(CH_DX_RELU, 0, 1, 0.f, mwA);
        CH_GEMM(CH_DX_RELU, 1, 2, 0.f, mwB);
        CH_GEMM(CH_DX_RELU, 2, -1, 0.f, mwC);
    }
#undef CH_GEMM
    if (!(a.ablate & CH_ABL_COPY)) flush_copy<ROWS>(sm, rows_valid, t);      // the last image
#ifdef MRGAN_STAMPS
    if (a.stamps && t == 0)
        for (int i = 0; i < 8; ++i) a.stamps[(long)blockIdx.x * 8 + i] = st_acc[i];
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Stand-alone loss head for wide feature layers (chain.h: HeadWideArgs).  The same three MFMA products as chain_head, with the
// feature dimension walked in chunks of 256 columns: the block's 64 rows x 256 features arrive by LDS-DMA into one of two
// images (the next chunk is in flight while the current one is consumed), wave w owns features [32 w, 32 w + 32) of a chunk.
//   pass 1 (chunks ascending): logits partial products, accumulated over ALL chunks in the wave's registers
//   row phase (wave 0): losses, error, dlogits as bf16 addends
//   pass 2 (chunks descending: the last chunk is still resident): dL/d(pre5) of the chunk -> output images -> HBM, dW6^T
// W6 enters as bf16 addends prepared once per launch by w6_split_kernel (class-major for the logits' B operand, row-major for
// dL/d(pre5)'s), so a fragment is one 16-byte load from a 200 KB array that stays in L2.
// Q8: dL/d(pre5) leaves as the two e5m2 images the fp8 products read (row-major and transposed), packed from the accumulators:
// a lane's four consecutive rows of one column are one dword of the transposed image, the row-major dword comes from a 4 x 4
// byte transpose inside the lane quad (gemm.h does the same in the fp8 epilogues); both images are assembled in LDS and leave
// as 16-byte stores.  Otherwise dL/d(pre5) leaves as bf16 through the chain's image + copy_out.
// ------------------------------------------------------------------------------------------------------------------
constexpr int HW_FIMG = CH_ROWS * CH_PW * 2;                  // 32 KiB per feature image
constexpr int HW_TPITCH = CH_ROWS + 16, HW_RPITCH = CH_PW + 16;
constexpr int HW_X = 2 * HW_FIMG, HW_X_BYTES = 40 * 1024;     // pass 1: logits partials; pass 2: output image(s)
constexpr int HW_SMALL = HW_X + HW_X_BYTES;
constexpr int HW_LDS = HW_SMALL + 2 * 3 * CH_ROWS * KMAX * 2 + (3 + KMAX) * CH_ROWS * 4;
static_assert(CH_PW * HW_TPITCH + CH_ROWS * HW_RPITCH <= HW_X_BYTES && 8 * CH_ROWS * KMAX * 4 <= HW_X_BYTES, "head_wide LDS map");

__global__ __launch_bounds__(256) void w6_split_kernel(const float* w, int ldw, int feat, int feat_valid, int classes, __bf16* w6c, __bf16* w6r) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= feat) return;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        const float v = (j < feat_valid && c < classes) ? w[(long)j * ldw + c] : 0.f;
        __bf16 p0, p1, p2;
        split3(v, p0, p1, p2);
        w6c[(0L * KMAX + c) * feat + j] = p0; w6c[(1L * KMAX + c) * feat + j] = p1; w6c[(2L * KMAX + c) * feat + j] = p2;
        w6r[(0L * feat + j) * KMAX + c] = p0; w6r[(1L * feat + j) * KMAX + c] = p1; w6r[(2L * feat + j) * KMAX + c] = p2;
    }
}

template <bool Q8>
__global__ __launch_bounds__(CH_THREADS) void head_wide_kernel(const HeadWideArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const HeadArgs& h = a.h;
    char* xreg = lds + HW_X;
    __bf16* dl_rc = (__bf16*)(lds + HW_SMALL);                // [3][CH_ROWS][KMAX]   dlogits addends, row-major
    __bf16* dl_t = dl_rc + 3 * CH_ROWS * KMAX;                // [3][KMAX][CH_ROWS]   ... class-major
    float* red = (float*)(dl_t + 3 * KMAX * CH_ROWS);         // [3 + KMAX][CH_ROWS]
    const int t = threadIdx.x, lane = t & 63, lc = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int seg = blockIdx.y, rb = blockIdx.x, nrb = gridDim.x, kind = h.seg_kind[seg];
    const int row_blk = rb * CH_ROWS, rows_valid = min(CH_ROWS, h.rows - row_blk), blk = seg * nrb + rb;
    const int nch = h.feat / CH_PW;
    const bf16x8 zero8 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};

    // ---- feature chunks by LDS-DMA: [4 k-tiles][64 rows][64 k] with the chain's swizzle; rows >= h.rows arrive as zeros ----
    const __bf16* fseg = (const __bf16*)h.f + (long)seg * h.f_bs;
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void*)fseg, 0, (int)((long)h.rows * h.ldf * 2), 0x00020000);
    int fvoff[4], fdst[4];
    {
        const int lrow = lane >> 3, lp = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pce = wave + 8 * i, kt = pce >> 3, pr = pce & 7, R = pr * 8 + lrow;
            fvoff[i] = (int)(((long)(row_blk + R) * h.ldf + kt * 64 + ((lp ^ ((R >> 1) & 7)) << 3)) * 2);
            fdst[i] = kt * (CH_ROWS * 128) + pr * 1024;
        }
    }
    auto issue_chunk = [&](int c) {
        char* img = lds + (c & 1) * HW_FIMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(rsF, img + fdst[i], fvoff[i], c * (CH_PW * 2));
    };
    // B fragments of the logits product for chunk c: lane <-> (class lc, features 16 (2 wave + u) + 8 lh .. + 7 of the chunk)
    auto load_w6c = [&](int c, bf16x8 (&fb)[2][3]) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const bf16x8 v = *(const bf16x8*)(a.w6c + ((long)p * KMAX + (lc & (KMAX - 1))) * h.feat + c * CH_PW + 16 * (2 * wave + u) + 8 * lh);
                fb[u][p] = lc < KMAX ? v : zero8;             // columns 8 .. 31 of the product are padding
            }
    };

    // =========================== pass 1: logits ===========================
    f32x16 lacc[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) lacc[mi][r] = 0.f;
    bf16x8 fbn[2][3];
    issue_chunk(0);
    load_w6c(0, fbn);
    for (int c = 0; c < nch; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of chunk c and its W6 fragments
        __builtin_amdgcn_s_barrier();                         // ... everyone's pieces; everyone is done with chunk c - 1
        asm volatile("" ::: "memory");
        bf16x8 fb[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[u][p] = fbn[u][p];
        if (c + 1 < nch) { issue_chunk(c + 1); load_w6c(c + 1, fbn); }
        const char* fimg = lds + (c & 1) * HW_FIMG;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kg = 2 * wave + u;
            const char* As = fimg + (kg >> 2) * (CH_ROWS * 128);
            const int ch = (kg & 3) * 2 + lh;
            const bf16x8 fa0 = *(const bf16x8*)(As + kc_off(lc, ch)), fa1 = *(const bf16x8*)(As + kc_off(32 + lc, ch));
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                lacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb[u][p], lacc[0], 0, 0, 0);
                lacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb[u][p], lacc[1], 0, 0, 0);
            }
        }
    }
    float* lpart = (float*)xreg;                              // [8 waves][CH_ROWS][KMAX]
    if (lc < KMAX) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) lpart[(wave * CH_ROWS + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * KMAX + lc] = lacc[mi][r];
    }
    lds_barrier();
    // the chunk before the last one is needed next (pass 2 walks downwards): its image is free now
    if (nch > 1) issue_chunk(nch - 2);

    // =========================== row phase (wave 0: lane <-> row), as chain_head step 3 ===========================
    float* part_row = h.part + (long)blk * h.part_stride;
    if (wave == 0) {
        const int r = lane;
        float l[KMAX];
#pragma unroll
        for (int c = 0; c < KMAX; ++c) l[c] = 0.f;
#pragma unroll
        for (int w = 0; w < CH_THREADS / 64; ++w) {
            const f32x4 p0 = *(const f32x4*)(lpart + (w * CH_ROWS + r) * KMAX), p1 = *(const f32x4*)(lpart + (w * CH_ROWS + r) * KMAX + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) { l[c] += p0[c]; l[4 + c] += p1[c]; }
        }
        const bool rowvalid = r < rows_valid;
        float mx = -3.0e38f;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
            if (c < h.classes) { l[c] += h.b[c]; mx = fmaxf(mx, l[c]); }
        }
        int am = 0;
        float se = 0.f, p[KMAX];
#pragma unroll
        for (int c = KMAX - 1; c >= 0; --c) {
            p[c] = (c < h.classes) ? expf(l[c] - mx) : 0.f;
            se += p[c];
            if (c < h.classes && l[c] == mx) am = c;          // ties -> first index (theano argmax)
        }
        const float lse = mx + logf(se);
        const float inv_se = 1.0f / se;
        float loss0 = 0.f, loss1 = 0.f, err = 0.f;
        float dl[KMAX];
#pragma unroll
        for (int c = 0; c < KMAX; ++c) dl[c] = 0.f;
        if (rowvalid) {
            if (kind == HEAD_LAB) {
                const long lo = h.labels_stream ? (long)h.st->batch * h.rows : 0;
                const int y = h.labels[lo + row_blk + r];
                err = (am != y) ? 1.f : 0.f;
                float ly = 0.f;
#pragma unroll
                for (int c = 0; c < KMAX; ++c) {
                    if (c == y) ly = l[c];
                    dl[c] = (p[c] * inv_se - (c == y ? 1.f : 0.f)) * h.inv_count;
                }
                loss0 = lse - ly;
            } else {
                const float sg = sigmoid_f(lse), sp = softplus_f(lse);
                const float k = 0.5f * h.inv_count * h.unl_weight * (kind == HEAD_UNL ? (sg - 1.0f) : sg);
                loss1 = (kind == HEAD_UNL) ? 0.5f * (sp - lse) : 0.5f * sp;
#pragma unroll
                for (int c = 0; c < KMAX; ++c) dl[c] = k * p[c] * inv_se;
            }
            if (h.logits) {
                float* lp = h.logits + (long)seg * h.logits_bs + (long)(row_blk + r) * KMAX;
#pragma unroll
                for (int c = 0; c < KMAX; ++c) lp[c] = (c < h.classes) ? l[c] : 0.f;
            }
        }
        bf16x8 d3[3];
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
            __bf16 p0, p1, p2;
            split3(dl[c], p0, p1, p2);
            d3[0][c] = p0; d3[1][c] = p1; d3[2][c] = p2;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            *(bf16x8*)(dl_rc + (q * CH_ROWS + r) * KMAX) = d3[q];
#pragma unroll
            for (int c = 0; c < KMAX; ++c) dl_t[(q * KMAX + c) * CH_ROWS + r] = d3[q][c];
        }
        red[0 * CH_ROWS + r] = loss0; red[1 * CH_ROWS + r] = loss1; red[2 * CH_ROWS + r] = err;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) red[(3 + c) * CH_ROWS + r] = dl[c];
    }
    lds_barrier();
    if (wave == CH_THREADS / 64 - 1 && lane < 3 + KMAX) {     // the eleven sums over the block's rows that leave it
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CH_ROWS / 4; ++i) {
            const f32x4 v = *(const f32x4*)(red + lane * CH_ROWS + 4 * i);
            s4[i & 3] += (v[0] + v[1]) + (v[2] + v[3]);
        }
        const float tot = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        if (lane < 3) h.loss_part[blk * 4 + lane] = tot;
        else part_row[h.off_db + lane - 3] = tot;
        if (lane == 0) h.loss_part[blk * 4 + 3] = 0.f;
    }

    // =========================== pass 2: dL/d(pre5) and dW6^T, chunk by chunk ===========================
    // A operands that do not depend on the chunk, in registers for the whole pass
    bf16x8 da[2][3], dt[4][3];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            da[mi][q] = *(const bf16x8*)(dl_rc + (q * CH_ROWS + mi * 32 + lc) * KMAX);
            if (lh) da[mi][q] = zero8;                        // k = 8 .. 15: padding
        }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            dt[ks][q] = *(const bf16x8*)(dl_t + (q * KMAX + (lc & (KMAX - 1))) * CH_ROWS + 16 * ks + 8 * lh);
            if (lc >= KMAX) dt[ks][q] = zero8;
        }
    const uint16_t* mseg = a.mask + (long)seg * a.mask_bs;
    // per chunk: the W6 rows of this lane's feature as bf16 addends (B operand, k = class) and the relu-mask words of its column
    auto load_chunk_inputs = [&](int c, bf16x8 (&bw)[3], uint32_t (&mw)[2]) {
        const int col = c * CH_PW + wave * 32 + lc;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const bf16x8 v = *(const bf16x8*)(a.w6r + ((long)q * h.feat + col) * KMAX);
            bw[q] = lh ? zero8 : v;                           // lh = 1: k = 8 .. 15, zeros
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const bool ok = row_blk + mi * 32 < h.rows;
            const uint32_t w = mseg[((long)((ok ? row_blk + mi * 32 : 0) >> 5) * a.ldm + col) * 2 + lh];
            mw[mi] = ok ? w : 0u;
        }
    };
    const float q8s = Q8 ? h.q8_slot->scale : 1.f;
    float q8_amax = 0.f;
    bf16x8 bwn[3];
    uint32_t mwn[2];
    load_chunk_inputs(nch - 1, bwn, mwn);
    for (int c = nch - 1; c >= 0; --c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // chunk c (this wave's pieces), its inputs; the previous copy-out
        __builtin_amdgcn_s_barrier();                         // ... everyone's: the output region and chunk c + 1's image are free
        asm volatile("" ::: "memory");
        bf16x8 bw[3];
        uint32_t mw[2];
#pragma unroll
        for (int q = 0; q < 3; ++q) bw[q] = bwn[q];
        mw[0] = mwn[0]; mw[1] = mwn[1];
        if (c >= 1) load_chunk_inputs(c - 1, bwn, mwn);
        if (c >= 1 && c != nch - 1) issue_chunk(c - 1);       // (chunk nch - 2 was issued before the row phase)
        const char* fimg = lds + (c & 1) * HW_FIMG;
        const int c0 = c * CH_PW, cip = wave * 32 + lc;
        // ---- dL/d(pre5) = (dlogits W6^T) * relu'(pre5): one 16-deep k-step x 6 addend pairs ----
        {
            f32x16 acc[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
            constexpr int PA[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
            for (int i = 5; i >= 0; --i)                      // smallest terms first
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[mi][PA[i]], bw[PB[i]], acc[mi], 0, 0, 0);
            float s1 = 0.f;
            if constexpr (Q8) {
                unsigned char* timg = (unsigned char*)xreg;
                unsigned char* rimg = timg + CH_PW * HW_TPITCH;
                const int kq = lane & 3;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float o4[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int r = 4 * g + j;
                            const float av = acc[mi][r];
                            o4[j] = ((mw[mi] >> r) & 1u) ? av : 0.f;              // (a select: see chain_gemm)
                            s1 += o4[j];
                            q8_amax = fmaxf(q8_amax, fabsf(o4[j]));
                        }
                        const uint32_t w = fp8_pack4<FP8_E5M2>(o4[0], o4[1], o4[2], o4[3], q8s);
                        const int rl = mi * 32 + 8 * g + 4 * lh;                    // rows rl .. rl + 3 of column cip
                        *(uint32_t*)(timg + cip * HW_TPITCH + rl) = w;
                        *(uint32_t*)(rimg + (rl + kq) * HW_RPITCH + (cip - kq)) = quad_byte_transpose(w);
                    }
            } else {
                char* oimg = xreg;
                int obase[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    obase[i] = (cip >> 6) * (CH_ROWS * 128) + lh * 512 + (((((cip & 63) >> 3) ^ (lh << 1)) ^ ((i & 1) | ((i >> 1) << 2))) << 4) + (cip & 7) * 2;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float av = acc[mi][r];
                        const float v = ((mw[mi] >> r) & 1u) ? av : 0.f;
                        s1 += v;
                        *(__bf16*)(oimg + obase[((r >> 1) & 1) | (((r >> 2) & 1) << 1)] + (mi * 32 + (r & 3) + 8 * (r >> 2)) * 128) = (__bf16)v;
                    }
            }
            s1 += __shfl_xor(s1, 32, 64);
            if (lh == 0) part_row[h.off_dbf + c0 + cip] = s1;                   // bias gradient of the feature layer
        }
        // ---- dW6^T [class][feature] = dlogits^T F for this wave's 32 features of the chunk ----
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
            const int f0 = wave * 32 + (g4 & 1) * 16 + 4 * pp;
#pragma unroll
            for (int ks = 0; ks < CH_ROWS / 16; ++ks) {
                const int m0 = ks * 16 + (g4 >> 1) * 8 + q;
                const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(fimg + act_off(m0, f0)));
                const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(fimg + act_off(m0 + 4, f0)));
                const bf16x8 fb = __builtin_bit_cast(bf16x8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int p = 2; p >= 0; --p) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dt[ks][p], fb, acc, 0, 0, 0);
            }
            *(f32x4*)(part_row + (long)(c0 + cip) * KMAX + 4 * lh) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
        }
        lds_barrier();                                        // the output image(s) of the chunk are complete
        if constexpr (Q8) {
            const unsigned char* timg = (const unsigned char*)xreg;
            const unsigned char* rimg = timg + CH_PW * HW_TPITCH;
            unsigned char* q8t = h.q8t ? h.q8t + (long)seg * h.q8t_bs : nullptr;
            unsigned char* q8 = h.q8 ? h.q8 + (long)seg * h.q8_bs : nullptr;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int qi = t + CH_THREADS * u;
                // transposed copy: column (row of q8t) x 16 rows; rows >= h.rows of the block are zero bytes (zero dlogits)
                if (q8t) *(u32x4*)(q8t + (long)(c0 + (qi >> 2)) * h.ldq8t + row_blk + 16 * (qi & 3)) = *(const u32x4*)(timg + (qi >> 2) * HW_TPITCH + 16 * (qi & 3));
                if (q8 && (qi >> 4) < rows_valid) *(u32x4*)(q8 + (long)(row_blk + (qi >> 4)) * h.ldq8 + c0 + 16 * (qi & 15)) = *(const u32x4*)(rimg + (qi >> 4) * HW_RPITCH + 16 * (qi & 15));
            }
        } else {
            copy_out<CH_ROWS>(xreg, (__bf16*)h.dpre + (long)seg * h.dpre_bs + (long)row_blk * h.ldd, h.ldd, c0, h.feat, rows_valid, t);
        }
    }
    if constexpr (Q8) fp8_amax_commit(h.q8_slot, q8_amax);
}

}  // namespace

// the bf16 addends of W6 for launch_head_wide (once per D sub-step: W6 changes with every Adam update)
int launch_w6_split(const HeadWideArgs& a, hipStream_t s) {
    const HeadArgs& h = a.h;
    if (!a.w6c || !a.w6r || !h.w) return -3;
    MRGAN_LAUNCH(w6_split_kernel, dim3((h.feat + 255) / 256), dim3(256), 0, s, h.w, h.ldw, h.feat, h.feat_valid, h.classes, a.w6c, a.w6r);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_head_wide(const HeadWideArgs& a, hipStream_t s) {
    const HeadArgs& h = a.h;
    if ((h.feat % CH_PW) != 0 || h.classes > KMAX || !a.mask || !a.w6c || !a.w6r || !h.part || !h.loss_part) return -3;
    for (int i = 0; i < h.nseg; ++i)
        if (h.seg_kind[i] != HEAD_LAB && h.seg_kind[i] != HEAD_UNL && h.seg_kind[i] != HEAD_FAKE) return -3;
    if ((long)h.rows * h.ldf * 2 >= (1L << 31)) return -3;
    const bool q8 = h.q8_slot != nullptr;
    if (q8 ? !(h.q8 || h.q8t) : !h.dpre) return -3;
    // (the dynamic-LDS limit of both instantiations is raised by chain_init_attributes, outside any stream capture)
    const dim3 grid((h.rows + CH_ROWS - 1) / CH_ROWS, h.nseg), block(CH_THREADS);
    if (q8) MRGAN_LAUNCH((head_wide_kernel<true>), grid, block, HW_LDS, s, a);
    else MRGAN_LAUNCH((head_wide_kernel<false>), grid, block, HW_LDS, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int chain_init_attributes() {
    hipError_t e = hipFuncSetAttribute((const void*)chain_kernel<CH_V_DTAIL, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes(64));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)chain_kernel<CH_V_GFWD, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes(64));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)chain_kernel<CH_V_GBWD, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes(64));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)chain_kernel<CH_V_GFWD, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes(32));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)chain_kernel<CH_V_GBWD, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, chain_lds_bytes(32));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)head_wide_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, HW_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)head_wide_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, HW_LDS);
    return e == hipSuccess ? 0 : -2;
}

int launch_chain(const ChainArgs& a, hipStream_t s) {
    // the three fixed op lists (chain.h)
    static const int want[3][CH_MAX_OPS] = {{CH_OP_GEMM, CH_OP_GEMM, CH_OP_GEMM, CH_OP_HEAD, CH_OP_GEMM, CH_OP_GEMM, CH_OP_GEMM, -1},
                                            {CH_OP_GEMM, CH_OP_GEMM, CH_OP_GEMM, -1, -1, -1, -1, -1},
                                            {CH_OP_GEMM, CH_OP_GEMM, CH_OP_GEMM, -1, -1, -1, -1, -1}};
    static const int nops[3] = {7, 3, 3};
    if (a.variant < 0 || a.variant > 2 || a.nops != nops[a.variant]) return -3;
    if (a.block_rows != 64 && !(a.block_rows == 32 && a.variant != CH_V_DTAIL)) return -3;
    for (int i = 0; i < a.nops; ++i) {
        const ChainOp& op = a.op[i];
        if (op.kind != want[a.variant][i]) return -3;
        if (op.kind != CH_OP_GEMM) continue;
        const bool fwd = a.variant == CH_V_GFWD || (a.variant == CH_V_DTAIL && i < 3);
        if (op.mode != (fwd ? CH_FWD_RELU : CH_DX_RELU)) return -3;
        if ((op.K % 64) || (op.N % 64) || op.K > CH_KMAX || op.K < 128 || op.N > 2 * CH_PW || !op.W) return -3;
        if (fwd && op.N > CH_PW) return -3;                    // a forward output is the next product's A image
        if (!fwd && !op.mask) return -3;
        if (fwd && !op.bias) return -3;
        if ((long)op.N * op.K * 2 >= (1L << 31)) return -3;
    }
    if (a.variant != CH_V_GBWD && ((a.a_cols % 64) || a.a_cols > CH_KMAX || a.a_cols != a.op[0].K || (long)a.rows * a.lda * 2 >= (1L << 31))) return -3;
    const int nrb = (a.rows + a.block_rows - 1) / a.block_rows;
    const dim3 grid(nrb * a.nseg), block(CH_THREADS);
    const int lds = chain_lds_bytes(a.block_rows);
    if (a.variant == CH_V_DTAIL) MRGAN_LAUNCH((chain_kernel<CH_V_DTAIL, 2>), grid, block, lds, s, a);
    else if (a.variant == CH_V_GFWD && a.block_rows == 64) MRGAN_LAUNCH((chain_kernel<CH_V_GFWD, 2>), grid, block, lds, s, a);
    else if (a.variant == CH_V_GFWD) MRGAN_LAUNCH((chain_kernel<CH_V_GFWD, 1>), grid, block, lds, s, a);
    else if (a.block_rows == 64) MRGAN_LAUNCH((chain_kernel<CH_V_GBWD, 2>), grid, block, lds, s, a);
    else MRGAN_LAUNCH((chain_kernel<CH_V_GBWD, 1>), grid, block, lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace mrgan
