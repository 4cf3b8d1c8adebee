// fp8 products on the matrix cores (OCP e4m3 / e5m2 operands, fp32 accumulate): the three products of a dense layer for
// the wide stack BASELINE configs[4] asks for (hidden 4096 x 5, batch 8192).
//
//   FWD   Y  = act(X W + b) (+ noise)     A = X    e4m3 [rows][K]     Bt = W^T  e4m3 [N][K]
//   DX    dX = (dY W^T) * relu'           A = dY   e5m2 [rows][N]     Bt = W    e4m3 [K][N]
//   SLAB  dW = X^T dY                     A = X^T  e4m3 [K][rows]     Bt = dY^T e5m2 [N][rows]    (no split-K: one fp32 slab)
// All three are "A Bt^T" with the reduction index contiguous in both operands, so ONE kernel serves them: the producers
// write every activation / gradient twice, row-major for the next FWD / DX and transposed for the SLAB product.
//
// MFMA: v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales -- the only fp8 form that runs at 2x the bf16 rate on
// gfx950 (MI355X_MICROARCH.md, Matrix cores; the non-scaled 32x32x16_fp8_fp8 form issues at the bf16 rate).  Per-tensor
// power-of-two scales (Fp8Slot, delayed scaling) are undone on the accumulator.  Lane l holds k = 32 (l >> 5) .. + 31 of row
// l & 31: two ds_read_b128 per fragment from the same XOR-swizzled [rows][128 B] LDS image as the bf16 kernels, filled by
// LDS-DMA (buffer_load ... lds) into a 2-stage ring, one s_barrier per k-tile (128 reduction elements), persistent blocks in
// XCD-aware tile order, 128x128 (4 waves) or 256x256 (8 waves of 128x64) blocks.
//
// Epilogue: the shared one of gemm.h (bias / relu / mask / GaussianNoise / column sums).  With OUT8 it packs the fp8 output
// straight from the fp32 accumulators (one rounding; the oracle's fp8 mirror does the same) into a transposed and a
// row-major LDS byte image -- a lane's four consecutive rows of one column are one dword of the transposed image, the
// row-major dword comes from a 4 x 4 byte transpose inside the lane quad -- which copy_tile writes out with 16-byte stores;
// max |v| goes to the output slot.  Without OUT8 the output is the bf16 tile of the bf16 kernels.
#include <algorithm>
#include <string>

#include "gemm.h"

namespace mrgan {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int BKB = 128;                        // reduction BYTES (= fp8 elements) per k-tile

__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int xcd_tile(int bid, int nt) { return (nt & 7) == 0 ? (bid & 7) * (nt >> 3) + (bid >> 3) : bid; }
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_dst, 16, voff, soff, 0, 0);
}

template <int FMT>
__device__ __forceinline__ u32x4 pack16(const float (&v)[16], float qs) {
    return (u32x4){fp8_pack4<FMT>(v[0], v[1], v[2], v[3], qs), fp8_pack4<FMT>(v[4], v[5], v[6], v[7], qs),
                   fp8_pack4<FMT>(v[8], v[9], v[10], v[11], qs), fp8_pack4<FMT>(v[12], v[13], v[14], v[15], qs)};
}
// the two fp8 images the shared epilogue (Q8 mode) left in LDS -> global memory, 16 bytes per lane
template <int BM, int BNT, int NT>
__device__ __forceinline__ void copy_tile(const unsigned char* tr, const Epi& e, int batch, int row_blk, int col_blk, int M, int N) {
    constexpr int PR = BNT + 16, PT = BM + 16;
    const unsigned char* tt = tr + BM * PR;
    if (e.q8) {
        unsigned char* q8 = (unsigned char*)e.q8 + (long)batch * e.q8_bs;
        constexpr int CPR = BNT / 16;
        for (int c = threadIdx.x; c < BM * CPR; c += NT) {
            const int r = c / CPR, cc = (c - r * CPR) * 16;
            if (row_blk + r < M && col_blk + cc < N)
                *(u32x4*)(q8 + (long)(row_blk + r) * e.ldq8 + col_blk + cc) = *(const u32x4*)(tr + r * PR + cc);
        }
    }
    if (e.q8t) {
        // 16 lanes cover the BM rows of one column: BM contiguous bytes of the transposed row.  Row groups that start at or
        // beyond M keep the zeros of mrgan_create (the weight-gradient product reduces over all S rows of a segment).
        unsigned char* q8t = (unsigned char*)e.q8t + (long)batch * e.q8t_bs;
        constexpr int RG = BM / 16;
        for (int c = threadIdx.x; c < BNT * RG; c += NT) {
            const int col = c / RG, r0 = (c - col * RG) * 16;
            if (row_blk + r0 < M && col_blk + col < N)
                *(u32x4*)(q8t + (long)(col_blk + col) * e.ldq8t + row_blk + r0) = *(const u32x4*)(tt + col * PT + r0);
        }
    }
}

// operand formats per product: cbsz (A) / blgp (B) of the scaled MFMA, 0 = e4m3, 1 = e5m2
template <int EPI> struct Fp8Fmt { static constexpr int A = EPI == EPI_DX ? 1 : 0, B = EPI == EPI_SLAB ? 1 : 0; };

// OUT8: the output leaves as fp8 copies (e.q8 / e.q8t) packed from the accumulators; otherwise as the bf16 tile (e.out)
template <int EPI, int BM, int BNT, int WM, int WN, int VAR, bool OUT8>
__global__ __launch_bounds__(64 * WM * WN) void gemm_fp8_kc_kernel(const GemmArgs g) {
    constexpr int NW = WM * WN;
    constexpr int MR = BM / WM / 32, NR = BNT / WN / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BNT * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BNT / 8 / NW;
    constexpr int FMT_A = Fp8Fmt<EPI>::A, FMT_B = Fp8Fmt<EPI>::B;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane >> 3, lp = lane & 7, lr = lane & 31, lh = lane >> 5;
    const int ntn = (g.N + BNT - 1) / BNT, ntm = (g.M + BM - 1) / BM;
    // split-K (weight gradients of narrow layers only: too few output tiles to fill the chip): nbatch == 1, the reduction
    // range is cut into g.splits chunks of g.kchunk elements, chunk s goes to fp32 slab s
    const int nsplit = EPI == EPI_SLAB ? g.splits : 1;
    const int ntiles = ntn * ntm * g.nbatch * nsplit, nk = (EPI == EPI_SLAB ? g.kchunk : g.K) / BKB;
    // undo the operand scales on the accumulator
    const float us = g.e.qa ? g.e.qa->inv_scale * g.e.qb->inv_scale : g.e.acc_scale;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int tidx = xcd_tile(tl, ntiles);
        const int bs = tidx / (ntn * ntm), rem = tidx - bs * (ntn * ntm);
        const int batch = EPI == EPI_SLAB ? 0 : bs, split = EPI == EPI_SLAB ? bs : 0;
        const int k0 = split * (EPI == EPI_SLAB ? g.kchunk : 0);
        // consecutive tiles (= the CUs of one XCD at any moment) form patches of 4 tile rows x 8 tile columns instead of
        // 2 x 16: a third less distinct operand data per k-step has to enter that XCD's L2
        int tile_m, tile_n;
        if ((ntm & 3) == 0) {
            const int grp = rem / (4 * ntn), in = rem - grp * (4 * ntn);
            tile_m = grp * 4 + (in & 3); tile_n = in >> 2;
        } else { tile_m = rem / ntn; tile_n = rem - tile_m * ntn; }
        const int row_blk = tile_m * BM, col_blk = tile_n * BNT;
        const char* Ab = (const char*)g.A + (long)batch * g.a_bs;
        const char* Bb = (const char*)g.B + (long)batch * g.b_bs;
        // rows >= M (A) / >= N (Bt) fall outside the descriptors and arrive as zeros
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)((long)g.M * g.a_si), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)((long)g.N * g.b_sj), 0x00020000);
        int voffA[A_INSTR], voffB[B_INSTR];
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const int R = (wave * A_INSTR + i) * 8 + lrow;
            voffA[i] = (int)((long)(row_blk + R) * g.a_si + ((lp ^ ((R >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const int R = (wave * B_INSTR + i) * 8 + lrow;
            voffB[i] = (int)((long)(col_blk + R) * g.b_sj + ((lp ^ ((R >> 1) & 7)) << 4));
        }
        // one LDS-DMA piece = 1 KiB of a wave's share of the next k-tile.  A piece costs its wave 60-185 issue cycles
        // (MI355X_MICROARCH.md), so the pieces of tile kt + 1 are spread over the MFMA steps of tile kt instead of being
        // issued together behind the barrier, where all eight waves would sit in them with the matrix pipe idle.
        constexpr int NPIECE = A_INSTR + B_INSTR;
        auto issue_piece = [&](int kb, int buf, int p) {
            if (p < A_INSTR) glds16(rsA, lds + buf * STAGE + (wave * A_INSTR + p) * 1024, voffA[p], kb);
            else glds16(rsB, lds + buf * STAGE + A_BYTES + (wave * B_INSTR + (p - A_INSTR)) * 1024, voffB[p - A_INSTR], kb);
        };
        if (nk > 0) {
#pragma unroll
            for (int p = 0; p < NPIECE; ++p) issue_piece(k0, 0, p);
        }
        f32x16 acc[MR][NR];
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        EpiPrefetch<MR, NR> pf;
        epilogue_prefetch<__bf16, EPI, MR, NR, VAR>(pf, g, batch, row_blk, col_blk, wm, wn, lane);
        int buf = 0;
#ifdef MRGAN_STAMPS
        const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long st_wait = 0;
#endif
        for (int kt = 0; kt < nk; ++kt) {
#ifdef MRGAN_STAMPS
            const unsigned long long st_a = __builtin_amdgcn_s_memtime();
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of k-tile kt have landed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#ifdef MRGAN_STAMPS
            st_wait += __builtin_amdgcn_s_memtime() - st_a;
#endif
            const bool more = kt + 1 < nk;
            const int nbuf = buf ^ 1;
            const char* As = lds + buf * STAGE;
            const char* Bs = As + A_BYTES;
            buf ^= 1;
            // two k-steps of 64 per tile: lane (r = lane & 31, h = lane >> 5) holds elements k = 32 h .. 32 h + 31 of row r.
            // Software pipeline inside the wave: the A fragment of the next (k-step, row block) is read while the MFMAs of
            // the current one run; the B fragments of both k-steps are read up front.
            auto frag = [&](const char* base, int row, int ks) -> i32x8 {
                const i32x4 lo = *(const i32x4*)(base + kc_off(row, ks * 4 + lh * 2)), hi = *(const i32x4*)(base + kc_off(row, ks * 4 + lh * 2 + 1));
                return (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            i32x8 b[2][NR];
#pragma unroll
            for (int ni = 0; ni < NR; ++ni) b[0][ni] = frag(Bs, (wn * NR + ni) * 32 + lr, 0);
            i32x8 a_cur = frag(As, (wm * MR) * 32 + lr, 0);
#pragma unroll
            for (int step = 0; step < 2 * MR; ++step) {
                const int ks = step / MR, mi = step % MR;
                i32x8 a_next = a_cur;
                if (step + 1 < 2 * MR) a_next = frag(As, (wm * MR + (step + 1) % MR) * 32 + lr, (step + 1) / MR);
                if (step == 0) {
#pragma unroll
                    for (int ni = 0; ni < NR; ++ni) b[1][ni] = frag(Bs, (wn * NR + ni) * 32 + lr, 1);
                }
                __builtin_amdgcn_sched_barrier(0);       // keep the reads above ahead of these MFMAs (hipcc otherwise sinks them
                                                         // into one register set and waits lgkmcnt(0) before every pair)
#pragma unroll
                for (int ni = 0; ni < NR; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_cur, b[ks][ni], acc[mi][ni], FMT_A, FMT_B, 0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    // all pieces within the first half of the steps: the last one still has half a k-tile to land
                    constexpr int STEPS = MR, PPS = (NPIECE + STEPS - 1) / STEPS;
#pragma unroll
                    for (int q = 0; q < PPS; ++q)
                        if (step * PPS + q < NPIECE) issue_piece(k0 + (kt + 1) * BKB, nbuf, step * PPS + q);
                }
                __builtin_amdgcn_sched_barrier(0);
                a_cur = a_next;
            }
        }
#ifdef MRGAN_STAMPS
        if (g.e.cs2 && g.e.cs_mode == CS_NONE && t == 0) {       // diagnostic build: cs2 carries the stamp buffer
            unsigned long long* sp = (unsigned long long*)g.e.cs2 + (size_t)blockIdx.x * 4;
            sp[0] += __builtin_amdgcn_s_memtime() - st_t0; sp[1] += st_wait;
            sp[2] += __builtin_amdgcn_s_memrealtime() - st_r0; sp[3] += 1;
        }
#endif
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= us;
        __syncthreads();
        if constexpr (EPI == EPI_SLAB) {
            epilogue<__bf16, EPI_SLAB, MR, NR, WM, false, VAR>(acc, g, batch, split, tile_m, row_blk, col_blk, wm, wn, lane, (float*)lds, BNT);
        } else {
            // shared epilogue (bias / activation / mask / noise / column sums).  An fp8 output (e4m3 after a forward product,
            // e5m2 after a dX product) is packed straight from the accumulators into two LDS byte images and copied out;
            // a bf16 output (feature layer, d loss / d fake x) takes the bf16 tile path of the bf16 kernels.
            if constexpr (OUT8) {
                constexpr int QFMT = EPI == EPI_FWD ? FP8_E4M3 : FP8_E5M2;
                constexpr int Q_BYTES = BM * (BNT + 16) + BNT * (BM + 16);
                epilogue<__bf16, EPI, MR, NR, WM, true, VAR, QFMT>(acc, g, batch, 0, tile_m, row_blk, col_blk, wm, wn, lane,
                                                                   (float*)(lds + Q_BYTES), BNT, (__bf16*)lds, &pf, nullptr);
                __syncthreads();
                copy_tile<BM, BNT, 64 * NW>((const unsigned char*)lds, g.e, batch, row_blk, col_blk, g.M, g.N);
            } else {
                epilogue<__bf16, EPI, MR, NR, WM, true, VAR>(acc, g, batch, 0, tile_m, row_blk, col_blk, wm, wn, lane,
                                                             (float*)(lds + BM * BNT * 2), BNT, (__bf16*)lds, &pf, nullptr);
            }
        }
        __syncthreads();
    }
}

// fp32 -> e4m3 (round to nearest even, saturating at +-448) of x * scale (debug entry's operand staging)
__global__ void to_fp8_kernel(const float* src, long lds_, unsigned char* dst, long ldd, int rows, int cols, int prow, int pcol, float scale, int transpose) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (r >= prow || c >= pcol) return;
    float v = (r < rows && c < cols) ? src[(long)r * lds_ + c] * scale : 0.f;
    v = fminf(fmaxf(v, -448.f), 448.f);
    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false);
    if (transpose) dst[(long)c * ldd + r] = (unsigned char)(w & 0xFF);
    else dst[(long)r * ldd + c] = (unsigned char)(w & 0xFF);
}

// bf16 [nb][rows][ld] -> fp8 (row-major and / or transposed) of v * slot->scale; max |v| -> slot.
// Block = 128 rows x 64 columns; thread = 4 rows x 8 columns, so a column's four rows are one dword of the transposed copy
// without any byte shuffling.  The transposed dwords go through an LDS tile (pitch 33 dwords: conflict-free writes) and
// leave as 16-byte pieces, 128 contiguous bytes of a transposed row per 8 lanes.
constexpr int Q8_ROWS = 128, Q8_PITCH = 132;
template <int FMT>
__global__ __launch_bounds__(256) void quant8_kernel(const Quant8Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char tq[64 * Q8_PITCH];
    const int t = threadIdx.x, batch = blockIdx.z;
    const int row0 = blockIdx.y * Q8_ROWS, col0 = blockIdx.x * 64;
    const float qs = a.slot->scale;
    const __bf16* src = a.src + (long)batch * a.src_bs;
    const int cg = t & 7, rg = t >> 3;
    float v[4][8];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + rg * 4 + i;
        bf16x8 x = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < a.rows) x = *(const bf16x8*)(src + (long)r * a.ld + col0 + cg * 8);
#pragma unroll
        for (int c = 0; c < 8; ++c) { v[i][c] = (float)x[c]; amax = fmaxf(amax, fabsf(v[i][c])); }
    }
    if (a.dst) {
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = row0 + rg * 4 + i;
            if (r < a.prow)
                *(u32x2*)(a.dst + (long)batch * a.dst_bs + (long)r * a.ldd + col0 + cg * 8) =
                    (u32x2){fp8_pack4<FMT>(v[i][0], v[i][1], v[i][2], v[i][3], qs), fp8_pack4<FMT>(v[i][4], v[i][5], v[i][6], v[i][7], qs)};
        }
    }
    if (a.dstt) {
#pragma unroll
        for (int c = 0; c < 8; ++c) *(uint32_t*)(tq + (cg * 8 + c) * Q8_PITCH + rg * 4) = fp8_pack4<FMT>(v[0][c], v[1][c], v[2][c], v[3][c], qs);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ch = t + 256 * u, col = ch >> 3, r0 = (ch & 7) * 16;
            if (row0 + r0 < a.prow) {
                const uint32_t* p = (const uint32_t*)(tq + col * Q8_PITCH + r0);
                *(u32x4*)(a.dstt + (long)(col0 + col) * a.lddt + (long)batch * a.dstt_bs + row0 + r0) = (u32x4){p[0], p[1], p[2], p[3]};
            }
        }
    }
    fp8_amax_commit(a.slot, amax);
}

// delayed scaling: next scale = 2^floor(log2(target / amax)) from the exponent field of the fp32 quotient (bit-exact on
// the CPU mirror); a slot nobody wrote (amax 0) keeps its scale
__global__ void fp8_update_scales_kernel(Fp8Slot* slots, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    Fp8Slot s = slots[i];
    const float a = __uint_as_float(s.amax_bits);
    if (a > 0.f) {
        const float q = s.target / a;
        int e = (int)((__float_as_uint(q) >> 23) & 0xFF) - 127;
        e = max(-100, min(100, e));
        s.scale = __uint_as_float((uint32_t)(127 + e) << 23);
        s.inv_scale = __uint_as_float((uint32_t)(127 - e) << 23);
    }
    s.amax_bits = 0u;
    slots[i] = s;
}
__global__ void fp8_init_slots_kernel(Fp8Slot* slots, int n, const float* targets) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    slots[i] = Fp8Slot{0u, 1.0f, 1.0f, targets[i]};
}

template <int EPI, int BM, int BNT, int WM, int WN, int VAR, bool OUT8>
int launch_fp8_cfg(const GemmArgs& g, hipStream_t s) {
    constexpr int STAGE = BM * 128 + BNT * 128, SCRATCH = 4 * WM * BNT * 4;
    constexpr int OUT = (OUT8 ? BM * (BNT + 16) + BNT * (BM + 16) : BM * BNT * 2) + SCRATCH;
    constexpr int LDS = 2 * STAGE > OUT ? 2 * STAGE : OUT;
    static_assert(LDS <= 160 * 1024, "the ring exceeds the LDS of a CU");
    const int tiles = ceil_div(g.M, BM) * ceil_div(g.N, BNT) * g.nbatch * (EPI == EPI_SLAB ? g.splits : 1);
    dim3 grid(std::min(tiles, 256 * std::max(1, (160 * 1024) / LDS)));
    auto kern = gemm_fp8_kc_kernel<EPI, BM, BNT, WM, WN, VAR, OUT8>;
    static DeviceOnce attr;
    if (attr.first()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -2;
        attr.mark();
    }
    MRGAN_LAUNCH(kern, grid, dim3(64 * WM * WN), LDS, s, g);
    return 0;
}

template <int EPI, int VAR, bool OUT8>
int launch_fp8_var(const GemmArgs& g, hipStream_t s, bool big) {
    return big ? launch_fp8_cfg<EPI, 256, 256, 2, 4, VAR, OUT8>(g, s) : launch_fp8_cfg<EPI, 128, 128, 2, 2, VAR, OUT8>(g, s);
}

}  // namespace

int launch_to_fp8(const float* src, long ld_src, unsigned char* dst, long ld_dst, int rows, int cols, int prow, int pcol, float scale,
                  int transpose, hipStream_t s) {
    hipLaunchKernelGGL(to_fp8_kernel, dim3(ceil_div(pcol, 64), ceil_div(prow, 4)), dim3(256), 0, s, src, ld_src, dst, ld_dst, rows, cols,
                       prow, pcol, scale, transpose);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_quant8(const Quant8Args& a, hipStream_t s) {
    if ((a.cols % 64) || (a.prow % 64) || (a.ld % 8) || !a.slot || a.nb < 1) return -3;
    if (a.dst && (a.ldd % 16)) return -3;
    if (a.dstt && ((a.lddt % 16) || (a.dstt_bs % 16))) return -3;
    const dim3 grid(a.cols / 64, ceil_div(a.prow, Q8_ROWS), a.nb);
    if (a.fmt == FP8_E5M2) MRGAN_LAUNCH(quant8_kernel<FP8_E5M2>, grid, dim3(256), 0, s, a);
    else MRGAN_LAUNCH(quant8_kernel<FP8_E4M3>, grid, dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_fp8_update_scales(Fp8Slot* slots, int n, hipStream_t s) {
    MRGAN_LAUNCH(fp8_update_scales_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, s, slots, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_fp8_init_slots(Fp8Slot* slots, int n, const float* targets_dev, hipStream_t s) {
    hipLaunchKernelGGL(fp8_init_slots_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, s, slots, n, targets_dev);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// g.A [M][K] bytes (a_si = row pitch in bytes), g.B = Bt [N][K] bytes (b_sj), K % 128 == 0; formats fixed by the product
// (FWD e4m3 x e4m3, DX e5m2 x e4m3, SLAB e4m3 x e5m2)
int launch_gemm_fp8(int epi, const GemmArgs& g, hipStream_t s, const char** kname) {
    if ((g.K % BKB) != 0 || g.a_sk != 1 || g.b_sk != 1 || g.splits < 1) return -3;
    if (g.splits > 1 && (epi != EPI_SLAB || g.nbatch != 1 || (g.kchunk % BKB) != 0 || (long)g.kchunk * g.splits != g.K)) return -3;
    if (epi == EPI_SLAB && g.splits == 1 && g.kchunk != g.K) return -3;
    if ((long)g.M * g.a_si >= (1L << 31) || (long)g.N * g.b_sj >= (1L << 31)) return -3;
    const Epi& e = g.e;
    if ((e.q8 || e.q8t) && !e.qo) return -3;
    if (e.q8 && (e.ldq8 % 16)) return -3;
    if (e.q8t && ((e.ldq8t % 16) || (e.q8t_bs % 16))) return -3;
    if ((e.qa == nullptr) != (e.qb == nullptr)) return -3;
    // 256x256 blocks (8 waves of 128x64) once they fill the chip; e.tune_kc_cfg 1 / 3 force the small / large tile
    const int t256 = ceil_div(g.M, 256) * ceil_div(g.N, 256) * g.nbatch * g.splits;
    // (measured at configs[4]: the K = 512 first layer runs 0.18 vs 0.31 ms per launch on the large tile -- its epilogue writes
    // three outputs per element and dominates, and the large tile halves the per-element epilogue overhead of the waves)
    bool big = g.K >= 512 && t256 >= 192 && (g.N % 256) == 0;
    if (e.tune_kc_cfg == 1) big = false;
    if (e.tune_kc_cfg == 3 && (g.N % 256) == 0) big = true;
    const bool noise = e.sigma > 0.f, mask = e.mask != nullptr;
    int r = -3;
    const char* nm = "?";
    const bool out8 = e.q8 || e.q8t;
    if (epi != EPI_SLAB && out8 == (e.out != nullptr)) return -3;      // exactly one output form per forward / dX launch
    if (epi == EPI_FWD) {
        nm = big ? "gemm_fp8_kc_kernel<0, 256, 256>" : "gemm_fp8_kc_kernel<0, 128, 128>";
        if (e.cs_mode != CS_NONE && e.cs_mode != CS_SUM) return -3;
        if (out8) {
            if (e.cs_mode != CS_NONE) return -3;            // (the fp8-output epilogue carries column sums for dX only)
            if (e.act == ACT_RELU && noise && mask) r = launch_fp8_var<EPI_FWD, ACT_RELU | VAR_NOISE | VAR_MASK, true>(g, s, big);
        } else {
            if (e.act == ACT_RELU && !noise && mask) r = launch_fp8_var<EPI_FWD, ACT_RELU | VAR_MASK, false>(g, s, big);
            else if (e.act == ACT_RELU && !noise && !mask) r = launch_fp8_var<EPI_FWD, ACT_RELU, false>(g, s, big);
            else if (e.act == ACT_LINEAR && !noise && !mask) r = launch_fp8_var<EPI_FWD, ACT_LINEAR, false>(g, s, big);
            else if (e.act == ACT_SOFTPLUS && !noise && !mask) r = launch_fp8_var<EPI_FWD, ACT_SOFTPLUS, false>(g, s, big);
        }
    } else if (epi == EPI_DX) {
        nm = big ? "gemm_fp8_kc_kernel<1, 256, 256>" : "gemm_fp8_kc_kernel<1, 128, 128>";
        if (out8) {
            if (e.cs_mode != CS_NONE && e.cs_mode != CS_SUM) return -3;
            if (e.act == ACT_RELU && mask) r = launch_fp8_var<EPI_DX, ACT_RELU, true>(g, s, big);
        } else if (e.act == ACT_LINEAR) {
            // CS_SUM_XHAT (BatchNorm backward sums of the generator): the epilogue reads e.h from global memory
            if (e.cs_mode == CS_SUM_XHAT && !(e.h && e.bn_mu && e.bn_rstd && e.cs2)) return -3;
            if (e.cs_mode == CS_SUM_SQ) return -3;
            r = launch_fp8_var<EPI_DX, ACT_LINEAR, false>(g, s, big);
        }
    } else if (epi == EPI_SLAB) {
        nm = big ? "gemm_fp8_kc_kernel<2, 256, 256>" : "gemm_fp8_kc_kernel<2, 128, 128>";
        if (!e.slab) return -3;
        r = launch_fp8_var<EPI_SLAB, ACT_LINEAR, false>(g, s, big);
    }
    if (kname) *kname = nm;
    if (r) return r;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace mrgan
