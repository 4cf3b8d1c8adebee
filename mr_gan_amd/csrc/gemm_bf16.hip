// bf16-in / fp32-accumulate MFMA GEMMs (v_mfma_f32_32x32x16_bf16) for gfx950.
//
// Kernels, by where the reduction index lives in memory:
//   KC  ("k contiguous"): A[i][k], Bt[j][k] both have the reduction index innermost.
//        forward  Y = X Wt^T   (A = activations [M][K], Bt = transposed bf16 weight copy [N][K])
//        dX       dX = dY W^T  (A = dY [M][N], Bt = bf16 weight copy W [K][N]; reduction over N)
//        fragments are 16-B ds_read_b128 from an XOR-swizzled [row][64 k] image.
//   KS  ("k strided"): A[k][i], B[k][j] have the reduction index outermost.
//        dW = X^T dY (reduction over the batch rows): both operands are row-major activations, so the
//        MFMA fragments (8 consecutive k for one row/col) are gathered with the hardware transposing
//        LDS read ds_read_b64_tr_b16 from a [k][free] image.
//
// Staging is LDS-DMA: `buffer_load_dwordx4 ... offen lds` (16 B per lane, 1 KiB per wave-instruction)
// straight from HBM/L2 into a double-buffered LDS image -- no VGPR round trip, no ds_write pass.
//   * the LDS destination of a wave-instruction is lane-linear, so the bank-conflict swizzle is applied
//     to the per-lane SOURCE address and undone by the same XOR on the fragment read;
//   * the k-offset of a tile is the instruction's SGPR offset: advancing a tile costs no VALU;
//   * rows past the end of an operand fall outside the buffer descriptor's range and arrive as zeros,
//     which is what makes ragged M / N safe without per-lane predicates;
//   * one raw s_barrier per k-tile: wait own loads (vmcnt(0)) -> barrier -> issue tile t+1 -> MFMA on
//     tile t, so the next tile's loads are in flight under this tile's MFMAs.
// Tile order is XCD-aware: consecutive block ids round-robin over the 8 XCDs, so each XCD is handed a
// contiguous run of tiles that share the same A row panel in its private L2.
//
// Block = 4 waves as 2x2.  KC: BM x 128 x 64 with BM = 128 (wave 64x64) or 64 (wave 32x64, used when the
// 128-row grid would not fill the chip).  KS: 128 x 128 x 64.
// The weight-gradient product over ragged segments (batch not a multiple of 128) keeps a register-staged
// kernel that can zero-fill arbitrary reduction rows.
#include <algorithm>
#include <cstring>
#include <string>

#include "gemm.h"

namespace mrgan {

namespace {
constexpr int BN = 128, BK = 64;
constexpr int KS_PITCH = 320;                   // register-staged KS image: 256 data + 64 pad bytes per k-row
constexpr int KS_TILE_BYTES = 64 * KS_PITCH;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ int kc_off(int row, int chunk) {
    // 16-B chunk `chunk` (0..7) of row `row`; (row>>1)&7 spreads the 16 rows of a ds_read_b128
    // lane group over all sixteen 16-B slots of the 256-B bank row
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// XCD-aware tile index: blocks b and b+8 share an XCD, so give XCD x the tiles [x*nt/8, (x+1)*nt/8)
__device__ __forceinline__ int xcd_tile(int bid, int nt) {
    return (nt & 7) == 0 ? (bid & 7) * (nt >> 3) + (bid >> 3) : bid;
}

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_dst, 16, voff, soff, 0, 0);
}

// =====================================================================================================
// KC: forward and input-gradient products
// =====================================================================================================
// counted wait: all but the newest `n` LDS-DMA groups of LPT instructions each have landed
template <int LPT>
__device__ __forceinline__ void wait_groups(int n) {
    if (n >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Block tile BM x BNT, WM x WN waves (each (BM/WM) x (BNT/WN), as MR x NR accumulators of 32x32), NS-stage ring.
// LDS-DMA issue is the scarce resource of this loop (~60-100 issue cycles per 1 KiB wave-instruction), so the
// achievable MFMA share grows with the tile's arithmetic intensity BM*BNT/(BM+BNT): 64x128 -> 43, 128x128 -> 64,
// 256x128 -> 85, 256x256 -> 128 flop per staged byte.
// PIPE: the fragments of k-tile kt+1 are read from LDS into a second register set while the MFMAs of k-tile kt run
// (needs k-tile kt+1 landed one iteration early, so NS >= 4 to keep two k-tiles of LDS-DMA in flight).  Meant for
// launches with <= 1 block per CU, where no second block hides the barrier -> ds_read -> MFMA latency chain.
#ifdef MRGAN_STAMPS
struct KcStamps { unsigned long long acc[6], prev, epi[5]; };      // make STAMPS=1: cycles per phase, summed over the block's tiles
#define KC_STAMPS_PARAM , KcStamps& stamps_
#define KC_STAMPS_ARG , stamps_
#else
#define KC_STAMPS_PARAM
#define KC_STAMPS_ARG
#endif
// one output tile (batch, tile_m, tile_n) of the product described by g: prologue, main loop, epilogue, ending with
// the barrier after which the LDS ring may be refilled
template <int EPI, int BM, int BNT, int WM, int WN, int NS, int VAR, bool PIPE = false>
__device__ __forceinline__ void kc_tile(const GemmArgs& g, const int batch, const int tile_m, const int tile_n, char* lds KC_STAMPS_PARAM) {
    constexpr int NW = WM * WN;
    constexpr int MR = BM / WM / 32, NR = BNT / WN / 32;   // 32x32 accumulators per wave
    constexpr int A_BYTES = BM * 128, B_BYTES = BNT * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BNT / 8 / NW;   // wave-instructions per wave per k-tile (8 rows each)
    static_assert(NS >= 2 && NS <= 4 && (!PIPE || NS >= 3), "ring depth");
    static_assert(MR >= 1 && NR >= 1 && A_INSTR >= 1 && B_INSTR >= 1 && BM % (8 * NW) == 0 && BNT % (8 * NW) == 0, "tile/wave layout");
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane >> 3, lp = lane & 7;       // lane -> (row within an instruction's 8 rows, 16-B chunk)
    const int lr = lane & 31, lh = lane >> 5;
    const int nk = (g.e.ablate & 4) ? 0 : g.K / BK;
#ifdef MRGAN_STAMPS
#define STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); stamps_.acc[i] += n_ - stamps_.prev; stamps_.prev = n_; } while (0)
#else
#define STAMP(i)
#endif
    {
        const int row_blk = tile_m * BM, col_blk = tile_n * BNT;

        // descriptors bounded at the operand's end: rows >= M (A) / >= N (Bt) read as zeros
        const __bf16* Ab = (const __bf16*)g.A + (long)batch * g.a_bs;
        const __bf16* Bb = (const __bf16*)g.B + (long)batch * g.b_bs;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)((long)g.M * g.a_si * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)((long)g.N * g.b_sj * 2), 0x00020000);
        int voffA[A_INSTR], voffB[B_INSTR];          // swizzled per-lane source offsets
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const int R = (wave * A_INSTR + i) * 8 + lrow;
            voffA[i] = (int)(((long)(row_blk + R) * g.a_si + ((lp ^ ((R >> 1) & 7)) << 3)) * 2);
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const int R = (wave * B_INSTR + i) * 8 + lrow;
            voffB[i] = (int)(((long)(col_blk + R) * g.b_sj + ((lp ^ ((R >> 1) & 7)) << 3)) * 2);
        }
        auto issue = [&](int k0, int buf) {
            char* a_dst = lds + buf * STAGE + wave * A_INSTR * 1024;
            char* b_dst = lds + buf * STAGE + A_BYTES + wave * B_INSTR * 1024;
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i) glds16(rsA, a_dst + i * 1024, voffA[i], k0 * 2);
#pragma unroll
            for (int i = 0; i < B_INSTR; ++i) glds16(rsB, b_dst + i * 1024, voffB[i], k0 * 2);
        };

        // ring of NS stages, tiles are issued NS-1 ahead of their use (the previous tile's epilogue ended with a
        // barrier, so the ring is free)
#pragma unroll
        for (int p = 0; p < NS - 1; ++p)
            if (p < nk) issue(p * BK, p);

        // DX epilogues that read e.h element-wise (softplus derivative, xhat sums: one column per lane, rows strided)
        // get the block's tile of h copied into LDS behind the ring with 16-byte LDS-DMA loads; it lands under the main
        // loop (the loads are older than every k-tile group issued inside the loop, so the counted waits cover them)
        const __bf16* htile = nullptr;
        if constexpr (EPI == EPI_DX && (VAR & VAR_ACT_MASK) != ACT_RELU) {
            if (dx_needs_h<VAR>(g.e)) {
                constexpr int OUTB = BM * BNT * 2 + 4 * WM * BNT * 4, RING = NS * STAGE;
                char* hdst = lds + (RING > OUTB ? RING : OUTB);
                const __bf16* hb = (const __bf16*)g.e.h + (long)batch * g.e.h_bs;
                const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)hb, 0, (int)((long)g.M * g.e.ldh * 2), 0x00020000);
                constexpr int RPI = 1024 / (BNT * 2);              // tile rows per wave-instruction
                constexpr int LPR = 64 / RPI;                      // lanes per row
                constexpr int HI = BM / RPI / NW;                  // wave-instructions per wave
#pragma unroll
                for (int i = 0; i < HI; ++i) {
                    const int R = (wave * HI + i) * RPI + lane / LPR;
                    const int voff = (int)(((long)(row_blk + R) * g.e.ldh + col_blk) * 2) + (lane % LPR) * 16;
                    glds16(rsH, hdst + (wave * HI + i) * 1024, voff, 0);
                }
                htile = (const __bf16*)hdst;
            }
        }

        f32x16 acc[MR][NR];
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // bias / relu-mask words of this wave's sub-tiles: their global-load latency hides under the main loop.
        // (issued after the LDS-DMA groups so that the counted waits below still cover every DMA group)
        EpiPrefetch<MR, NR> pf;
        epilogue_prefetch<__bf16, EPI, MR, NR, VAR>(pf, g, batch, row_blk, col_blk, wm, wn, lane);

        if constexpr (!PIPE) {
            int buf = 0;
            for (int kt = 0; kt < nk; ++kt) {
                // vmcnt counts in issue order (stores of the previous tile and the prefetch loads are older than or as old
                // as the group being waited for, so waiting for the group also retires them)
                if (kt == 0) STAMP(0);                                        // tile setup + issue
                wait_groups<A_INSTR + B_INSTR>(min(NS - 2, nk - 1 - kt));   // this wave's loads of k-tile kt have landed
                __builtin_amdgcn_s_barrier();                                 // ... everyone's; everyone finished k-tile kt-1
                asm volatile("" ::: "memory");
                if (kt == 0) STAMP(1);                                        // first k-tile landed (pipeline fill)
                if (kt + NS - 1 < nk) {                                       // refill the stage read during k-tile kt-1
                    int nb = buf + NS - 1; if (nb >= NS) nb -= NS;
                    issue((kt + NS - 1) * BK, nb);
                }
                const char* As = lds + buf * STAGE;
                const char* Bs = As + A_BYTES;
                buf = (buf + 1 == NS) ? 0 : buf + 1;
                // fragments of KG k-steps are fetched as one batch ahead of their MFMAs: the LDS latency is paid once per
                // batch (counted lgkmcnt waits) instead of once per MFMA
                constexpr int KG = (MR + NR <= 4) ? 4 : 2;
    #pragma unroll
                for (int kg = 0; kg < BK / 16; kg += KG) {
                    bf16x8 a[KG][MR], b[KG][NR];
    #pragma unroll
                    for (int kk = 0; kk < KG; ++kk) {
    #pragma unroll
                        for (int mi = 0; mi < MR; ++mi) a[kk][mi] = *(const bf16x8*)(As + kc_off((wm * MR + mi) * 32 + lr, (kg + kk) * 2 + lh));
    #pragma unroll
                        for (int ni = 0; ni < NR; ++ni) b[kk][ni] = *(const bf16x8*)(Bs + kc_off((wn * NR + ni) * 32 + lr, (kg + kk) * 2 + lh));
                    }
                    __builtin_amdgcn_sched_barrier(0);       // keep the scheduler from re-serialising read -> wait -> MFMA
    #pragma unroll
                    for (int kk = 0; kk < KG; ++kk)
    #pragma unroll
                        for (int mi = 0; mi < MR; ++mi)
    #pragma unroll
                            for (int ni = 0; ni < NR; ++ni)
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk][mi], b[kk][ni], acc[mi][ni], 0, 0, 0);
                }
            }
        } else {
            constexpr int KS = BK / 16;
            auto read_frags = [&](bf16x8 (&fa)[KS][MR], bf16x8 (&fb)[KS][NR], int stage) {
                const char* As = lds + stage * STAGE;
                const char* Bs = As + A_BYTES;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
                    for (int mi = 0; mi < MR; ++mi) fa[kk][mi] = *(const bf16x8*)(As + kc_off((wm * MR + mi) * 32 + lr, kk * 2 + lh));
#pragma unroll
                    for (int ni = 0; ni < NR; ++ni) fb[kk][ni] = *(const bf16x8*)(Bs + kc_off((wn * NR + ni) * 32 + lr, kk * 2 + lh));
                }
            };
            bf16x8 fa0[KS][MR], fb0[KS][NR], fa1[KS][MR], fb1[KS][NR];
            int stg = 0;                                                    // stage of k-tile kt
            // one pipeline step: on entry the reads of k-tile kt's fragments into (ca, cb) have been issued
            auto step = [&](bf16x8 (&ca)[KS][MR], bf16x8 (&cb)[KS][NR], bf16x8 (&na)[KS][MR], bf16x8 (&nb)[KS][NR], int kt) {
                int s1 = stg + 1; if (s1 >= NS) s1 -= NS;                   // stage of k-tile kt+1
                if (kt + 1 < nk) wait_groups<A_INSTR + B_INSTR>(min(NS - 3, nk - 2 - kt));   // k-tile kt+1 has landed (this wave)
                __builtin_amdgcn_s_barrier();     // ... everyone's; and every wave has its k-tile kt-1 fragments in registers
                asm volatile("" ::: "memory");
                if (kt + NS - 1 < nk) {                                     // refill the stage of k-tile kt-1
                    int nb = stg - 1; if (nb < 0) nb += NS;
                    issue((kt + NS - 1) * BK, nb);
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0), as a builtin so that the compiler's own wait insertion
                                                       // knows (ca, cb) are complete (issued a whole k-tile ago)
                __builtin_amdgcn_sched_barrier(0);
                if (kt + 1 < nk) read_frags(na, nb, s1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                    for (int mi = 0; mi < MR; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NR; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[kk][mi], cb[kk][ni], acc[mi][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                stg = s1;
            };
            if (nk > 0) {
                wait_groups<A_INSTR + B_INSTR>(min(NS - 2, nk - 1));
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                read_frags(fa0, fb0, 0);
            }
            int kt = 0;
            for (; kt + 1 < nk; kt += 2) { step(fa0, fb0, fa1, fb1, kt); step(fa1, fb1, fa0, fb0, kt + 1); }
            if (kt < nk) step(fa0, fb0, fa1, fb1, kt);
        }
        STAMP(2);               // main loop
        __syncthreads();
        STAMP(3);               // barrier after the main loop
        // the BM x BNT bf16 output tile is assembled at the start of the (now dead) ring, column-sum scratch behind it
        epilogue<__bf16, EPI, MR, NR, WM, true, VAR>(acc, g, batch, 0, tile_m, row_blk, col_blk, wm, wn, lane,
                                                     (float*)(lds + BM * BNT * 2), BNT, (__bf16*)lds, &pf, htile
#ifdef MRGAN_STAMPS
                                                     , stamps_.epi
#endif
                                                     );
        STAMP(4);               // epilogue (math, staging, copy-out issue, column sums)
        __syncthreads();        // the copy-out has read the staged tile: the ring may be refilled
        STAMP(5);
    }
#undef STAMP
}

// Persistent block: walks the tiles bid, bid + grid, ...  One tile's output stores drain while the next tile's
// loads and MFMAs run, and co-resident blocks drift out of phase instead of all hitting HBM at once.
template <int EPI, int BM, int BNT, int WM, int WN, int NS, int VAR, bool PIPE = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16_kc_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];        // max(NS * STAGE, BM * BNT * 2 + scratch) bytes
    const int ntn = (g.N + BNT - 1) / BNT, ntm = (g.M + BM - 1) / BM;
    const int ntiles = ntn * ntm * g.nbatch;
#ifdef MRGAN_STAMPS
    KcStamps stamps_ = {{0, 0, 0, 0, 0, 0}, __builtin_amdgcn_s_memtime(), {0, 0, 0, 0, 0}};
#endif
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int tidx = xcd_tile(tl, ntiles);
        const int batch = tidx / (ntn * ntm), rem = tidx - batch * (ntn * ntm);
        // large problems (operands beyond an XCD's L2): consecutive tiles form patches of 4 tile rows x 8 tile columns, so the
        // CUs of one XCD fetch a third less distinct operand data per k-step than with plain row-major order (gemm_fp8.hip)
        int tile_m, tile_n;
        if ((ntm & 3) == 0 && ntn >= 8 && g.K >= 2048) {
            const int grp = rem / (4 * ntn), in = rem - grp * (4 * ntn);
            tile_m = grp * 4 + (in & 3); tile_n = in >> 2;
        } else { tile_m = rem / ntn; tile_n = rem - tile_m * ntn; }
        kc_tile<EPI, BM, BNT, WM, WN, NS, VAR, PIPE>(g, batch, tile_m, tile_n, lds KC_STAMPS_ARG);
    }
#ifdef MRGAN_STAMPS
    if (g.e.slab && threadIdx.x == 0)
    {
        for (int i = 0; i < 6; ++i) ((unsigned long long*)g.e.slab)[(long)blockIdx.x * 12 + i] = stamps_.acc[i];
        for (int i = 0; i < 4; ++i) ((unsigned long long*)g.e.slab)[(long)blockIdx.x * 12 + 6 + i] = stamps_.epi[i];
    }
#endif
}

thread_local const char* g_last_kernel = "";

template <int EPI, int BM, int BNT, int WM, int WN, int NS, int VAR, bool PIPE = false>
int launch_kc(const GemmArgs& g, hipStream_t s) {
    static const std::string name = "gemm_bf16_kc_kernel<" + std::to_string(EPI) + ", " + std::to_string(BM) + ", " +
                                    std::to_string(BNT) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", " +
                                    std::to_string(NS) + ", " + std::to_string(VAR) + (PIPE ? ", true>" : ", false>");
    g_last_kernel = name.c_str();
    constexpr int STAGE = BM * 128 + BNT * 128;
    // staged output tile + column-sum scratch (+ the tile of e.h for the DX epilogues that read it)
    constexpr int OUT = BM * BNT * 2 + 4 * WM * BNT * 4;
    constexpr int LDS = (NS * STAGE > OUT ? NS * STAGE : OUT) + ((EPI == EPI_DX && (VAR & VAR_ACT_MASK) != ACT_RELU) ? BM * BNT * 2 : 0);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static DeviceOnce attr;
    auto kern = gemm_bf16_kc_kernel<EPI, BM, BNT, WM, WN, NS, VAR, PIPE>;
    if (attr.first()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -2;
        attr.mark();
    }
    // persistent blocks: at most as many as can be co-resident (LDS-limited) on the 256 CUs
    const int tiles = ceil_div(g.M, BM) * ceil_div(g.N, BNT) * g.nbatch;
    const int per_cu = std::max(1, (160 * 1024) / LDS);
    dim3 grid(std::min(tiles, 256 * per_cu));
    MRGAN_LAUNCH(kern, grid, dim3(64 * WM * WN), LDS, s, g);
    return 0;
}

// =====================================================================================================
// KS: weight-gradient product, LDS-DMA version (reduction rows contiguous and a multiple of 64)
// LDS image [64 k][128 free] bf16 with 256-B rows; 16-B chunk c of k-row kr sits at chunk c ^ ((kr&3)<<2),
// which makes the four k-rows x two 32-B column blocks that a 32-lane half of ds_read_b64_tr_b16 touches
// hit eight distinct 32-B bank groups.
// =====================================================================================================
__device__ __forceinline__ bf16x8 ks_frag_swz(const char* tile, int fb, int ks, int lane) {
    const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int f0 = fb + (g4 & 1) * 16;
    const int m0 = ks * 16 + (g4 >> 1) * 8;                        // multiple of 8: (m0 + q) & 3 == q
    const char* a0 = tile + (m0 + q) * 256 + ((((f0 >> 3) + (p >> 1)) ^ (q << 2)) << 4) + ((p & 1) << 3);
    const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * 256));
    const s16x8 tt = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, tt);
}

template <int NS, int WM, int WN>
__device__ __forceinline__ void ks_fast_body(const GemmArgs& g, const int bid) {
    constexpr int NW = WM * WN;
    constexpr int MR = 128 / WM / 32, NR = 128 / WN / 32;
    constexpr int T_BYTES = 64 * 256, STAGE = 2 * T_BYTES;
    constexpr int T_INSTR = 16 / NW;                    // wave-instructions per wave per operand tile (4 k-rows x 256 B each)
    static_assert(MR >= 1 && NR >= 1 && T_INSTR >= 1, "tile/wave layout");
    extern __shared__ __attribute__((aligned(16))) char lds[];        // NS * STAGE bytes

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = (g.N + 127) / 128, ntm = (g.M + 127) / 128;
    const int tidx = xcd_tile(bid, ntn * ntm * g.splits);
    const int split = tidx / (ntn * ntm), rem = tidx - split * (ntn * ntm);
    // wide layers: consecutive tiles (the 64 resident blocks of one XCD) form 8 x 8 patches instead of 2 x 32 strips, half
    // the distinct operand columns per k-step into that XCD's L2 (as in the KC kernels)
    int tile_m, tile_n;
    if ((ntm & 7) == 0 && ntn >= 16) {
        const int grp = rem / (8 * ntn), in = rem - grp * (8 * ntn);
        tile_m = grp * 8 + (in & 7); tile_n = in >> 3;
    } else { tile_m = rem / ntn; tile_n = rem - tile_m * ntn; }
    const int row_blk = tile_m * 128, col_blk = tile_n * 128;
    const int k_begin = split * g.kchunk;
    const int k_end = (g.e.ablate & 4) ? k_begin : min(g.K, k_begin + g.kchunk);

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)((long)g.K * g.a_sk * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, (int)((long)g.K * g.b_sk * 2), 0x00020000);
    // a wave-instruction covers 4 k-rows x 256 B; lane -> (k-row, swizzled source chunk)
    const int lrow = lane >> 4, lp = lane & 15;
    int voffA[T_INSTR], voffB[T_INSTR];
#pragma unroll
    for (int i = 0; i < T_INSTR; ++i) {
        const int kr = (wave * T_INSTR + i) * 4 + lrow;
        const int c = lp ^ ((kr & 3) << 2);
        voffA[i] = (int)(((long)kr * g.a_sk + row_blk + c * 8) * 2);
        voffB[i] = (int)(((long)kr * g.b_sk + col_blk + c * 8) * 2);
    }
    auto issue = [&](int k0, int buf) {
        char* a_dst = lds + buf * STAGE + wave * T_INSTR * 1024;
        char* b_dst = a_dst + T_BYTES;
#pragma unroll
        for (int i = 0; i < T_INSTR; ++i) glds16(rsA, a_dst + i * 1024, voffA[i], (int)((long)k0 * g.a_sk * 2));
#pragma unroll
        for (int i = 0; i < T_INSTR; ++i) glds16(rsB, b_dst + i * 1024, voffB[i], (int)((long)k0 * g.b_sk * 2));
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (k_end - k_begin + BK - 1) / BK;
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nk) issue(k_begin + p * BK, p);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        wait_groups<2 * T_INSTR>(min(NS - 2, nk - 1 - kt));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NS - 1 < nk) {
            int nb = buf + NS - 1; if (nb >= NS) nb -= NS;
            issue(k_begin + (kt + NS - 1) * BK, nb);
        }
        const char* As = lds + buf * STAGE;
        const char* Bs = As + T_BYTES;
        buf = (buf + 1 == NS) ? 0 : buf + 1;
        constexpr int KG = (MR + NR <= 4) ? 4 : 2;       // fragment batches, as in the KC kernel
#pragma unroll
        for (int kg = 0; kg < BK / 16; kg += KG) {
            bf16x8 a[KG][MR], b[KG][NR];
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
#pragma unroll
                for (int mi = 0; mi < MR; ++mi) a[kk][mi] = ks_frag_swz(As, (wm * MR + mi) * 32, kg + kk, lane);
#pragma unroll
                for (int ni = 0; ni < NR; ++ni) b[kk][ni] = ks_frag_swz(Bs, (wn * NR + ni) * 32, kg + kk, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KG; ++kk)
#pragma unroll
                for (int mi = 0; mi < MR; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NR; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk][mi], b[kk][ni], acc[mi][ni], 0, 0, 0);
        }
    }
    __syncthreads();
    epilogue<__bf16, EPI_SLAB, MR, NR, WM>(acc, g, 0, split, tile_m, row_blk, col_blk, wm, wn, lane, (float*)lds, BN);
}

template <int NS, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16_ks_fast_kernel(const GemmArgs g) {
    ks_fast_body<NS, WM, WN>(g, blockIdx.x);
}

// All weight-gradient products of one sub-step in ONE launch: they have no consumer before the Adam kernel, each is
// small (32 .. 256 blocks, 5-7 us of launch + drain per launch for 3-12 us of work), and together they fill the chip.
// Block b works on problem i with blk_end[i-1] <= b < blk_end[i].  Blocks whose index inside the problem agrees mod 8
// still share an XCD (a constant rotation of b % 8), which is all the XCD-aware tile order of ks_fast_body relies on.
template <int NS, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16_ks_group_kernel(const KsGroup grp) {
    const int ngemm = grp.blk_end[grp.n - 1];
    if ((int)blockIdx.x >= ngemm) {
        // the fold job's blocks (see FoldJob): only the first 256 threads of a block take part
        const FoldJob& f = grp.fold;
        const int fb = (int)blockIdx.x - ngemm, gidx = fb / f.blocks_x;
        const int i = (fb - gidx * f.blocks_x) * 256 + (int)threadIdx.x;
        if (threadIdx.x >= 256 || i >= f.n) return;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int p = gidx;
        for (; p + 3 * f.ngroups < f.nsrc; p += 4 * f.ngroups) {
            s0 += f.src[(long)p * f.stride + i]; s1 += f.src[(long)(p + f.ngroups) * f.stride + i];
            s2 += f.src[(long)(p + 2 * f.ngroups) * f.stride + i]; s3 += f.src[(long)(p + 3 * f.ngroups) * f.stride + i];
        }
        for (; p < f.nsrc; p += f.ngroups) s0 += f.src[(long)p * f.stride + i];
        f.dst[(long)gidx * f.stride + i] = (s0 + s1) + (s2 + s3);
        return;
    }
    int i = 0, b0 = 0;
#pragma unroll
    for (int j = 0; j < KS_GROUP_MAX - 1; ++j)
        if (j + 1 < grp.n && (int)blockIdx.x >= grp.blk_end[j]) { i = j + 1; b0 = grp.blk_end[j]; }
    ks_fast_body<NS, WM, WN>(grp.g[i], (int)blockIdx.x - b0);
}

template <int NS, int WM, int WN>
int launch_ks_fast(const GemmArgs& g, hipStream_t s) {
    static const std::string name = "gemm_bf16_ks_fast_kernel<" + std::to_string(NS) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ">";
    g_last_kernel = name.c_str();
    constexpr int STAGE = 2 * 64 * 256;
    static DeviceOnce attr;
    auto kern = gemm_bf16_ks_fast_kernel<NS, WM, WN>;
    if (attr.first()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE) != hipSuccess) return -2;
        attr.mark();
    }
    dim3 grid(ceil_div(g.N, 128) * ceil_div(g.M, 128) * g.splits);
    MRGAN_LAUNCH(kern, grid, dim3(64 * WM * WN), NS * STAGE, s, g);
    return 0;
}

// =====================================================================================================
// KS, register-staged: handles reduction ranges with holes (segments of a batch that is not a multiple
// of 128) by zero-filling rows per lane.  LDS image [64 k][128 free] with a 320-B pitch.
// =====================================================================================================
__device__ __forceinline__ bf16x8 ks_frag(const char* tile, int fb, int ks, int lane) {
    const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int f0 = fb + (g4 & 1) * 16;
    const int m0 = ks * 16 + (g4 >> 1) * 8;
    const char* a0 = tile + (m0 + q) * KS_PITCH + (f0 + 4 * p) * 2;
    // (the v4i16 form: per-element use of the v4bf16 form's result is mis-folded by hipcc 7.2)
    const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * KS_PITCH));
    const s16x8 tt = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, tt);
}

__global__ __launch_bounds__(256) void gemm_bf16_ks_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char lds[2 * KS_TILE_BYTES];
    char* As = lds;
    char* Bs = lds + KS_TILE_BYTES;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_n = blockIdx.x, tile_m = blockIdx.y;
    const int batch = blockIdx.z / g.splits, split = blockIdx.z % g.splits;
    const int row_blk = tile_m * 128, col_blk = tile_n * BN;
    const int k_begin = split * g.kchunk;
    const int k_end = (g.e.ablate & 4) ? k_begin : min(g.K, k_begin + g.kchunk);

    const __bf16* A = (const __bf16*)g.A + (long)batch * g.a_bs;
    const __bf16* B = (const __bf16*)g.B + (long)batch * g.b_bs;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[4], rb[4];
    auto load_tile = [&](int k0) {
        const int kloc = k0 % g.seg_stride;             // BK divides seg_stride: a tile never straddles segments
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cidx = t + 256 * u, kr = cidx >> 4, c = cidx & 15;
            const u32x4 z = {0u, 0u, 0u, 0u};
            const bool kok = (k0 + kr) < k_end && (kloc + kr) < g.seg_rows;
            ra[u] = (kok && row_blk + c * 8 < g.M) ? *(const u32x4*)(A + (long)(k0 + kr) * g.a_sk + row_blk + c * 8) : z;
            rb[u] = (kok && col_blk + c * 8 < g.N) ? *(const u32x4*)(B + (long)(k0 + kr) * g.b_sk + col_blk + c * 8) : z;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cidx = t + 256 * u, kr = cidx >> 4, c = cidx & 15;
            *(u32x4*)(As + kr * KS_PITCH + c * 16) = ra[u];
            *(u32x4*)(Bs + kr * KS_PITCH + c * 16) = rb[u];
        }
    };

    if (k_begin < k_end) {
        load_tile(k_begin);
        for (int k0 = k_begin; k0 < k_end; k0 += BK) {
            __syncthreads();
            store_tile();
            __syncthreads();
            if (k0 + BK < k_end) load_tile(k0 + BK);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = ks_frag(As, (wm * 2 + mi) * 32, ks, lane);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) b[ni] = ks_frag(Bs, (wn * 2 + ni) * 32, ks, lane);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    epilogue<__bf16, EPI_SLAB, 2, 2, 2>(acc, g, batch, split, tile_m, row_blk, col_blk, wm, wn, lane, (float*)lds, BN);
}

// diagnostic: out[0..511]  = what ds_read_b64_tr_b16 returns when lane l supplies the address of u16 elements
//                            4l..4l+3 of a linear image whose element i holds the value i;
//             out[512..1023] = ks_frag() on a [16][160] image whose element (row, col) holds row<<8 | col.
__global__ void tr_probe_kernel(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short img[16 * 160];
    const int l = threadIdx.x;
    for (int i = l; i < 16 * 160; i += 64) img[i] = (unsigned short)i;
    __syncthreads();
    const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((const char*)img + l * 8));
    for (int j = 0; j < 4; ++j) out[l * 8 + j] = (unsigned short)t[j];
    for (int j = 4; j < 8; ++j) out[l * 8 + j] = 0;
    __syncthreads();
    for (int i = l; i < 16 * 160; i += 64) img[i] = (unsigned short)(((i / 160) << 8) | (i % 160));
    __syncthreads();
    const s16x8 f = __builtin_bit_cast(s16x8, ks_frag((const char*)img, 0, 0, l));
    for (int j = 0; j < 8; ++j) out[512 + l * 8 + j] = (unsigned short)f[j];
}
}  // namespace

// epilogue variants compiled for the bf16 path (anything else is a host-side error)
// tile configs: 0 = 64x128 / 4 waves / 3 stages ; 1 = 128x128 / 4 waves / 2 stages ; 2 = 256x128 / 8 waves / 2 stages ;
//               3 = 256x256 / 8 waves / 2 stages ; 4 = cfg 0 with pipelined fragments ; 5 = 64x128 / 4 waves / 2 stages
//               (three blocks per CU) ; 6 = 128x256 / 8 waves ; 7 = 128x128 / EIGHT waves of 64x32 / 2 stages ; 9 = 64x64 / 4 waves /
//               3 stages.
//               mrgan_set_tuning(MRGAN_TUNE_KC_CFG) forces one; default picks by grid size.
template <int EPI, int VAR>
static int launch_kc_tile(const GemmArgs& g, hipStream_t s) {
    const int forced = g.e.tune_kc_cfg;
    int cfg = forced;
    if (cfg < 0) {
        // measured on MI355X (scripts/gemm_bench.py): bigger tiles win once they still give >= ~1.5 blocks per CU
        const int t128 = ceil_div(g.M, 128) * ceil_div(g.N, 128) * g.nbatch;
        const int t256 = ceil_div(g.M, 256) * ceil_div(g.N, 256) * g.nbatch;
        // 128 x 128 tiles as EIGHT waves of 64 x 32 (cfg 7; two blocks = 16 waves per CU): more waves hide the barrier -> LDS read ->
        // MFMA chain of the short k-loops better than four waves of 64 x 64 (cfg 1) although they read 1.5 instead of 1.0
        // fragments per MFMA: D1 forward over 3 segments 22.9 -> 21.3 us, dX through D2 20.3 -> 18.7 us (scripts/gemm_bench.py)
        cfg = t128 >= 384 ? 7 : 0;
        // long reductions with enough 256x256 tiles to fill the chip (the wide stack of BASELINE configs[4]: K = N = 4096):
        // 128 flop per staged byte instead of 64 -- 1.37 vs 0.95 PFLOP/s measured at 24576 x 4096 x 4096 (scripts/fp8_bench.py)
        if (g.K >= 2048 && t256 >= 256 && (g.N % 256) == 0) cfg = 3;
        // long reductions into a narrow output (D2 forward: K = 1024, N = 512): three 64x128 blocks per CU with a
        // 2-stage ring hide the k-loop latency better than two 128x128 blocks (20.7 vs 21.7 us)
        if (EPI == EPI_FWD && cfg == 7 && g.K >= 1024 && g.K < 2048 && g.N <= 512) cfg = 5;
        // ... and with exactly one 128x128 tile per CU (D2 forward over the two segments of the G sub-step) the 8-wave tile
        // beats two 64x128 blocks per CU: 13.9 vs 15.0 us (scripts/gemm_bench.py, KC_CFG sweep)
        if (EPI == EPI_FWD && cfg == 0 && g.K >= 1024 && g.K < 2048 && t128 >= 256 && (g.N % 128) == 0) cfg = 7;
        // launches with at most one 64x128 tile per CU (the one-segment products of the G sub-step, small batches): 64x64 tiles
        // put twice as many blocks on the chip -- dX through D1 11.5 -> 9.7 us, d(BatchNorm output) 11.0 -> 8.5, G2 forward of one
        // segment 8.6 -> 7.4; with two tiles per CU already (two-segment launches) the smaller tile loses (10.5 -> 12.4)
        if (cfg == 0 && ceil_div(g.M, 64) * ceil_div(g.N, 128) * g.nbatch <= 256) cfg = 9;
    }
    if (cfg >= 2 && cfg != 9 && (g.N % 128) != 0) cfg = 0;
    if (cfg == 6 && (g.N % 256) != 0) cfg = 1;
    // DX epilogues that may stage a tile of e.h in LDS (softplus derivative, xhat sums) only exist for the small tiles
    constexpr bool H_TILE = EPI == EPI_DX && (VAR & VAR_ACT_MASK) != ACT_RELU;
    if (H_TILE && cfg >= 2 && cfg != 9) cfg = 1;
    if constexpr (!H_TILE) {
        if (cfg == 7) return launch_kc<EPI, 128, 128, 2, 4, 2, VAR>(g, s);
        if (cfg == 6 && (g.N % 256) == 0) return launch_kc<EPI, 128, 256, 2, 4, 2, VAR>(g, s);      // 8 waves of 64x64
        if (cfg == 2) return launch_kc<EPI, 256, 128, 4, 2, 2, VAR>(g, s);
        if (cfg == 3) return (g.N % 256) == 0 ? launch_kc<EPI, 256, 256, 2, 4, 2, VAR>(g, s) : launch_kc<EPI, 256, 128, 4, 2, 2, VAR>(g, s);
    }
    if (cfg == 9) return launch_kc<EPI, 64, 64, 2, 2, 3, VAR>(g, s);       // 48 KiB ring: three blocks per CU
    if (cfg == 1) return launch_kc<EPI, 128, 128, 2, 2, 2, VAR>(g, s);
    if (cfg == 5) return launch_kc<EPI, 64, 128, 2, 2, 2, VAR>(g, s);      // 48 KiB ring: three blocks per CU
    // cfg 4 / TUNE_BIT_KC_PIPE (launches with at most one 64x128 tile per CU): pipelined fragments, 4-stage ring.
    // Measured no faster than cfg 0 on MI355X -- these launches are bound by the L2 -> LDS fill, not by the LDS -> MFMA chain.
    const int pipe = g.e.tune_bits & TUNE_BIT_KC_PIPE;
    const int t64 = ceil_div(g.M, 64) * ceil_div(g.N, 128) * g.nbatch;
    if (cfg == 4 || (forced < 0 && pipe && t64 <= 256)) return launch_kc<EPI, 64, 128, 2, 2, 4, VAR, true>(g, s);
    return launch_kc<EPI, 64, 128, 2, 2, 3, VAR>(g, s);
}

static int launch_kc_any(int epi, const GemmArgs& g, hipStream_t s) {
    const Epi& e = g.e;
    if (epi == EPI_FWD) {
        const bool noise = e.sigma > 0.f, mask = e.mask != nullptr;
        if (e.act == ACT_RELU && noise && mask) return launch_kc_tile<EPI_FWD, ACT_RELU | VAR_NOISE | VAR_MASK>(g, s);
        if (e.act == ACT_RELU && !noise && mask) return launch_kc_tile<EPI_FWD, ACT_RELU | VAR_MASK>(g, s);
        if (e.act == ACT_RELU && !noise && !mask) return launch_kc_tile<EPI_FWD, ACT_RELU>(g, s);
        if (e.act == ACT_LINEAR && !mask) return noise ? launch_kc_tile<EPI_FWD, ACT_LINEAR | VAR_NOISE>(g, s)
                                                       : launch_kc_tile<EPI_FWD, ACT_LINEAR>(g, s);
        if (e.act == ACT_SOFTPLUS && !noise && !mask) return launch_kc_tile<EPI_FWD, ACT_SOFTPLUS>(g, s);
        return -3;
    }
    if (e.act == ACT_RELU) return launch_kc_tile<EPI_DX, ACT_RELU>(g, s);
    if (e.act == ACT_SOFTPLUS) return launch_kc_tile<EPI_DX, ACT_SOFTPLUS>(g, s);
    return launch_kc_tile<EPI_DX, ACT_LINEAR>(g, s);
}

static bool ks_dense_k(const GemmArgs& g) {
    return g.a_si == 1 && g.b_sj == 1 && g.nbatch == 1 && (g.K % BK) == 0 && (g.kchunk % BK) == 0 &&
           (g.seg_rows >= g.seg_stride || g.K <= g.seg_rows) &&
           (long)g.K * g.a_sk * 2 < (1L << 31) && (long)g.K * g.b_sk * 2 < (1L << 31);
}

// n weight-gradient products as one launch; returns 1 (nothing launched) when a problem does not fit the grouped kernel
int launch_gemm_bf16_dw_group(const GemmArgs* gs, int n, hipStream_t s, const char** kname, const FoldJob* fold) {
    if (n < 1 || n > KS_GROUP_MAX) return 1;
    if (gs[0].e.tune_bits & TUNE_BIT_NO_KS_GROUP) return 1;
    KsGroup grp;
    memset(&grp, 0, sizeof grp);
    grp.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        if (!ks_dense_k(gs[i])) return 1;
        const int blocks = ceil_div(gs[i].N, 128) * ceil_div(gs[i].M, 128) * gs[i].splits;
        total += blocks;
        grp.blk_end[i] = total;
        grp.g[i] = gs[i];
    }
    if (fold && fold->n > 0) {
        grp.fold = *fold;
        grp.fold.blocks_x = ceil_div(fold->n, 256);
        total += grp.fold.blocks_x * fold->ngroups;
    }
    // Block shape, by measurement (B = 4096, D = 512, the discriminator's launch): 8 waves of 64x32 with a 2-stage ring, two
    // blocks = 16 waves per CU: 58.7 us; 4 waves of 64x64, two blocks per CU (round 2's default, 1.0 instead of 1.5 transposing
    // reads per MFMA): 64.7 us; 8 waves with a 3-stage ring, one block per CU: 89 us.  More resident waves hide the
    // barrier -> LDS read -> MFMA chain; the LDS read rate is not the limit at two blocks per CU.
    // MRGAN_TUNE_KS_W8: 0 (default) the first, 1 the third, 2 the second.
    constexpr int STAGE = 2 * 64 * 256;
    static DeviceOnce attr8, attr4, attr16;
    if (gs[0].e.tune_bits & TUNE_BIT_KS_W8) {
        auto kern = gemm_bf16_ks_group_kernel<3, 2, 4>;
        if (attr8.first()) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE) != hipSuccess) return -2;
            attr8.mark();
        }
        MRGAN_LAUNCH(kern, dim3(total), dim3(512), 3 * STAGE, s, grp);
        if (kname) *kname = "gemm_bf16_ks_group_kernel<3, 2, 4>";
    } else if (gs[0].e.tune_bits & TUNE_BIT_KS_W4) {
        auto kern = gemm_bf16_ks_group_kernel<2, 2, 2>;
        if (attr4.first()) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE) != hipSuccess) return -2;
            attr4.mark();
        }
        MRGAN_LAUNCH(kern, dim3(total), dim3(256), 2 * STAGE, s, grp);
        if (kname) *kname = "gemm_bf16_ks_group_kernel<2, 2, 2>";
    } else {
        auto kern = gemm_bf16_ks_group_kernel<2, 2, 4>;
        if (attr16.first()) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE) != hipSuccess) return -2;
            attr16.mark();
        }
        MRGAN_LAUNCH(kern, dim3(total), dim3(512), 2 * STAGE, s, grp);
        if (kname) *kname = "gemm_bf16_ks_group_kernel<2, 2, 4>";
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_gemm_bf16(int epi, const GemmArgs& g, hipStream_t s, const char** kname) {
    int r = 0;
    if (epi == EPI_SLAB) {
        if (g.a_si != 1 || g.b_sj != 1) return -3;
        const bool dense_k = g.nbatch == 1 && (g.K % BK) == 0 && (g.kchunk % BK) == 0 &&
                             (g.seg_rows >= g.seg_stride || g.K <= g.seg_rows) &&
                             (long)g.K * g.a_sk * 2 < (1L << 31) && (long)g.K * g.b_sk * 2 < (1L << 31);
        if (dense_k) {
            r = launch_ks_fast<3, 2, 4>(g, s);      // single product: 8 waves (64x32 each), 3-stage ring
        } else {
            dim3 grid(ceil_div(g.N, BN), ceil_div(g.M, 128), g.nbatch * g.splits);
            MRGAN_LAUNCH(gemm_bf16_ks_kernel, grid, dim3(256), 0, s, g);
            g_last_kernel = "gemm_bf16_ks_kernel";
        }
    } else {
        if (g.a_sk != 1 || g.b_sk != 1 || g.splits != 1 || (g.K % BK) != 0) return -3;
        if ((long)g.M * g.a_si * 2 >= (1L << 31) || (long)g.N * g.b_sj * 2 >= (1L << 31)) return -3;
        r = launch_kc_any(epi, g, s);
    }
    if (r) return r;
    if (kname) *kname = g_last_kernel;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_tr_probe(unsigned short* out, hipStream_t s) {
    MRGAN_LAUNCH(tr_probe_kernel, dim3(1), dim3(64), 0, s, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace mrgan
