// fp8 (OCP e4m3) forward product on the matrix cores: Y[M][N] = act((A[M][K] Bt[N][K]^T) / (sA sB) + b), bf16 out.
//
// First member of the fp8 family BASELINE configs[4] asks for (hidden 4096 x 5): v_mfma_f32_32x32x16_fp8_fp8, fp32
// accumulate, per-tensor power-of-two scales sA / sB applied to the accumulator.  The non-scaled fp8 MFMA issues at the bf16
// rate (MI355X_MICROARCH.md, Matrix cores); what fp8 buys here is bytes: an operand tile of [rows][128 B] now holds 128
// reduction elements instead of 64, so the L2 -> LDS fill -- which bounds the bf16 forward kernels of this build at 128x128
// tiles -- is halved per FLOP.  Structure = the bf16 KC kernel (gemm_bf16.hip): LDS-DMA
// (buffer_load ... lds) into an XOR-swizzled 2-stage ring, one s_barrier per k-tile, persistent blocks in XCD-aware
// tile order, the shared fused epilogue of gemm.h.  Fragments are 8-byte ds_read_b64 (8 fp8 per lane and k-step).
// Used through mrgan_debug_gemm_fp8 / mrgan_debug_gemm_time (kernel-level parity and timing); the training path does not
// select it yet (what the engine still needs -- e5m2 gradients, scale tracking in the Adam kernel).
#include <algorithm>
#include <string>

#include "gemm.h"

namespace mrgan {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
constexpr int BKB = 128;                        // reduction BYTES (= fp8 elements) per k-tile

__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int xcd_tile(int bid, int nt) { return (nt & 7) == 0 ? (bid & 7) * (nt >> 3) + (bid >> 3) : bid; }
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_dst, 16, voff, soff, 0, 0);
}

template <int BM, int BNT, int WM, int WN, int VAR>
__global__ __launch_bounds__(64 * WM * WN) void gemm_fp8_kc_kernel(const GemmArgs g) {
    constexpr int NW = WM * WN;
    constexpr int MR = BM / WM / 32, NR = BNT / WN / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BNT * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BNT / 8 / NW;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane >> 3, lp = lane & 7, lr = lane & 31, lh = lane >> 5;
    const int ntn = (g.N + BNT - 1) / BNT, ntm = (g.M + BM - 1) / BM;
    const int ntiles = ntn * ntm * g.nbatch, nk = g.K / BKB;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int tidx = xcd_tile(tl, ntiles);
        const int batch = tidx / (ntn * ntm), rem = tidx - batch * (ntn * ntm);
        const int tile_m = rem / ntn, tile_n = rem - tile_m * ntn;
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
        auto issue = [&](int kb, int buf) {
            char* a_dst = lds + buf * STAGE + wave * A_INSTR * 1024;
            char* b_dst = lds + buf * STAGE + A_BYTES + wave * B_INSTR * 1024;
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i) glds16(rsA, a_dst + i * 1024, voffA[i], kb);
#pragma unroll
            for (int i = 0; i < B_INSTR; ++i) glds16(rsB, b_dst + i * 1024, voffB[i], kb);
        };
        if (nk > 0) issue(0, 0);
        f32x16 acc[MR][NR];
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        EpiPrefetch<MR, NR> pf;
        epilogue_prefetch<__bf16, EPI_FWD, MR, NR, VAR>(pf, g, batch, row_blk, col_blk, wm, wn, lane);
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of k-tile kt have landed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 1 < nk) issue((kt + 1) * BKB, buf ^ 1);
            const char* As = lds + buf * STAGE;
            const char* Bs = As + A_BYTES;
            buf ^= 1;
#pragma unroll
            for (int kg = 0; kg < 8; kg += 4) {
                long a[4][MR], b[4][NR];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int mi = 0; mi < MR; ++mi) a[kk][mi] = *(const long*)(As + kc_off((wm * MR + mi) * 32 + lr, kg + kk) + lh * 8);
#pragma unroll
                    for (int ni = 0; ni < NR; ++ni) b[kk][ni] = *(const long*)(Bs + kc_off((wn * NR + ni) * 32 + lr, kg + kk) + lh * 8);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int mi = 0; mi < MR; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NR; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a[kk][mi], b[kk][ni], acc[mi][ni], 0, 0, 0);
            }
        }
        // undo the operand scales on the accumulator, then the shared epilogue (bias / activation / bf16 tile via LDS)
        const float us = g.e.acc_scale;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= us;
        __syncthreads();
        epilogue<__bf16, EPI_FWD, MR, NR, WM, true, VAR>(acc, g, batch, 0, tile_m, row_blk, col_blk, wm, wn, lane,
                                                         (float*)(lds + BM * BNT * 2), BNT, (__bf16*)lds, &pf, nullptr);
        __syncthreads();
    }
}

// fp32 -> e4m3 (round to nearest even, saturating at +-448) of x * scale: four elements per thread
__global__ void to_fp8_kernel(const float* src, long lds_, unsigned char* dst, long ldd, int rows, int cols, int prow, int pcol, float scale, int transpose) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (r >= prow || c >= pcol) return;
    float v = (r < rows && c < cols) ? src[(long)r * lds_ + c] * scale : 0.f;
    v = fminf(fmaxf(v, -448.f), 448.f);
    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false);
    if (transpose) dst[(long)c * ldd + r] = (unsigned char)(w & 0xFF);
    else dst[(long)r * ldd + c] = (unsigned char)(w & 0xFF);
}

}  // namespace

int launch_to_fp8(const float* src, long ld_src, unsigned char* dst, long ld_dst, int rows, int cols, int prow, int pcol, float scale,
                  int transpose, hipStream_t s) {
    hipLaunchKernelGGL(to_fp8_kernel, dim3(ceil_div(pcol, 64), ceil_div(prow, 4)), dim3(256), 0, s, src, ld_src, dst, ld_dst, rows, cols,
                       prow, pcol, scale, transpose);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// forward product with e4m3 operands: g.A [M][K] bytes (a_si = row pitch in bytes), g.B = Bt [N][K] bytes (b_sj), K % 128 == 0
int launch_gemm_fp8_fwd(const GemmArgs& g, hipStream_t s, const char** kname) {
    if ((g.K % BKB) != 0 || g.a_sk != 1 || g.b_sk != 1 || g.splits != 1) return -3;
    if ((long)g.M * g.a_si >= (1L << 31) || (long)g.N * g.b_sj >= (1L << 31)) return -3;
    const Epi& e = g.e;
    if (e.sigma > 0.f || e.mask) return -3;                      // (noise / mask variants: with the engine integration)
    constexpr int BM = 128, BNT = 128, WM = 2, WN = 2;
    constexpr int STAGE = BM * 128 + BNT * 128, OUT = BM * BNT * 2 + 4 * WM * BNT * 4;
    constexpr int LDS = 2 * STAGE > OUT ? 2 * STAGE : OUT;
    const int tiles = ceil_div(g.M, BM) * ceil_div(g.N, BNT) * g.nbatch;
    dim3 grid(std::min(tiles, 256 * std::max(1, (160 * 1024) / LDS)));
    static bool attr[3] = {false, false, false};
#define FP8_LAUNCH(IDX, VARV)                                                                                                   \
    do {                                                                                                                        \
        auto kern = gemm_fp8_kc_kernel<BM, BNT, WM, WN, VARV>;                                                                  \
        if (!attr[IDX]) {                                                                                                       \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -2; \
            attr[IDX] = true;                                                                                                   \
        }                                                                                                                       \
        MRGAN_LAUNCH(kern, grid, dim3(64 * WM * WN), LDS, s, g);                                                                \
    } while (0)
    if (e.act == ACT_RELU) { FP8_LAUNCH(0, ACT_RELU); if (kname) *kname = "gemm_fp8_kc_kernel<128, 128, 2, 2, 1>"; }
    else if (e.act == ACT_SOFTPLUS) { FP8_LAUNCH(1, ACT_SOFTPLUS); if (kname) *kname = "gemm_fp8_kc_kernel<128, 128, 2, 2, 2>"; }
    else { FP8_LAUNCH(2, ACT_LINEAR); if (kname) *kname = "gemm_fp8_kc_kernel<128, 128, 2, 2, 0>"; }
#undef FP8_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace mrgan
