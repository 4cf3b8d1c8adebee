// fp32-in / fp32-accumulate MFMA GEMM (v_mfma_f32_32x32x2_f32) -- the "logits within 1e-3" mode.
// The product is bit-for-bit a k-ordered fmaf chain per output element, so results are reproducible.
// Operands are addressed through generic (row, k) strides, so the same kernel serves the forward
// (X W), input-gradient (dY W^T) and weight-gradient (X^T dY) products of every dense layer.
//
// Block tile TS x TS (128 or 64), BK=16, 4 waves as 2x2, each wave (TS/2)^2 = (TS/64)^2 accumulators of 32x32.
// The 64x64 tile serves the reference's own batch of 50: a 128x128 tile there is 61 % padding rows and the few
// blocks (3 x N/128) leave the chip idle while each grinds through K/16 k-tiles of 32 MFMAs; 64x64 blocks do a
// quarter of the MFMAs per k-tile and there are four times as many of them.
// LDS image is k-major ([k][row]) for both operands: the 32x32x2 fragments are one float per lane,
// lane l reads (row l&31, k = l>>5), i.e. 32 consecutive floats per half-wave -> conflict-free.
#include "gemm.h"

namespace mrgan {

namespace {
constexpr int BK = 16;

template <int EPI, int TS>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
    constexpr int BM = TS, BN = TS, LDT = TS, MR = TS / 64, KPT = TS / 16;       // KPT: k values staged per thread
    __shared__ __attribute__((aligned(16))) float lds[2 * BK * 128];              // (the epilogue's scratch needs the 128 form)
    float* As = lds;
    float* Bs = lds + BK * LDT;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_n = blockIdx.x, tile_m = blockIdx.y;
    const int batch = blockIdx.z / g.splits, split = blockIdx.z % g.splits;
    const int row_blk = tile_m * BM, col_blk = tile_n * BN;
    const int k_begin = split * g.kchunk;
    const int k_end = min(g.K, k_begin + g.kchunk);

    const float* A = (const float*)g.A + (long)batch * g.a_bs;
    const float* B = (const float*)g.B + (long)batch * g.b_bs;

    // staging map: thread -> (element t % TS along the free dim, KPT consecutive k starting at (t / TS) * KPT)
    const int si = t % TS, sk = (t / TS) * KPT;
    const bool a_row_ok = (row_blk + si) < g.M;
    const bool b_col_ok = (col_blk + si) < g.N;
    const float* a_ptr = A + (long)(row_blk + si) * g.a_si;
    const float* b_ptr = B + (long)(col_blk + si) * g.b_sj;

    f32x16 acc[MR][MR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < MR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[KPT], rb[KPT];
    auto load_tile = [&](int k0) {
        const int kloc = (k0 % g.seg_stride) + sk;      // BK divides seg_stride: a tile never straddles segments
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int k = k0 + sk + j;
            const bool kok = k < k_end && (kloc + j) < g.seg_rows;
            ra[j] = (a_row_ok && kok) ? a_ptr[(long)k * g.a_sk] : 0.f;
            rb[j] = (b_col_ok && kok) ? b_ptr[(long)k * g.b_sk] : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            As[(sk + j) * LDT + si] = ra[j];
            Bs[(sk + j) * LDT + si] = rb[j];
        }
    };

    if (k_begin < k_end) {
        load_tile(k_begin);
        for (int k0 = k_begin; k0 < k_end; k0 += BK) {
            __syncthreads();                 // previous tile's fragment reads done
            store_tile();
            __syncthreads();
            if (k0 + BK < k_end) load_tile(k0 + BK);      // in flight under the MFMAs below
            const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                float a[MR], b[MR];
#pragma unroll
                for (int mi = 0; mi < MR; ++mi) a[mi] = As[(2 * kk + lh) * LDT + (wm * MR + mi) * 32 + lr];
#pragma unroll
                for (int ni = 0; ni < MR; ++ni) b[ni] = Bs[(2 * kk + lh) * LDT + (wn * MR + ni) * 32 + lr];
#pragma unroll
                for (int mi = 0; mi < MR; ++mi)
#pragma unroll
                    for (int ni = 0; ni < MR; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    epilogue<float, EPI, MR, MR, 2>(acc, g, batch, split, tile_m, row_blk, col_blk, wm, wn, lane, lds, BN);
}
}  // namespace

template <int TS>
static int launch_f32_ts(int epi, const GemmArgs& g, hipStream_t s) {
    dim3 grid(ceil_div(g.N, TS), ceil_div(g.M, TS), g.nbatch * g.splits);
    dim3 block(256);
    switch (epi) {
        case EPI_FWD:  MRGAN_LAUNCH((gemm_f32_kernel<EPI_FWD, TS>), grid, block, 0, s, g); break;
        case EPI_DX:   MRGAN_LAUNCH((gemm_f32_kernel<EPI_DX, TS>), grid, block, 0, s, g); break;
        case EPI_SLAB: MRGAN_LAUNCH((gemm_f32_kernel<EPI_SLAB, TS>), grid, block, 0, s, g); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_gemm_f32(int epi, const GemmArgs& g, hipStream_t s, const char** kname) {
    // 64x64 blocks while 128x128 ones would leave most CUs without work (small batches); the choice never changes
    // the result: every output element is the same k-ordered fmaf chain in both
    const int blocks128 = ceil_div(g.N, 128) * ceil_div(g.M, 128) * g.nbatch * g.splits;
    const bool small = blocks128 < 128;
    static const char* names[2][3] = {{"gemm_f32_kernel<0, 128>", "gemm_f32_kernel<1, 128>", "gemm_f32_kernel<2, 128>"},
                                      {"gemm_f32_kernel<0, 64>", "gemm_f32_kernel<1, 64>", "gemm_f32_kernel<2, 64>"}};
    if (kname && epi >= 0 && epi < 3) *kname = names[small ? 1 : 0][epi];
    return small ? launch_f32_ts<64>(epi, g, s) : launch_f32_ts<128>(epi, g, s);
}

}  // namespace mrgan
