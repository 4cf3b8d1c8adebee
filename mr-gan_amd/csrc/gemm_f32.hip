// fp32-in / fp32-accumulate MFMA GEMM (v_mfma_f32_32x32x2_f32) -- the "logits within 1e-3" mode.
// The product is bit-for-bit a k-ordered fmaf chain per output element, so results are reproducible.
// Operands are addressed through generic (row, k) strides, so the same kernel serves the forward
// (X W), input-gradient (dY W^T) and weight-gradient (X^T dY) products of every dense layer.
//
// Block tile 128x128, BK=16, 4 waves as 2x2, each wave 64x64 = 2x2 accumulators of 32x32.
// LDS image is k-major ([k][row]) for both operands: the 32x32x2 fragments are one float per lane,
// lane l reads (row l&31, k = l>>5), i.e. 32 consecutive floats per half-wave -> conflict-free.
#include "gemm.h"

namespace mrgan {

namespace {
constexpr int BM = 128, BN = 128, BK = 16, LDT = 128;

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[2 * BK * LDT];
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

    // staging map: thread -> (element t&127 along the free dim, 8 consecutive k starting at (t>>7)*8)
    const int si = t & 127, sk = (t >> 7) * 8;
    const bool a_row_ok = (row_blk + si) < g.M;
    const bool b_col_ok = (col_blk + si) < g.N;
    const float* a_ptr = A + (long)(row_blk + si) * g.a_si;
    const float* b_ptr = B + (long)(col_blk + si) * g.b_sj;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[8], rb[8];
    auto load_tile = [&](int k0) {
        const int kloc = (k0 % g.seg_stride) + sk;      // BK divides seg_stride: a tile never straddles segments
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + sk + j;
            const bool kok = k < k_end && (kloc + j) < g.seg_rows;
            ra[j] = (a_row_ok && kok) ? a_ptr[(long)k * g.a_sk] : 0.f;
            rb[j] = (b_col_ok && kok) ? b_ptr[(long)k * g.b_sk] : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
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
                float a[2], b[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = As[(2 * kk + lh) * LDT + (wm * 2 + mi) * 32 + lr];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) b[ni] = Bs[(2 * kk + lh) * LDT + (wn * 2 + ni) * 32 + lr];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    epilogue<float, EPI, 2, 2, 2>(acc, g, batch, split, tile_m, row_blk, col_blk, wm, wn, lane, lds, BN);
}
}  // namespace

int launch_gemm_f32(int epi, const GemmArgs& g, hipStream_t s, const char** kname) {
    static const char* names[3] = {"gemm_f32_kernel<0>", "gemm_f32_kernel<1>", "gemm_f32_kernel<2>"};
    if (kname && epi >= 0 && epi < 3) *kname = names[epi];
    dim3 grid(ceil_div(g.N, BN), ceil_div(g.M, BM), g.nbatch * g.splits);
    dim3 block(256);
    switch (epi) {
        case EPI_FWD:  MRGAN_LAUNCH(gemm_f32_kernel<EPI_FWD>, grid, block, 0, s, g); break;
        case EPI_DX:   MRGAN_LAUNCH(gemm_f32_kernel<EPI_DX>, grid, block, 0, s, g); break;
        case EPI_SLAB: MRGAN_LAUNCH(gemm_f32_kernel<EPI_SLAB>, grid, block, 0, s, g); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace mrgan
