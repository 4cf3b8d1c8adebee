// bf16-in / fp32-accumulate MFMA GEMMs (v_mfma_f32_32x32x16_bf16) for gfx950.
//
// Two kernels, by where the reduction index lives in memory:
//   KC  ("k contiguous"): A[i][k], Bt[j][k] both have the reduction index innermost.
//        forward  Y = X Wt^T   (A = activations [M][K], Bt = transposed bf16 weight copy [N][K])
//        dX       dX = dY W^T  (A = dY [M][N], Bt = bf16 weight copy W [K][N]; reduction over N)
//        fragments are 16-B ds_read_b128 from an XOR-swizzled [row][64 k] image.
//   KS  ("k strided"): A[k][i], B[k][j] have the reduction index outermost.
//        dW = X^T dY (reduction over the batch rows): both operands are row-major activations, so the
//        MFMA fragments (8 consecutive k for one row/col) are gathered with the hardware transposing
//        LDS read ds_read_b64_tr_b16 from a [k][free] image padded to a 320-B row pitch.
// Block tile 128x128, BK = 64, 4 waves (2x2), each wave 64x64 = 2x2 accumulators of 32x32.
// Global->LDS goes through registers (16-B loads issued one tile ahead of the MFMAs that consume
// them) so that out-of-range rows can be zero-filled and so later revisions can fuse transforms
// into the staging path.
#include "gemm.h"

namespace mrgan {

namespace {
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int KC_TILE_BYTES = 128 * 128;        // 128 rows x 64 bf16
constexpr int KS_PITCH = 320;                   // bytes per k-row: 256 data + 64 pad (tr-read conflict-free)
constexpr int KS_TILE_BYTES = 64 * KS_PITCH;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ int kc_off(int row, int chunk) {
    // 16-B chunk `chunk` (0..7) of row `row`; (row>>1)&7 spreads the 16 rows of a ds_read_b128
    // lane group over all sixteen 16-B slots of the 256-B bank row
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kc_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char lds[2 * KC_TILE_BYTES + 2048];   // + column-sum scratch
    char* As = lds;
    char* Bs = lds + KC_TILE_BYTES;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_n = blockIdx.x, tile_m = blockIdx.y, batch = blockIdx.z;
    const int row_blk = tile_m * BM, col_blk = tile_n * BN;

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
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cidx = t + 256 * u, row = cidx >> 3, c = cidx & 7;
            const u32x4 z = {0u, 0u, 0u, 0u};
            ra[u] = (row_blk + row < g.M) ? *(const u32x4*)(A + (long)(row_blk + row) * g.a_si + k0 + c * 8) : z;
            rb[u] = (col_blk + row < g.N) ? *(const u32x4*)(B + (long)(col_blk + row) * g.b_sj + k0 + c * 8) : z;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cidx = t + 256 * u, row = cidx >> 3, c = cidx & 7;
            *(u32x4*)(As + kc_off(row, c)) = ra[u];
            *(u32x4*)(Bs + kc_off(row, c)) = rb[u];
        }
    };

    const int lr = lane & 31, lh = lane >> 5;
    load_tile(0);
    const int kc_end = (g.e.ablate & 4) ? 0 : g.K;
    for (int k0 = 0; k0 < kc_end; k0 += BK) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (k0 + BK < g.K) load_tile(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = *(const bf16x8*)(As + kc_off((wm * 2 + mi) * 32 + lr, ks * 2 + lh));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = *(const bf16x8*)(Bs + kc_off((wn * 2 + ni) * 32 + lr, ks * 2 + lh));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    __syncthreads();
    // the 128x128 bf16 output tile (32 KB) is assembled in the now-dead staging buffers
    epilogue<__bf16, EPI, 2, 2, 2, true>(acc, g, batch, 0, tile_m, row_blk, col_blk, wm, wn, lane,
                                         (float*)(lds + 2 * KC_TILE_BYTES), BN, (__bf16*)lds);
}

// transposed fragment: 8 consecutive k (rows of the LDS image) for free index f0 + (lane&15)
__device__ __forceinline__ bf16x8 ks_frag(const char* tile, int fb, int ks, int lane) {
    const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int f0 = fb + (g4 & 1) * 16;
    const int m0 = ks * 16 + (g4 >> 1) * 8;
    const char* a0 = tile + (m0 + q) * KS_PITCH + (f0 + 4 * p) * 2;
    // (the v4i16 form: per-element use of the v4bf16 form's result is mis-folded by hipcc 7.2)
    const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * KS_PITCH));
    const s16x8 t = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, t);
}

__global__ __launch_bounds__(256) void gemm_bf16_ks_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char lds[2 * KS_TILE_BYTES];
    char* As = lds;
    char* Bs = lds + KS_TILE_BYTES;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_n = blockIdx.x, tile_m = blockIdx.y;
    const int batch = blockIdx.z / g.splits, split = blockIdx.z % g.splits;
    const int row_blk = tile_m * BM, col_blk = tile_n * BN;
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

int launch_gemm_bf16(int epi, const GemmArgs& g, hipStream_t s) {
    dim3 grid(ceil_div(g.N, BN), ceil_div(g.M, BM), g.nbatch * g.splits);
    dim3 block(256);
    if (epi == EPI_SLAB) {
        if (g.a_si != 1 || g.b_sj != 1) return -3;
        hipLaunchKernelGGL(gemm_bf16_ks_kernel, grid, block, 0, s, g);
    } else {
        if (g.a_sk != 1 || g.b_sk != 1 || g.splits != 1 || (g.K % BK) != 0) return -3;
        if (epi == EPI_FWD) hipLaunchKernelGGL(gemm_bf16_kc_kernel<EPI_FWD>, grid, block, 0, s, g);
        else hipLaunchKernelGGL(gemm_bf16_kc_kernel<EPI_DX>, grid, block, 0, s, g);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_tr_probe(unsigned short* out, hipStream_t s) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, s, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace mrgan
