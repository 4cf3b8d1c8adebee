// Stand-alone diagnostic: how fast can one CU pull bytes from L2 / MALL / HBM into LDS with LDS-DMA
// (buffer_load_dwordx4 ... lds), as a function of waves per block, ring depth and sharing between blocks?
// This is the resource that bounds every main loop of gemm_bf16.hip ("fill rate").
//   hipcc --offload-arch=gfx950 -O3 -o scripts/_fillbench scripts/fillbench.hip && scripts/_fillbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rs, char* dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, voff, soff, 0, 0);
}

// each wave issues IPW 1-KiB wave-instructions per tile; ring of DEPTH tiles; no compute at all.
// USE_REG: plain global_load_dwordx4 into registers + ds_write instead of LDS-DMA.
template <int NW, int IPW, int DEPTH, bool USE_REG>
__global__ __launch_bounds__(64 * NW) void fill_kernel(const char* src, long region_bytes, int share, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int TILE = NW * IPW * 1024;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // share > 0: `share` consecutive blocks read the same bytes (consecutive blocks sit on different XCDs: no L2 reuse
    // below 8 sharers).  share < 0: -share blocks OF THE SAME XCD (blockIdx % 8) read the same bytes.
    long ridx;
    if (share > 0) ridx = blockIdx.x / share;
    else { const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per = (gridDim.x / 8 + (-share) - 1) / (-share); ridx = (long)xcd * per + local / (-share); }
    const char* base = src + ridx * region_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)region_bytes, 0x00020000);
    const int voff = (wave * IPW) * 1024 + lane * 16;
    unsigned acc = 0;
    if constexpr (!USE_REG) {
        auto issue = [&](int it, int stage) {
#pragma unroll
            for (int i = 0; i < IPW; ++i) glds16(rs, lds + stage * TILE + (wave * IPW + i) * 1024, voff + i * 1024, it * TILE);
        };
#pragma unroll
        for (int p = 0; p < DEPTH - 1; ++p) issue(p, p);
        int stage = 0;
        for (int it = 0; it < iters; ++it) {
            // all but the newest DEPTH-2 groups have landed
            if (DEPTH >= 4 && it + 2 < iters) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");
            else if (DEPTH >= 3 && it + 1 < iters) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (it + DEPTH - 1 < iters) { int nb = stage + DEPTH - 1; if (nb >= DEPTH) nb -= DEPTH; issue(it + DEPTH - 1, nb); }
            acc += *(const unsigned*)(lds + stage * TILE + threadIdx.x * 4);      // touch the tile
            stage = stage + 1 == DEPTH ? 0 : stage + 1;
        }
    } else {
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        for (int it = 0; it < iters; it += 1) {
            u4 v[IPW];
#pragma unroll
            for (int i = 0; i < IPW; ++i) v[i] = *(const u4*)(base + (long)it * TILE + voff + i * 1024);
#pragma unroll
            for (int i = 0; i < IPW; ++i) *(u4*)(lds + (it & 1) * TILE + (wave * IPW + i) * 1024 + lane * 16) = v[i];
            __syncthreads();
            acc += *(const unsigned*)(lds + (it & 1) * TILE + threadIdx.x * 4);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int NW, int IPW, int DEPTH, bool USE_REG>
static void run(const char* label, const char* src, size_t src_bytes, int blocks, int share, int iters, unsigned* sink) {
    constexpr int TILE = NW * IPW * 1024;
    const long region = (long)iters * TILE;
    const int ash = share > 0 ? share : 1;
    if ((size_t)((blocks + ash - 1) / ash) * region > src_bytes) { printf("%-44s skipped (buffer)\n", label); return; }
    auto k = fill_kernel<NW, IPW, DEPTH, USE_REG>;
    const int ldsb = (USE_REG ? 2 : DEPTH) * TILE;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), ldsb, 0, src, region, share, iters, sink);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), ldsb, 0, src, region, share, iters, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double us = 1e3 * ms / reps, bytes = (double)blocks * region;
    printf("%-44s blocks %4d share %2d tile %3d KiB x %3d: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz, 256 CU)\n", label, blocks, share,
           TILE / 1024, iters, us, bytes / us * 1e-6, bytes / (us * 1e-6) / 2.4e9 / 256.0);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    char* src = nullptr; unsigned* sink = nullptr;
    if (hipMalloc((void**)&src, bytes) != hipSuccess || hipMalloc((void**)&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 1, bytes);
    // iters x tile = bytes per block; "share" blocks read the same region (L2 hits for all but the first)
    printf("== one block per CU, 4 waves, 24 KiB k-tiles (the 64x128 GEMM block), 8 k-tiles\n");
    run<4, 6, 3, false>("dma  4w depth3 unique", src, bytes, 256, 1, 8, sink);
    run<4, 6, 3, false>("dma  4w depth3 share8", src, bytes, 256, 8, 8, sink);
    run<4, 6, 3, false>("dma  4w depth3 share32", src, bytes, 256, 32, 8, sink);
    run<4, 6, 4, false>("dma  4w depth4 share8", src, bytes, 256, 8, 8, sink);
    printf("== longer streams (64 k-tiles)\n");
    run<4, 6, 3, false>("dma  4w depth3 unique", src, bytes, 256, 1, 64, sink);
    run<4, 6, 3, false>("dma  4w depth3 share8", src, bytes, 256, 8, 64, sink);
    run<4, 6, 4, false>("dma  4w depth4 share8", src, bytes, 256, 8, 64, sink);
    run<4, 6, 4, false>("dma  4w depth4 share32", src, bytes, 256, 32, 64, sink);
    run<8, 4, 3, false>("dma  8w 32KiB depth3 share8", src, bytes, 256, 8, 64, sink);
    run<8, 4, 4, false>("dma  8w 32KiB depth4 share8", src, bytes, 256, 8, 64, sink);
    run<16, 2, 4, false>("dma 16w 32KiB depth4 share8", src, bytes, 256, 8, 64, sink);
    run<4, 8, 4, false>("dma  4w 32KiB depth4 share8", src, bytes, 256, 8, 64, sink);
    printf("== sharing inside an XCD (L2 hits): -N = N blocks of one XCD read the same bytes\n");
    run<4, 6, 3, false>("dma  4w depth3 xcd-share 4", src, bytes, 256, -4, 64, sink);
    run<4, 6, 3, false>("dma  4w depth3 xcd-share 8", src, bytes, 256, -8, 64, sink);
    run<4, 6, 3, false>("dma  4w depth3 xcd-share 32", src, bytes, 256, -32, 64, sink);
    run<4, 6, 4, false>("dma  4w depth4 xcd-share 32", src, bytes, 256, -32, 64, sink);
    run<4, 6, 2, false>("dma  4w depth2 xcd-share 32", src, bytes, 256, -32, 64, sink);
    run<8, 4, 3, false>("dma  8w 32KiB depth3 xcd-share 32", src, bytes, 256, -32, 64, sink);
    run<8, 4, 4, false>("dma  8w 32KiB depth4 xcd-share 32", src, bytes, 256, -32, 64, sink);
    run<16, 2, 4, false>("dma 16w 32KiB depth4 xcd-share 32", src, bytes, 256, -32, 64, sink);
    run<4, 6, 3, false>("dma  4w depth3 xcd-share 32, 8 k-tiles", src, bytes, 256, -32, 8, sink);
    run<4, 6, 3, false>("dma  4w depth3 xcd-share 64, 2 blocks/CU", src, bytes, 512, -64, 64, sink);
    run<4, 6, 3, false>("dma  4w depth3 xcd-share 8, 2 blocks/CU", src, bytes, 512, -8, 64, sink);
    run<4, 6, 2, true>("reg  4w 24KiB xcd-share 32", src, bytes, 256, -32, 64, sink);
    run<16, 2, 2, true>("reg 16w 32KiB xcd-share 32", src, bytes, 256, -32, 64, sink);
    run<4, 6, 2, true>("reg  4w 24KiB xcd-share 64, 2 blocks/CU", src, bytes, 512, -64, 64, sink);
    printf("== two / three blocks per CU\n");
    run<4, 6, 3, false>("dma  4w depth3 share8, 2 blocks/CU", src, bytes, 512, 8, 64, sink);
    run<4, 6, 3, false>("dma  4w depth3 share16, 2 blocks/CU", src, bytes, 512, 16, 64, sink);
    run<4, 4, 3, false>("dma  4w 16KiB depth3 share8, 3 blocks/CU", src, bytes, 768, 8, 64, sink);
    run<4, 4, 2, false>("dma  4w 16KiB depth2 share8, 4 blocks/CU", src, bytes, 1024, 8, 64, sink);
    printf("== register path (global_load_dwordx4 + ds_write_b128)\n");
    run<4, 6, 2, true>("reg  4w 24KiB share8", src, bytes, 256, 8, 64, sink);
    run<8, 4, 2, true>("reg  8w 32KiB share8", src, bytes, 256, 8, 64, sink);
    run<4, 6, 2, true>("reg  4w 24KiB share8, 2 blocks/CU", src, bytes, 512, 8, 64, sink);
    run<16, 2, 2, true>("reg 16w 32KiB share8", src, bytes, 256, 8, 64, sink);
    hipFree(src); hipFree(sink);
    return 0;
}
