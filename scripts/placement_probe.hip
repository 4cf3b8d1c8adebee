// Diagnostic: which blocks share a CU?  Launches G blocks of 256 threads with 72 KiB of LDS (two fit per CU) and
// records each block's (XCC, SE, SH, CU) from the hardware-id registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
    extern __shared__ char lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
    lds[threadIdx.x] = 1;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);      // keep the block resident while the grid fills
    __syncthreads();
}
int main() {
    const int G = 512;
    unsigned* d; hipMalloc((void**)&d, G * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipLaunchKernelGGL(probe, dim3(G), dim3(256), 72 * 1024, 0, d, 200);
    std::vector<unsigned> h(G * 2);
    hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < G; ++b) {
        const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
        const unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
        cu[key].push_back(b);
    }
    printf("%zu distinct CUs for %d blocks\n", cu.size(), G);
    int n = 0;
    for (auto& kv : cu) {
        if (n++ < 24) { printf("xcc %u se %u sh %u cu %2u :", kv.first >> 16, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 15); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    }
    std::map<int, int> diff;
    for (auto& kv : cu) if (kv.second.size() == 2) diff[kv.second[1] - kv.second[0]]++;
    for (auto& kv : diff) printf("block-index distance %d between the two blocks of a CU: %d CUs\n", kv.first, kv.second);
    return 0;
}
