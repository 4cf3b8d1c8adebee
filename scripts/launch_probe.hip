// Diagnostic: back-to-back launch cost of kernels that do (almost) nothing, as a function of grid size, LDS size and
// kernarg size -- the floor under every short kernel of the step.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { int v[72]; float* p; };                      // ~300 bytes, like GemmArgs
__global__ void k_empty(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void k_big(const Big b) { if (b.p && b.v[71] == 12345 && b.v[3] == 7 && threadIdx.x == 9999) b.p[0] = 1.f; }
__global__ void k_lds(float* p) { extern __shared__ char lds[]; if (p && threadIdx.x == 9999) { lds[0] = 1; p[0] = lds[1]; } }
template <typename F> static float timeit(F f, int reps = 200) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return 1e3f * ms / reps;
}
int main() {
    float* d; hipMalloc((void**)&d, 64);
    Big b{}; b.p = d;
    hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int g : {1, 256, 512, 1536}) {
        printf("grid %4d x 256 threads: empty %.2f us | 300-B kernarg %.2f us | 72 KiB LDS %.2f us | 512 threads %.2f us\n", g,
               timeit([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, 0, d); }),
               timeit([&] { hipLaunchKernelGGL(k_big, dim3(g), dim3(256), 0, 0, b); }),
               timeit([&] { hipLaunchKernelGGL(k_lds, dim3(g), dim3(256), 72 * 1024, 0, d); }),
               timeit([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(512), 0, 0, d); }));
    }
    // the same inside a graph (what the step uses)
    hipStream_t s; hipStreamCreate(&s);
    for (int g : {256, 512}) {
        hipGraph_t gr; hipGraphExec_t ex;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k_lds, dim3(g), dim3(256), 72 * 1024, s, d);
        hipStreamEndCapture(s, &gr); hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipGraphLaunch(ex, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s); for (int i = 0; i < 20; ++i) hipGraphLaunch(ex, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("graph of 40 x (grid %d, 72 KiB LDS): %.2f us per kernel\n", g, 1e3f * ms / (20 * 40));
    }
    return 0;
}
