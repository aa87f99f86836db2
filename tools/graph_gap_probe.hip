// How long does one dependent kernel node of a replayed hipGraph cost?  N empty kernels in a chain, captured from a stream;
// also two kernels alternating (different code objects' entry points) and kernels with static / dynamic LDS.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_empty(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void k_lds(float* p) { __shared__ float s[12288]; s[threadIdx.x] = 1.f; __syncthreads(); if (p && threadIdx.x == 9999) p[0] = s[3]; }
__global__ void k_dyn(float* p) { extern __shared__ float d[]; d[threadIdx.x] = 1.f; __syncthreads(); if (p && threadIdx.x == 9999) p[0] = d[3]; }
int run(const char* name, int mode, int N, hipStream_t s)
{
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i)
    {
        const int m = mode == 3 ? i % 3 : mode;
        if (m == 0) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, nullptr);
        if (m == 1) hipLaunchKernelGGL(k_lds, dim3(1024), dim3(256), 0, s, nullptr);
        if (m == 2) hipLaunchKernelGGL(k_dyn, dim3(1024), dim3(256), 40000, s, nullptr);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const int reps = 20;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%-28s %d nodes: %.2f us per node (graph)\n", name, N, us / reps / N);
    // same chain eagerly
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r)
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, nullptr);
    CK(hipStreamSynchronize(s));
    const double us2 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%-28s %d launches: %.2f us per launch (eager, empty kernel)\n", name, N, us2 / reps / N);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return 0;
}
int main()
{
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    run("empty", 0, 100, s);
    run("static LDS 48 KB", 1, 100, s);
    run("dynamic LDS 40 KB", 2, 100, s);
    run("alternating", 3, 99, s);
    return 0;
}
