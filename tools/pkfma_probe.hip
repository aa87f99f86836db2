// tools/pkfma_probe.hip -- does gfx950 issue v_pk_fma_f32 at the rate of v_fma_f32 (i.e. twice the FMAs per cycle)?  Independent accumulator
// chains, no memory traffic; prints TFLOP/s of both forms at 1 / 2 / 4 waves per SIMD.  (Round 5: the first-layer transform is vector-ALU bound.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N>
__global__ __launch_bounds__(256) void k_scalar(float* out, float a, float b, int iters)
{
    float x[N];
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = threadIdx.x * 1e-3f + j;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b)); // (plain C is SLP-packed by hipcc)
    float s = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) s += x[j];
    if (s == 12345.f) out[0] = s;
}
template <int N>
__global__ __launch_bounds__(256) void k_packed(float* out, float a, float b, int iters)
{
    f32x2 x[N / 2];
    const f32x2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.999f};
#pragma unroll
    for (int j = 0; j < N / 2; ++j) x[j] = (f32x2){threadIdx.x * 1e-3f + j, threadIdx.x * 2e-3f + j};
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < N / 2; ++j) x[j] = __builtin_elementwise_fma(x[j], a2, b2);
    float s = 0;
#pragma unroll
    for (int j = 0; j < N / 2; ++j) s += x[j].x + x[j].y;
    if (s == 12345.f) out[0] = s;
}
template <class F>
static double run(F launch, double flops)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return flops * 5 / (ms * 1e-3) / 1e12;
}
int main()
{
    float* out;
    hipMalloc(&out, 4);
    const int iters = 20000, N = 16;
    for (int wps : {1, 2, 4})
    {
        const int blocks = 256 * wps; // 4 waves per block: wps blocks per CU
        const double flops = 2.0 * N * iters * 256.0 * blocks;
        const double ts = run([&] { hipLaunchKernelGGL(k_scalar<N>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters); }, flops);
        const double tp = run([&] { hipLaunchKernelGGL(k_packed<N>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters); }, flops);
        printf("%d wave(s) per SIMD: v_fma_f32 %.1f TFLOP/s, v_pk_fma_f32 %.1f TFLOP/s\n", wps, ts, tp);
    }
    return 0;
}
