// tools/l2_probe.hip -- how many bytes per second can all CUs together pull out of L2 / the memory-side cache / HBM with 16-byte loads?
// Every block streams a window of `window` bytes that starts at (block * stride) % span; span = 1 MB (every XCD's 4 MB L2 holds it),
// 64 MB (memory-side cache), 2 GB (HBM).  The fused Winograd variants costed in DESIGN.md 3.4 need 6-23 TB/s of L2 -> CU traffic: this
// is the number they are priced against.  Not part of the product.   usage: l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ buf, float* __restrict__ sink, size_t span4, size_t window4, int iters)
{
    const size_t base = ((size_t)blockIdx.x * window4) % span4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it)
        for (size_t i = threadIdx.x; i < window4; i += 256 * 4)
        {
            // four independent requests per lane and trip
            const float4 a = buf[(base + i) % span4], b = buf[(base + i + 256) % span4], c = buf[(base + i + 512) % span4], d = buf[(base + i + 768) % span4];
            acc.x += a.x + b.x + c.x + d.x;
            acc.y += a.y + b.y + c.y + d.y;
            acc.z += a.z + b.z + c.z + d.z;
            acc.w += a.w + b.w + c.w + d.w;
        }
    if (acc.x + acc.y + acc.z + acc.w == 123456.789f) sink[0] = acc.x;
}

int main()
{
    const size_t total = (size_t)2 << 30;
    float4* buf;
    float* sink;
    CK(hipMalloc(&buf, total));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, total));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const size_t spans[] = {(size_t)1 << 20, (size_t)3 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)192 << 20, (size_t)2 << 30};
    for (size_t span : spans)
        for (int bpc : {4, 8})
        {
            const int blocks = 256 * bpc;
            const size_t window = 64 << 10; // 64 KB per block and pass
            const int iters = span >= ((size_t)1 << 30) ? 8 : 64;
            auto launch = [&] { hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, 0, buf, sink, span / 16, window / 16, iters); };
            launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a, 0));
            for (int r = 0; r < 5; ++r) launch();
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            const double bytes = 5.0 * blocks * (double)window * iters;
            printf("span %7.1f MB, %d blocks per CU: %8.1f GB/s pulled by the CUs\n", span / 1048576.0, bpc, bytes / ms / 1e6);
        }
    return 0;
}
