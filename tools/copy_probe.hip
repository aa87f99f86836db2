// tools/copy_probe.hip -- what HBM bandwidth does a plain streaming kernel reach on this chip?  (the practical roof the
// transform / depthwise / layer kernels are measured against).  Variants: one float4 per thread, grid-stride with unroll,
// nontemporal loads/stores, read-only (reduce) and write-only (fill).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy1(float4* __restrict__ d, const float4* __restrict__ s, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void copyU(float4* __restrict__ d, const float4* __restrict__ s, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride)
    {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            if (NT)
            {
                const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(&s[i + u * stride]));
                v[u] = make_float4(t[0], t[1], t[2], t[3]);
            }
            else
                v[u] = s[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            if (NT)
            {
                f32x4 t = {v[u].x, v[u].y, v[u].z, v[u].w};
                __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(&d[i + u * stride]));
            }
            else d[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) d[i] = s[i];
}
// contiguous chunk per block (like a block that owns whole rows / planes)
template <int U>
__global__ __launch_bounds__(256) void copy_chunk(float4* __restrict__ d, const float4* __restrict__ s, size_t n)
{
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = s[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) d[base + u * 256] = v[u];
}
__global__ __launch_bounds__(256) void fill1(float4* __restrict__ d, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void read1(float* __restrict__ out, const float4* __restrict__ s, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    {
        const float4 v = s[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <typename F>
static double time_ms(F f, int reps = 20)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv)
{
    for (size_t mb : {64, 256, 411, 1024, 2048})
    {
        const size_t bytes = mb << 20, n = bytes / 16;
        float4 *s, *d;
        CK(hipMalloc(&s, bytes));
        CK(hipMalloc(&d, bytes));
        CK(hipMemset(s, 1, bytes));
        const unsigned g1 = (unsigned)((n + 255) / 256);
        auto rep = [&](const char* name, double ms, double factor) { printf("  %-34s %8.3f ms  %7.1f GB/s\n", name, ms, factor * bytes / ms / 1e6); };
        printf("buffer %zu MiB (copy moves 2x)\n", mb);
        rep("copy 1 float4/thread", time_ms([&] { hipLaunchKernelGGL(copy1, dim3(g1), dim3(256), 0, 0, d, s, n); }), 2);
        rep("copy chunk U=4", time_ms([&] { hipLaunchKernelGGL((copy_chunk<4>), dim3((g1 + 3) / 4), dim3(256), 0, 0, d, s, n); }), 2);
        rep("copy chunk U=8", time_ms([&] { hipLaunchKernelGGL((copy_chunk<8>), dim3((g1 + 7) / 8), dim3(256), 0, 0, d, s, n); }), 2);
        for (unsigned g : {2048u, 4096u, 16384u})
        {
            char nm[64];
            snprintf(nm, 64, "copy gridstride U=4 grid %u", g);
            rep(nm, time_ms([&] { hipLaunchKernelGGL((copyU<4, false>), dim3(g), dim3(256), 0, 0, d, s, n); }), 2);
            snprintf(nm, 64, "copy gridstride U=4 NT grid %u", g);
            rep(nm, time_ms([&] { hipLaunchKernelGGL((copyU<4, true>), dim3(g), dim3(256), 0, 0, d, s, n); }), 2);
        }
        rep("fill (write only)", time_ms([&] { hipLaunchKernelGGL(fill1, dim3(g1), dim3(256), 0, 0, d, n); }), 1);
        rep("read only grid 4096", time_ms([&] { hipLaunchKernelGGL(read1, dim3(4096), dim3(256), 0, 0, (float*)d, s, n); }), 1);
        CK(hipFree(s));
        CK(hipFree(d));
    }
    return 0;
}
