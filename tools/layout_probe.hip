// tools/layout_probe.hip -- does the distance between the 64 frequency points of a Winograd tile in memory matter?  A lane-per-tile kernel
// (64 loads, 64 stores, lanes along the tile index -- the access pattern of every transform kernel) on V / M laid out as
//   A: [xi][C][P]            (today: a wave's 64 accesses are 64 pieces of 256 B, C * P * 4 bytes apart)
//   B: [C][P / 64][xi][64]   (xi inside a 64-column group: the same 64 pieces are one contiguous 16 KB)
// on VGG-16's 112-pixel boundary (C = 128, P = 11648: 381 MB in, 381 MB out), plus a plain float4 copy of the same bytes, plus the copy
// launched right behind each variant (what the next kernel inherits).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool B>
__device__ __forceinline__ size_t at(int xi, int c, int p, int C, int P)
{
    if (B) return (((size_t)c * (P >> 6) + (p >> 6)) * 64 + xi) * 64 + (p & 63);
    return ((size_t)xi * C + c) * P + p;
}

template <bool BIN, bool BOUT>
__global__ __launch_bounds__(256) void xform_like_kernel(float* __restrict__ out, const float* __restrict__ in, int C, int P)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)C * P) return;
    const int c = (int)(idx / P), p = (int)(idx - (long long)c * P);
    float v[64];
#pragma unroll
    for (int xi = 0; xi < 64; ++xi) v[xi] = in[at<BIN>(xi, c, p, C, P)];
#pragma unroll
    for (int xi = 0; xi < 64; ++xi) out[at<BOUT>(xi, c, p, C, P)] = v[xi] + 1.f;
}

__global__ __launch_bounds__(256) void copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

static hipEvent_t g_a, g_b;
template <class F>
static double time_us(F&& f)
{
    CK(hipEventRecord(g_a, 0));
    f();
    CK(hipEventRecord(g_b, 0));
    CK(hipEventSynchronize(g_b));
    float ms;
    CK(hipEventElapsedTime(&ms, g_a, g_b));
    return ms * 1e3;
}

int main(int argc, char** argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 128, P = argc > 2 ? atoi(argv[2]) : 11648, reps = 7;
    CK(hipEventCreate(&g_a));
    CK(hipEventCreate(&g_b));
    const size_t n = (size_t)64 * C * P;
    float *a, *b, *c2, *d2;
    CK(hipMalloc(&a, n * 4));
    CK(hipMalloc(&b, n * 4));
    CK(hipMalloc(&c2, n * 4));
    CK(hipMalloc(&d2, n * 4));
    CK(hipMemset(a, 0, n * 4));
    const unsigned grid = (unsigned)(((size_t)C * P + 255) / 256);
    auto AA = [&] { hipLaunchKernelGGL((xform_like_kernel<false, false>), dim3(grid), dim3(256), 0, 0, b, a, C, P); };
    auto BB = [&] { hipLaunchKernelGGL((xform_like_kernel<true, true>), dim3(grid), dim3(256), 0, 0, b, a, C, P); };
    auto AB = [&] { hipLaunchKernelGGL((xform_like_kernel<false, true>), dim3(grid), dim3(256), 0, 0, b, a, C, P); };
    auto BA = [&] { hipLaunchKernelGGL((xform_like_kernel<true, false>), dim3(grid), dim3(256), 0, 0, b, a, C, P); };
    auto copy = [&] { hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (float4*)d2, (const float4*)c2, n / 4); };
    struct R { const char* name; std::vector<double> t; } rs[] = {{"A -> A (today)", {}}, {"B -> B", {}}, {"A -> B", {}}, {"B -> A", {}}, {"float4 copy", {}},
                                                                  {"copy behind A -> A", {}}, {"copy behind B -> B", {}}, {"copy behind copy", {}}};
    for (int r = 0; r <= reps; ++r)
    {
        double t[8];
        t[0] = time_us(AA);
        t[1] = time_us(BB);
        t[2] = time_us(AB);
        t[3] = time_us(BA);
        copy();
        t[4] = time_us(copy);
        AA();
        t[5] = time_us(copy);
        BB();
        t[6] = time_us(copy);
        copy();
        t[7] = time_us(copy);
        if (r)
            for (int i = 0; i < 8; ++i) rs[i].t.push_back(t[i]);
    }
    const double mb = 2.0 * n * 4 / 1e6;
    printf("C %d P %d: %.0f MB per launch\n", C, P, mb);
    for (auto& v : rs)
    {
        std::sort(v.t.begin(), v.t.end());
        printf("%-22s median %7.1f us  %.2f TB/s   (min %.1f)\n", v.name, v.t[v.t.size() / 2], mb / v.t[v.t.size() / 2] / 1e3 , v.t[0]);
    }
    return 0;
}
