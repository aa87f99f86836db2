// tools/first_bench.hip -- variants of the fused first-layer + input-transform kernel (wino_first.h) on VGG-16's conv1_1 -> conv1_2 at batch 32,
// interleaved on one box (box-to-box spread is larger than most of the differences): channels per block, patch reads hoisted or not.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.h"
namespace fhip
{
__device__ __forceinline__ void bt8(float& r0, float& r1, float& r2, float& r3, float& r4, float& r5, float& r6, float& r7)
{
    const float o0 = (r0 - r6) + 5.25f * (r4 - r2);
    const float o7 = (r7 - r1) + 5.25f * (r3 - r5);
    const float t1 = (r2 + r6) - 4.25f * r4;
    const float t2 = (r1 + r5) - 4.25f * r3;
    const float p1 = r6 + (0.25f * r2 - 1.25f * r4);
    const float p2 = (0.5f * r1 - 2.5f * r3) + 2.f * r5;
    const float q1 = r6 + 4.f * (r2 - 1.25f * r4);
    const float q2 = (2.f * r1 - 2.5f * r3) + 0.5f * r5;
    r0 = o0; r1 = t1 + t2; r2 = t1 - t2; r3 = p1 + p2; r4 = p1 - p2; r5 = q1 + q2; r6 = q1 - q2; r7 = o7;
}
} // namespace fhip
#include "wino_first.h"

using namespace fhip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ void spin_kernel(long long ticks) // one wave waiting on the 100 MHz clock
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
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
    const int batch = argc > 1 ? atoi(argv[1]) : 32, reps = argc > 2 ? atoi(argv[2]) : 9;
    CK(hipEventCreate(&g_a));
    CK(hipEventCreate(&g_b));
    const int H = 224, W = 224, K = 64, TX = 38, T = 38 * 38;
    WinoFirstParams q;
    float *in, *w, *bias, *V;
    const size_t in_n = (size_t)batch * 3 * H * W, P = (size_t)T * batch, Pp = (P + 127) / 128 * 128;
    CK(hipMalloc(&in, in_n * 4));
    CK(hipMalloc(&w, K * 27 * 4));
    CK(hipMalloc(&bias, K * 4));
    CK(hipMalloc(&V, (size_t)64 * K * Pp * 4));
    std::vector<float> h(in_n);
    for (size_t i = 0; i < in_n; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    CK(hipMemcpy(in, h.data(), in_n * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < K * 27; ++i) h[i] = (float)((i * 40503u) % 201) / 500.f - 0.2f;
    CK(hipMemcpy(w, h.data(), K * 27 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data(), K * 4, hipMemcpyHostToDevice));
    q.in = in; q.w = w; q.bias = bias; q.V = V; q.K = K; q.H = H; q.W = W; q.TX = TX; q.T = T; q.P = (int)P; q.Pp = (int)Pp;
    q.in_bytes = (unsigned)(in_n * 4); q.relu = 1; q.N = batch; q.bpi = (T + kFirstTiles - 1) / kFirstTiles; q.LDW = 6 * TX + 4;
    q.rows = 6 * std::min(38, (TX + kFirstTiles - 2) / TX + 1) + 4;
    const size_t lds = (size_t)3 * q.rows * q.LDW * 4;
    const unsigned gx = (unsigned)(q.bpi * batch);
    struct Var { const char* name; std::vector<double> t; } vars[] = {{"direct (L1)", {}}, {"cpb 16 hoist", {}}, {"cpb 32 hoist", {}}, {"cpb 32 sunk", {}}, {"cpb 32 sunk 8 waves", {}}, {"cpb 64 sunk 8 waves", {}}};
    for (int r = 0; r < reps + 1; ++r)
    {
        double t[6];
        t[0] = time_us([&] { hipLaunchKernelGGL(wino_input_from_first_kernel<3>, dim3((unsigned)((P + 255) / 256), K), dim3(256), 0, 0, q); });
        t[1] = time_us([&] { hipLaunchKernelGGL((wino_input_from_first_staged_kernel<3, false, 16, true>), dim3(gx, K / 16), dim3(256), lds, 0, q); });
        t[2] = time_us([&] { hipLaunchKernelGGL((wino_input_from_first_staged_kernel<3, false, 32, true>), dim3(gx, K / 32), dim3(256), lds, 0, q); });
        t[3] = time_us([&] { hipLaunchKernelGGL((wino_input_from_first_staged_kernel<3, false, 32, false>), dim3(gx, K / 32), dim3(256), lds, 0, q); });
        t[4] = time_us([&] { hipLaunchKernelGGL((wino_input_from_first_staged_kernel<3, false, 32, false, 8>), dim3(gx, K / 32), dim3(512), lds, 0, q); });
        t[5] = time_us([&] { hipLaunchKernelGGL((wino_input_from_first_staged_kernel<3, false, 64, false, 8>), dim3(gx, K / 64), dim3(512), lds, 0, q); });
        CK(hipGetLastError());
        if (r)
            for (int i = 0; i < 6; ++i) vars[i].t.push_back(t[i]);
    }
    for (auto& v : vars)
    {
        std::sort(v.t.begin(), v.t.end());
        printf("%-14s median %7.1f us  min %7.1f\n", v.name, v.t[v.t.size() / 2], v.t[0]);
    }
    // the kernel AFTER it: a 757 MB -> 757 MB copy (what conv1_2's tile GEMM is to the memory system) launched back to back behind (a) another
    // copy, (b) the staged kernel, (c) the staged kernel + an idle gap, (d) the direct (L1-bound) form; events around the copy only
    float4* dst;
    const size_t vbytes = (size_t)64 * K * Pp * 4;
    CK(hipMalloc(&dst, vbytes));
    auto copy = [&] { hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, dst, (const float4*)V, vbytes / 16); };
    auto staged = [&] { hipLaunchKernelGGL((wino_input_from_first_staged_kernel<3, false, 16, true, 4, false>), dim3(gx, K / 16), dim3(256), lds, 0, q); };
    auto staged_xcd = [&] { hipLaunchKernelGGL((wino_input_from_first_staged_kernel<3, false, 16, true, 4, true>), dim3(gx, K / 16), dim3(256), lds, 0, q); };
    float* V2;
    CK(hipMalloc(&V2, vbytes));
    WinoFirstParams q2 = q;
    q2.V = V2;
    auto staged_other = [&] { hipLaunchKernelGGL((wino_input_from_first_staged_kernel<3, false, 16, true, 4, true>), dim3(gx, K / 16), dim3(256), lds, 0, q2); };
    auto copy_back = [&] { hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, (float4*)V, (const float4*)dst, vbytes / 16); };
    auto direct = [&] { hipLaunchKernelGGL(wino_input_from_first_kernel<3>, dim3((unsigned)((P + 255) / 256), K), dim3(256), 0, 0, q); };
    struct Seq { const char* name; std::vector<double> t; } seqs[] = {{"copy after copy", {}}, {"copy after staged", {}}, {"copy after staged + 100 us idle", {}},
                                                                       {"copy after staged, XCD-contiguous", {}}, {"copy after direct", {}}, {"staged", {}}, {"staged, XCD-contiguous", {}}, {"copy V->dst after copy dst->V (source freshly written by a streaming kernel)", {}},
                                                                       {"copy V->dst after staged wrote ANOTHER buffer", {}}};
    for (int r = 0; r < reps + 1; ++r)
    {
        double t[9];
        copy();
        t[0] = time_us(copy);
        staged();
        t[1] = time_us(copy);
        staged();
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, 10000LL);
        t[2] = time_us(copy);
        staged_xcd();
        t[3] = time_us(copy);
        direct();
        t[4] = time_us(copy);
        t[5] = time_us(staged);
        t[6] = time_us(staged_xcd);
        copy_back();
        t[7] = time_us(copy);
        staged_other();
        t[8] = time_us(copy);
        if (r)
            for (int i = 0; i < 9; ++i) seqs[i].t.push_back(t[i]);
    }
    for (auto& v : seqs)
    {
        std::sort(v.t.begin(), v.t.end());
        printf("%-40s median %7.1f us  min %7.1f\n", v.name, v.t[v.t.size() / 2], v.t[0]);
    }
    return 0;
}
