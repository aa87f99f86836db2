// tools/ip_stream_bench.hip -- the weight-streaming InnerProduct kernel (ip_stream.h) on VGG-16's fc6 / fc7 / fc8 at batch 32: unroll depth,
// number of reduction pieces, the weight stream alone, the input re-pack and the reduce kernel on their own.  Not part of the product.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ip_stream.h"

using namespace fhip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// weights only: no MFMA, no activations -- what the load stream alone reaches
template <int UNR>
__global__ __launch_bounds__(256) void ip_weights_only_kernel(const IpStreamParams p)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mgb = (p.Kg + 3) / 4;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int s = vid / mgb, g = (vid - s * mgb) * 4 + wave;
    if (g >= p.Kg) return;
    const int q_lo = (int)((long long)s * p.KQ / p.S), q_hi = (int)((long long)(s + 1) * p.KQ / p.S);
    const f32x4* a = reinterpret_cast<const f32x4*>(p.wp) + ((size_t)g * p.KQ + q_lo) * 64 + lane;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int q = q_lo; q + UNR <= q_hi; q += UNR)
    {
        f32x4 av[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) av[u] = a[(size_t)u * 64];
#pragma unroll
        for (int u = 0; u < UNR; ++u) sum += av[u];
        a += (size_t)UNR * 64;
    }
    if (sum.x + sum.y + sum.z + sum.w == 123456.789f) p.partial[0] = sum.x;
}

static hipEvent_t g_a, g_b;
template <class F>
static double time_us(F&& f, int reps = 30)
{
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(g_a, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(g_b, 0));
    CK(hipEventSynchronize(g_b));
    float ms;
    CK(hipEventElapsedTime(&ms, g_a, g_b));
    return ms / reps * 1e3;
}

int main()
{
    CK(hipEventCreate(&g_a));
    CK(hipEventCreate(&g_b));
    struct Case { const char* name; int C, K; } cases[] = {{"fc6", 25088, 4096}, {"fc7", 4096, 4096}, {"fc8", 4096, 1000}};
    const int batch = 32;
    for (auto& cs : cases)
    {
        IpStreamParams p;
        p.K = cs.K;
        p.Kg = (cs.K + 31) / 32;
        p.KQ = (cs.C + 7) / 8;
        p.batch = batch;
        float *wp, *xq, *x, *partial, *out;
        const size_t wn = (size_t)p.Kg * p.KQ * 256;
        CK(hipMalloc(&wp, wn * 4));
        CK(hipMalloc(&xq, (size_t)p.KQ * 256 * 4));
        CK(hipMalloc(&x, (size_t)batch * cs.C * 4));
        CK(hipMalloc(&partial, (size_t)512 * cs.K * batch * 4));
        CK(hipMalloc(&out, (size_t)cs.K * batch * 4));
        CK(hipMemset(wp, 0, wn * 4));
        CK(hipMemset(xq, 0, (size_t)p.KQ * 256 * 4));
        CK(hipMemset(x, 0, (size_t)batch * cs.C * 4));
        p.wp = wp;
        p.xq = xq;
        p.partial = partial;
        const double mb = wn * 4 / 1e6;
        printf("%s: C %d K %d, %.0f MB of weights\n", cs.name, cs.C, cs.K, mb);
        printf("   input re-pack %.1f us\n", time_us([&] { hipLaunchKernelGGL(ip_pack_input_kernel, dim3((p.KQ + kIpPackOctets - 1) / kIpPackOctets), dim3(256), 0, 0, xq, x, batch, cs.C, p.KQ); }));
        for (int S : {16, 24, 32, 48, 64})
        {
            if (p.KQ / S < 8) continue;
            p.S = S;
            const unsigned grid = (unsigned)S * (unsigned)((p.Kg + 3) / 4);
            const double t2 = time_us([&] { hipLaunchKernelGGL(ip_stream_kernel<2>, dim3(grid), dim3(256), 0, 0, p); });
            const double t4 = time_us([&] { hipLaunchKernelGGL(ip_stream_kernel<4>, dim3(grid), dim3(256), 0, 0, p); });
            const double t6 = time_us([&] { hipLaunchKernelGGL(ip_stream_kernel<6>, dim3(grid), dim3(256), 0, 0, p); });
            const double rd = time_us([&] { hipLaunchKernelGGL(ip_reduce_kernel, dim3((cs.K * batch + 255) / 256), dim3(256), 0, 0, out, partial, wp, cs.K, batch, S, 1, 1); });
            const double w4 = time_us([&] { hipLaunchKernelGGL(ip_weights_only_kernel<4>, dim3(grid), dim3(256), 0, 0, p); });
            const double w8 = time_us([&] { hipLaunchKernelGGL(ip_weights_only_kernel<8>, dim3(grid), dim3(256), 0, 0, p); });
            printf("   S %3d (%5u blocks): unroll 2 / 4 / 6: %6.1f %6.1f %6.1f us   reduce %5.1f us   weights only 4 / 8: %6.1f %6.1f us (%.0f GB/s)\n", S, grid, t2,
                   t4, t6, rd, w4, w8, mb / std::min(w4, w8) * 1e3);
        }
        (void)hipFree(wp);
        (void)hipFree(xq);
        (void)hipFree(x);
        (void)hipFree(partial);
        (void)hipFree(out);
    }
    return 0;
}
