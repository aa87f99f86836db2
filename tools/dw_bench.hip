// tools/dw_bench.hip -- depthwise 3x3 kernels on the MobileNet-V1 shapes (batch 256): the direct (no-LDS) kernel against the
// chunk-of-planes kernel at several chunk sizes, interleaved rounds, checked against each other.  Not part of the product: it
// includes the product's translation unit to reach the kernel templates.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../feathercnn_amd/csrc/depthwise.hip"

namespace fhip
{
int fail(int c, const char* m)
{
    printf("fail: %s\n", m);
    return c;
}
int fail_hip(hipError_t e, const char* w)
{
    printf("hip fail: %s %s\n", w, hipGetErrorString(e));
    return -3;
}
StageTimer::StageTimer(int, hipStream_t) {}
StageTimer::~StageTimer() {}
int device_compute_units() { return 256; }
} // namespace fhip
using namespace fhip;

#define CK(x)                                                                            \
    do                                                                                   \
    {                                                                                    \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess)                                                             \
        {                                                                                \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

// calibration: the plain float4 stream a depthwise layer cannot beat -- reads `in4` float4, writes the first `out4` of them
__global__ __launch_bounds__(256) void dw_copy_kernel(float4* __restrict__ out, const float4* __restrict__ in, long long in4, long long out4)
{
    // four independent 16-byte requests per lane and trip (one request per lane leaves the memory pipe short of bytes in flight)
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < in4; i += 4 * stride)
    {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = in[min(i + u * stride, in4 - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
            const long long j = i + u * stride;
            if (j < out4) out[j] = v[u];
            else if (v[u].x == 123456.789f) out[0] = v[u]; // keeps the load of the part that is not written (stride-2 layers)
        }
    }
}

static hipEvent_t g_a, g_b;
template <class F>
static double time_ms(F&& f, int reps)
{
    for (int i = 0; i < 2; ++i) f(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(g_a, 0));
    for (int i = 0; i < reps; ++i) f(i + 2);
    CK(hipEventRecord(g_b, 0));
    CK(hipEventSynchronize(g_b));
    float ms;
    CK(hipEventElapsedTime(&ms, g_a, g_b));
    return ms / reps;
}

struct Case
{
    const char* name;
    int C, H, S, N;
};

int main(int argc, char** argv)
{
    // dw_bench [reps] [rounds] [cold]: cold = 1 rotates over enough buffer sets that no launch finds its tensors in the 256 MB
    // memory-side cache (what a layer inside a network sees at best partially); cold = 0 re-runs one set (small layers then live in it)
    const int reps = argc > 1 ? atoi(argv[1]) : 20, rounds = argc > 2 ? atoi(argv[2]) : 5, cold = argc > 3 ? atoi(argv[3]) : 1;
    CK(hipEventCreate(&g_a));
    CK(hipEventCreate(&g_b));
    const Case cases[] = {{"conv2_dw", 32, 112, 1, 256},  {"conv3_dw", 64, 112, 2, 256},  {"conv4_dw", 128, 56, 1, 256},  {"conv5_dw", 128, 56, 2, 256},
                          {"conv6_dw", 256, 28, 1, 256},  {"conv7_dw", 256, 28, 2, 256},  {"conv8_dw", 512, 14, 1, 256},  {"conv13_dw", 512, 14, 2, 256},
                          {"conv14_dw", 1024, 7, 1, 256}};
    double tot_bytes = 0, tot_best = 0, tot_prod = 0, net_bytes = 0, net_best = 0, net_prod = 0;
    for (auto& cs : cases)
    {
        fhip_conv_param p;
        memset(&p, 0, sizeof p);
        p.input_channels = p.output_channels = p.group = cs.C;
        p.input_h = p.input_w = cs.H;
        p.kernel_h = p.kernel_w = 3;
        p.stride_h = p.stride_w = cs.S;
        p.pad_left = p.pad_right = p.pad_top = p.pad_bottom = 1;
        p.bias_term = 1;
        p.activation = FHIP_ACT_RELU;
        p.output_h = p.output_w = (cs.H + 2 - 3) / cs.S + 1;
        const size_t in_n = (size_t)cs.N * cs.C * cs.H * cs.H, out_n = (size_t)cs.N * cs.C * p.output_h * p.output_w;
        size_t w12 = 0;
        const size_t packed_n = depthwise_packed_floats(p, &w12);
        const int NB = cold ? (int)std::max<size_t>(1, (700u << 20) / ((in_n + out_n) * 4) + 1) : 1;
        std::vector<float*> ins(NB), outs(NB);
        float *ref, *w, *packed, *bias;
        for (int k = 0; k < NB; ++k)
        {
            CK(hipMalloc(&ins[k], in_n * 4));
            CK(hipMalloc(&outs[k], out_n * 4));
        }
        CK(hipMalloc(&ref, out_n * 4));
        CK(hipMalloc(&w, cs.C * 9 * 4));
        CK(hipMalloc(&packed, packed_n * 4));
        CK(hipMalloc(&bias, cs.C * 4));
        {
            std::vector<float> h(in_n);
            unsigned s = 12345;
            for (auto& x : h)
            {
                s = s * 1664525u + 1013904223u;
                x = (s >> 8) * (2.f / 16777216.f) - 1.f;
            }
            for (int k = 0; k < NB; ++k) CK(hipMemcpy(ins[k], h.data(), in_n * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(w, h.data(), cs.C * 9 * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(bias, h.data() + 1000, cs.C * 4, hipMemcpyHostToDevice));
        }
        depthwise_init(p, packed, w, nullptr);
        CK(hipDeviceSynchronize());

        DwParams q;
        q.in = ins[0];
        q.w = packed;
        q.w12 = packed + w12;
        q.bias = bias;
        q.out = outs[0];
        q.C = cs.C;
        q.H = q.W = cs.H;
        q.OH = q.OW = p.output_h;
        q.KH = q.KW = 3;
        q.SH = q.SW = cs.S;
        q.PL = q.PT = 1;
        q.planes = cs.N * cs.C;
        q.has_bias = 1;
        q.relu = 1;
        q.planes_per_chunk = 0;
        const int HW = cs.H * cs.H;
        auto at = [&](int i) {
            DwParams r = q;
            r.in = ins[i % NB];
            r.out = outs[i % NB];
            return r;
        };
        std::vector<std::pair<std::string, std::function<void(int)>>> vars;
        depthwise_forward(p, cs.N, ref, ins[0], packed, bias, nullptr);
        vars.push_back({"product routing", [&](int i) { depthwise_forward(p, cs.N, outs[i % NB], ins[i % NB], packed, bias, nullptr); }});
        {
            const int vx = ((q.W % 4) == 0 && (q.OW % 4) == 0) ? 4 : (((q.W % 2) == 0 && (q.OW % 2) == 0) ? 2 : 1);
            const int R = cs.S == 1 ? 4 : 2;
            const int yblocks = ceil_div(q.OH, R), xvecs = q.OW / vx;
            const long long total = (long long)q.planes * yblocks * xvecs;
            const int grid = (int)std::min(8192LL, (total + 255) / 256);
            vars.push_back({"direct", [=](int i) {
                                const DwParams q = at(i);
                                if (cs.S == 1)
                                {
                                    if (vx == 4) hipLaunchKernelGGL((depthwise3x3_direct_kernel<1, 4, 4>), dim3(grid), dim3(256), 0, 0, q, yblocks, xvecs, total);
                                    else if (vx == 2) hipLaunchKernelGGL((depthwise3x3_direct_kernel<1, 2, 4>), dim3(grid), dim3(256), 0, 0, q, yblocks, xvecs, total);
                                    else hipLaunchKernelGGL((depthwise3x3_direct_kernel<1, 1, 4>), dim3(grid), dim3(256), 0, 0, q, yblocks, xvecs, total);
                                }
                                else
                                {
                                    if (vx == 4) hipLaunchKernelGGL((depthwise3x3_direct_kernel<2, 4, 2>), dim3(grid), dim3(256), 0, 0, q, yblocks, xvecs, total);
                                    else if (vx == 2) hipLaunchKernelGGL((depthwise3x3_direct_kernel<2, 2, 2>), dim3(grid), dim3(256), 0, 0, q, yblocks, xvecs, total);
                                    else hipLaunchKernelGGL((depthwise3x3_direct_kernel<2, 1, 2>), dim3(grid), dim3(256), 0, 0, q, yblocks, xvecs, total);
                                }
                            }});
        }
        if (cs.S == 1 && (cs.H == 112 || cs.H == 56))
            vars.push_back({"band kernel", [=](int i) { dw_band_launch(at(i), 0); }});
        if (cs.H == 7 || cs.H == 14 || cs.H == 28)
        {
            // (rounds 3 - 5 scanned the chunk size here; since round 6 it is dw_flat_cp(H, S), a compile-time function of the plane size)
            for (int grid : {256 * 4, 256 * 8, 1 << 30})
            {
                char nm[64];
                snprintf(nm, sizeof nm, "flat cp%d grid %s", dw_flat_cp(cs.H, cs.S), grid == (1 << 30) ? "=chunks" : (grid == 1024 ? "4/CU" : "8/CU"));
                vars.push_back({nm, [=](int i) { dw_flat_launch(at(i), grid, 0); }});
            }
        }
        {
            const long long in4 = (long long)in_n / 4, out4 = (long long)out_n / 4;
            for (int grid : {2048, 8192})
            {
                char nm[64];
                snprintf(nm, sizeof nm, "COPY calibration g%d", grid);
                vars.push_back({nm, [=](int i) { hipLaunchKernelGGL(dw_copy_kernel, dim3(grid), dim3(256), 0, 0, (float4*)outs[i % NB], (const float4*)ins[i % NB], in4, out4); }});
            }
        }
        const double bytes = 4.0 * (in_n + out_n) + 40.0 * cs.C;
        printf("%-10s C%4d H%3d s%d : %.1f MB  (%d buffer sets)\n", cs.name, cs.C, cs.H, cs.S, bytes / 1e6, NB);
        std::vector<std::vector<double>> ms(vars.size());
        std::vector<double> diff(vars.size(), 0.0);
        for (int r = 0; r < rounds; ++r)
            for (size_t vv = 0; vv < vars.size(); ++vv)
            {
                const size_t v = r == 0 ? vv : (vv + r) % vars.size();
                if (r == 0 && v > 0)
                {
                    CK(hipMemset(outs[0], 0xff, out_n * 4));
                    vars[v].second(0);
                    std::vector<float> a(out_n), b(out_n);
                    CK(hipMemcpy(a.data(), outs[0], out_n * 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(b.data(), ref, out_n * 4, hipMemcpyDeviceToHost));
                    double worst = 0;
                    for (size_t i = 0; i < out_n; ++i)
                    {
                        const double d = std::abs((double)a[i] - b[i]);
                        if (!(d <= worst)) worst = d;
                    }
                    diff[v] = worst;
                }
                ms[v].push_back(time_ms(vars[v].second, reps));
            }
        double best = 1e9;
        for (size_t v = 0; v < vars.size(); ++v)
        {
            std::sort(ms[v].begin(), ms[v].end());
            const double m = ms[v][ms[v].size() / 2];
            const bool is_copy = vars[v].first.rfind("COPY", 0) == 0;
            if (!is_copy) best = std::min(best, m);
            printf("   %-30s %8.4f ms  %7.1f GB/s  %5.1f%% of 8 TB/s   max|diff| %.1e%s\n", vars[v].first.c_str(), m, bytes / m / 1e6, bytes / m / 1e6 / 80.0,
                   diff[v], (diff[v] > 1e-5 && !is_copy) ? "  !!WRONG" : "");
        }
        const int mult = !strcmp(cs.name, "conv8_dw") ? 5 : 1; // five identical 14x14 layers in the net
        tot_bytes += bytes * mult;
        tot_best += best * mult;
        tot_prod += ms[0][ms[0].size() / 2] * mult;
        // the launches that stay depthwise kernels inside MobileNet-V1 at fusion level >= 2 (conv3/4/5 are fused into their 1x1 layers)
        if (strcmp(cs.name, "conv3_dw") && strcmp(cs.name, "conv4_dw") && strcmp(cs.name, "conv5_dw"))
        {
            net_bytes += bytes * mult;
            net_best += best * mult;
            net_prod += ms[0][ms[0].size() / 2] * mult;
        }
        fflush(stdout);
        for (int k = 0; k < NB; ++k)
        {
            (void)hipFree(ins[k]);
            (void)hipFree(outs[k]);
        }
        (void)hipFree(ref);
        (void)hipFree(w);
        (void)hipFree(packed);
        (void)hipFree(bias);
    }
    printf("MobileNet-V1 b256 depthwise total (%s): %.2f GB; product routing %.3f ms = %.1f%% of 8 TB/s; best per layer %.3f ms = %.1f%%\n", cold ? "cold" : "cache-hot",
           tot_bytes / 1e9, tot_prod, tot_bytes / tot_prod / 1e6 / 80.0, tot_best, tot_bytes / tot_best / 1e6 / 80.0);
    printf("  the 10 launches of the fused net: %.2f GB; product routing %.3f ms = %.1f%%; best per layer %.3f ms = %.1f%%\n", net_bytes / 1e9, net_prod,
           net_bytes / net_prod / 1e6 / 80.0, net_best, net_bytes / net_best / 1e6 / 80.0);
    return 0;
}
