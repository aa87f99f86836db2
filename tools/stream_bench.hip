// stream_bench.hip -- the register-streamed 1x1 GEMM experiment (tools/experiments/stream_gemm_exp.h) against the product's implicit GEMM
// (C-ABI) on the 1x1 / stride-1 shapes of ResNet-50 b64 and MobileNet-V1 b256.   stream_bench [reps] [name filter]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "feather_hip/feather_hip.h"
#include "stream_gemm_exp.h"
#include "stream_split.h"

using namespace fhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define CF(x) do { int rc_ = (x); if (rc_) { printf("fhip error %d (%s) at %s:%d\n", rc_, fhip_last_error(), __FILE__, __LINE__); exit(1); } } while (0)

static int g_reps = 20;
static hipEvent_t g_a, g_b;
template <class F>
static double time_ms(F&& launch)
{
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(g_a, 0));
    for (int i = 0; i < g_reps; ++i) launch();
    CK(hipEventRecord(g_b, 0));
    CK(hipEventSynchronize(g_b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, g_a, g_b));
    return ms / g_reps;
}
static void fill_random(float* d, size_t n, unsigned seed, float scale)
{
    std::vector<float> h(1 << 22);
    unsigned s = seed * 2654435761u + 12345u;
    for (auto& x : h)
    {
        s = s * 1664525u + 1013904223u;
        x = ((s >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f) * scale;
    }
    for (size_t off = 0; off < n; off += h.size()) CK(hipMemcpy(d + off, h.data(), std::min(h.size(), n - off) * 4, hipMemcpyHostToDevice));
}
static double compare(const float* a, const float* b, size_t n)
{
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (size_t i = 0; i < n; ++i)
    {
        const double d = std::abs((double)ha[i] - hb[i]);
        if (!(d <= worst)) worst = d;
        scale = std::max(scale, (double)std::abs(hb[i]));
    }
    return worst / std::max(scale, 1e-30);
}
struct Case { const char* name; int C, K, H, N; };

template <int D, int WAVES, int MG, bool XCD = true, int ABL = 0>
static void launch_stream(const StreamParams& q)
{
    const int mg_blocks = (q.mgroups / MG + WAVES - 1) / WAVES;
    hipLaunchKernelGGL((stream_pw_kernel<D, WAVES, MG, XCD, ABL>), dim3((unsigned)(q.px_tiles * mg_blocks)), dim3(64 * WAVES), 0, 0, q);
}

template <int D, int SPLIT>
static void launch_split(const StreamParams& q)
{
    const int mg_blocks = q.mgroups / (4 / SPLIT);
    const size_t lds = (SPLIT == 4 ? 3 : 2) * 16 * 64 * 16;
    hipLaunchKernelGGL((stream_pw_split_kernel<D, SPLIT>), dim3((unsigned)(q.px_tiles * mg_blocks)), dim3(256), lds, 0, q);
}

template <int D>
static void launch_persistent(const StreamParams& q, int slots)
{
    // one wave per (m-group, starting pixel tile); the starting tiles are spread so that all waves are resident at once
    const int mg_blocks = (q.mgroups + 3) / 4;
    int starts = std::max(1, std::min(q.px_tiles, slots / (mg_blocks * 4)));
    StreamTiles ts;
    ts.tile_stride = starts;
    ts.tiles_per_wave = (q.px_tiles + starts - 1) / starts;
    hipLaunchKernelGGL((stream_pw_persistent_kernel<D>), dim3((unsigned)(starts * mg_blocks)), dim3(256), 0, 0, q, ts);
}

static void run(const Case& cs)
{
    if ((cs.H * cs.H) % 4) return; // the experiment needs 16-byte aligned pixel groups inside one image
    fhip_conv_param p;
    memset(&p, 0, sizeof p);
    p.input_channels = cs.C;
    p.output_channels = cs.K;
    p.input_h = p.input_w = cs.H;
    p.kernel_h = p.kernel_w = 1;
    p.stride_h = p.stride_w = 1;
    p.group = 1;
    p.bias_term = 1;
    p.activation = FHIP_ACT_RELU;
    CF(fhip_conv_assign_output_dim(&p));
    size_t buf_bytes = 0, packed_bytes = 0;
    CF(fhip_conv_get_buffer_size(&p, FHIP_IM2COL, cs.N, &buf_bytes, &packed_bytes));
    const int HW = cs.H * cs.H;
    const size_t in_n = (size_t)cs.N * cs.C * HW, out_n = (size_t)cs.N * cs.K * HW;
    float *in, *out_ref, *out, *w, *packed, *bias, *wp, *buf = nullptr;
    CK(hipMalloc(&in, in_n * 4));
    CK(hipMalloc(&out_ref, out_n * 4));
    CK(hipMalloc(&out, out_n * 4));
    CK(hipMalloc(&w, (size_t)cs.K * cs.C * 4));
    CK(hipMalloc(&wp, (size_t)cs.K * cs.C * 4));
    CK(hipMalloc(&packed, packed_bytes));
    CK(hipMalloc(&bias, cs.K * 4));
    if (buf_bytes) CK(hipMalloc(&buf, buf_bytes));
    fill_random(in, in_n, 1, 1.f);
    fill_random(w, (size_t)cs.K * cs.C, 2, 1.f / std::sqrt((float)cs.C));
    fill_random(bias, cs.K, 3, 0.1f);
    CF(fhip_conv_init(&p, FHIP_IM2COL, packed, w, nullptr));
    hipLaunchKernelGGL(stream_pack_weights, dim3((unsigned)(((size_t)cs.K * cs.C + 255) / 256)), dim3(256), 0, 0, wp, w, cs.K, cs.C);
    CK(hipDeviceSynchronize());
    StreamParams q;
    q.in = in;
    q.wp = wp;
    q.bias = bias;
    q.out = out;
    q.C = cs.C;
    q.K = cs.K;
    q.HW = HW;
    q.N = cs.N;
    q.relu = 1;
    q.total_px = (long long)cs.N * HW;
    q.mgroups = cs.K / 32;
    q.px_tiles = (int)((q.total_px + 127) / 128);
    q.mg_per_block = 4;
    const double fl = 2.0 * cs.K * cs.C * (double)q.total_px;
    printf("%-16s C %4d K %4d %3dpx b%d  (%.1f GF)\n", cs.name, cs.C, cs.K, cs.H, cs.N, fl / 1e9);
    std::vector<std::pair<std::string, std::function<void()>>> vars;
    vars.push_back({"product (C-ABI)", [&] { CF(fhip_conv_forward(&p, FHIP_IM2COL, cs.N, out_ref, in, packed, buf, bias, nullptr)); }});
    vars.push_back({"stream D8 w4", [&] { launch_stream<8, 4, 1>(q); }});
    vars.push_back({"stream D16 w4", [&] { launch_stream<16, 4, 1>(q); }});
    if (cs.C >= 32) vars.push_back({"persistent D8 3072", [&] { launch_persistent<8>(q, 3072); }});
    if (cs.C >= 32) vars.push_back({"persistent D8 2048", [&] { launch_persistent<8>(q, 2048); }});
    if (cs.C >= 64) vars.push_back({"persistent D16 3072", [&] { launch_persistent<16>(q, 3072); }});
    if (cs.C >= 32) vars.push_back({"persistent D8 6144", [&] { launch_persistent<8>(q, 6144); }});
    if (getenv("STREAM_SPLITN"))
    {
        // round 4: S-way split of the reduction over the waves of a block, any S (stream_split.h); waves = tiles * S
        vars.resize(3); // product, stream D8, stream D16
        const int Q = cs.C / 2 / 8, mgs = q.mgroups;
#define SPLITN(S_, G_) if (Q >= S_ && mgs % G_ == 0) vars.push_back({"split S" #S_ " x" #G_ " D8", [&] { launch_splitn<8, S_, G_>(q); }})
        SPLITN(2, 1); SPLITN(3, 1); SPLITN(4, 1); SPLITN(5, 1); SPLITN(6, 1); SPLITN(7, 1); SPLITN(8, 1);
        SPLITN(2, 2); SPLITN(3, 2); SPLITN(4, 2); SPLITN(2, 4);
#undef SPLITN
        printf("   wave tiles %d (x S waves on 1024 SIMDs: S=1 %.2f, 2 %.2f, 3 %.2f, 4 %.2f, 5 %.2f, 6 %.2f, 7 %.2f, 8 %.2f)\n", q.px_tiles * mgs, q.px_tiles * mgs / 1024.0,
               q.px_tiles * mgs * 2 / 1024.0, q.px_tiles * mgs * 3 / 1024.0, q.px_tiles * mgs * 4 / 1024.0, q.px_tiles * mgs * 5 / 1024.0, q.px_tiles * mgs * 6 / 1024.0,
               q.px_tiles * mgs * 7 / 1024.0, q.px_tiles * mgs * 8 / 1024.0);
    }
    std::vector<std::vector<double>> ms(vars.size());
    for (int round = 0; round < 3; ++round)
        for (size_t v = 0; v < vars.size(); ++v)
        {
            const size_t k = (v + round) % vars.size();
            ms[k].push_back(time_ms(vars[k].second));
        }
    for (size_t v = 0; v < vars.size(); ++v)
    {
        std::sort(ms[v].begin(), ms[v].end());
        const double t = ms[v][ms[v].size() / 2];
        double diff = 0;
        if (v)
        {
            CK(hipMemset(out, 0, out_n * 4));
            vars[v].second();
            CK(hipDeviceSynchronize());
            diff = compare(out, out_ref, out_n);
        }
        unsigned long long h[2] = {0, 0}, z[2] = {0, 0};
        double mhz = 0;
        if (v)
        {
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stream_clock_probe), z, sizeof z));
            vars[v].second();
            CK(hipDeviceSynchronize());
            CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stream_clock_probe), sizeof h));
            mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
        }
        printf("   %-22s %8.1f us  %6.1f TF  %5.1f %%   diff %.1e   %5.0f MHz -> %.1f %% of the peak at that clock\n", vars[v].first.c_str(), t * 1e3, fl / t / 1e9,
               fl / t / 1e9 / 157.3 * 100, diff, mhz, mhz > 0 ? fl / t / 1e9 / (157.3 * mhz / 2400.0) * 100 : 0.0);
    }
    for (float* x : {in, out_ref, out, w, wp, packed, bias, buf})
        if (x) CK(hipFree(x));
}

int main(int argc, char** argv)
{
    g_reps = argc > 1 ? atoi(argv[1]) : 20;
    const char* only = argc > 2 ? argv[2] : nullptr;
    CK(hipEventCreate(&g_a));
    CK(hipEventCreate(&g_b));
    const Case cases[] = {
        {"r50 2a_proj/2c", 64, 256, 56, 64}, {"r50 res2a_2a", 64, 64, 56, 64},   {"r50 res2b_2a", 256, 64, 56, 64},  {"r50 res3x_2c", 128, 512, 28, 64},
        {"r50 res3x_2a", 512, 128, 28, 64},  {"r50 res4x_2c", 256, 1024, 14, 64}, {"r50 res4x_2a", 1024, 256, 14, 64}, {"r50 res5x_2c", 512, 2048, 7, 64},
        {"r50 res5x_2a", 2048, 512, 7, 64},  {"mb conv4_pw", 128, 128, 56, 256}, {"mb conv6_pw", 256, 256, 28, 256}, {"mb conv8-12_pw", 512, 512, 14, 256},
        {"mb conv14_pw", 1024, 1024, 7, 256},
        {"quant b192", 512, 512, 14, 192}, {"quant b224", 512, 512, 14, 224}, {"quant b240", 512, 512, 14, 240}, {"quant b248", 512, 512, 14, 248},
        {"quant b256", 512, 512, 14, 256}, {"quant b272", 512, 512, 14, 272}, {"quant b360", 512, 512, 14, 360}, {"quant b376", 512, 512, 14, 376},
    };
    for (auto& c : cases)
        if (!only || strstr(c.name, only)) run(c);
    return 0;
}
