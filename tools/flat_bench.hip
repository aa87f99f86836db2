// tools/flat_bench.hip -- measurement harness for flat_gemm.h (the cross-tile pipelined LDS-DMA GEMM) against the product's
// kernels, on the GEMM shapes of the three benchmark nets.  The baseline (and the correctness reference) is whatever
// libfeather_hip.so does for the same problem through the C-ABI; the flat variants are compiled into this tool.
//   usage: flat_bench [reps] [conv|wino|all] [rounds]
// Not part of the product.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "flat_gemm.h"
#include "wave_gemm.h"
#include "conv_gemm_policy.h"
#include "gemm_core_probe.h"

using namespace fhip;

#define CK(x)                                                                                    \
    do                                                                                           \
    {                                                                                            \
        hipError_t e = (x);                                                                      \
        if (e != hipSuccess)                                                                     \
        {                                                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);          \
            exit(1);                                                                             \
        }                                                                                        \
    } while (0)
#define CF(x)                                                                      \
    do                                                                             \
    {                                                                              \
        int rc = (x);                                                              \
        if (rc)                                                                    \
        {                                                                          \
            printf("fhip error %d (%s) at %s:%d\n", rc, fhip_last_error(), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

static int g_reps = 20;
static hipEvent_t g_a, g_b;

template <class F>
static double time_ms(F&& launch)
{
    for (int i = 0; i < 2; ++i) launch();
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

static double read_clock_mhz()
{
    unsigned long long h[2] = {0, 0}, z[2] = {0, 0};
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_flat_clock_probe), sizeof h));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_flat_clock_probe), z, sizeof z));
    return h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
}

struct Result
{
    std::string name;
    std::vector<double> ms;
    double diff = 0;
    double mhz = 0;
};

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

// max |a - b| / max |b| over n floats (device buffers)
static double compare(const float* a, const float* b, size_t n)
{
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (size_t i = 0; i < n; ++i)
    {
        const double d = std::abs((double)ha[i] - hb[i]);
        if (!(d <= worst)) worst = d; // also catches NaN
        scale = std::max(scale, (double)std::abs(hb[i]));
    }
    return worst / std::max(scale, 1e-30);
}

// ---------------------------------------------------------------------------------------------------------------------------
struct ConvCase
{
    const char* name;
    int C, K, H, S, N; // input H = W, stride, batch
};

template <class Shape, int VEC, int ABL = 0>
static void launch_flat_conv(const FlatConvParams& g0, int tpb)
{
    FlatConvParams g = g0;
    g.m_tiles = (g.K + Shape::BM - 1) / Shape::BM;
    g.n_tiles = (g.Ntot + Shape::BN - 1) / Shape::BN;
    g.tpb = tpb;
    g.batches = 1;
    const int groups = (g.n_tiles + tpb - 1) / tpb;
    hipLaunchKernelGGL((flat_gemm_kernel<Shape, FlatConvPolicy<VEC>, ABL>), dim3(g.m_tiles * groups), dim3(256), 0, 0, g);
}

template <class Shape, int VEC, int ABL = 0>
static void launch_wave_conv(const FlatConvParams& g0, int tpb)
{
    FlatConvParams g = g0;
    g.m_tiles = (g.K + 63) / 64;
    g.n_tiles = (g.Ntot + 63) / 64;
    g.tpb = tpb;
    g.batches = 1;
    const int groups = (g.n_tiles + tpb - 1) / tpb;
    hipLaunchKernelGGL((wave_gemm_kernel<Shape, FlatConvPolicy<VEC>, ABL>), dim3(g.m_tiles * groups), dim3(64), 0, 0, g);
}

// the product's register-staged kernel (gemm_core.h + conv_gemm_policy.h) instantiated here with other occupancies / stamps
template <class Shape, int MODE, int ABL = 0, int TUNE = 3>
static void launch_core_conv(const FlatConvParams& f, const float* in, float* out, const float* bias)
{
    ConvGemmParams g;
    memset(&g, 0, sizeof g);
    g.batches = 1;
    g.Wt = f.Wt;
    g.in = in;
    g.out = out;
    g.bias = bias;
    g.C = f.C;
    g.K = f.K;
    g.H = g.W = f.W;
    g.OH = f.OHW / f.OW;
    g.OW = f.OW;
    g.SH = f.SH;
    g.SW = f.SW;
    g.KH = g.KW = 1;
    g.Kd = f.C;
    g.Kp = (f.K + f.bm - 1) / f.bm * f.bm;
    g.Kdp = f.Kdp;
    g.bm = f.bm;
    g.Ntot = f.Ntot;
    g.OHW = f.OHW;
    g.HW = f.HWin;
    g.KHW = 1;
    g.has_bias = 1;
    g.relu = 1;
    g.split_k = 1;
    g.k_tiles = f.Kdp / 16;
    g.m_tiles = g.Kp / Shape::BM;
    g.n_tiles = (g.Ntot + Shape::BN - 1) / Shape::BN;
    hipLaunchKernelGGL((gemm_mfma_probe_kernel<Shape, ConvGemmPolicy<MODE>, ABL, TUNE>), dim3(g.m_tiles * g.n_tiles), dim3(Shape::THREADS), 0, 0, g);
}

static void run_conv(const ConvCase& cs, int rounds)
{
    fhip_conv_param p;
    memset(&p, 0, sizeof p);
    p.input_channels = cs.C;
    p.output_channels = cs.K;
    p.input_h = p.input_w = cs.H;
    p.kernel_h = p.kernel_w = 1;
    p.stride_h = p.stride_w = cs.S;
    p.group = 1;
    p.bias_term = 1;
    p.activation = FHIP_ACT_RELU;
    CF(fhip_conv_assign_output_dim(&p));
    size_t buf_bytes = 0, packed_bytes = 0;
    CF(fhip_conv_get_buffer_size(&p, FHIP_IM2COL, cs.N, &buf_bytes, &packed_bytes));
    const size_t in_n = (size_t)cs.N * cs.C * cs.H * cs.H, out_n = (size_t)cs.N * cs.K * p.output_h * p.output_w;
    float *in, *out_ref, *out, *w, *packed, *bias, *buf = nullptr;
    CK(hipMalloc(&in, in_n * 4));
    CK(hipMalloc(&out_ref, out_n * 4));
    CK(hipMalloc(&out, out_n * 4));
    CK(hipMalloc(&w, (size_t)cs.K * cs.C * 4));
    CK(hipMalloc(&packed, packed_bytes));
    CK(hipMalloc(&bias, cs.K * 4));
    if (buf_bytes) CK(hipMalloc(&buf, buf_bytes));
    fill_random(in, in_n, 1, 1.f);
    fill_random(w, (size_t)cs.K * cs.C, 2, 1.f / std::sqrt((float)cs.C));
    fill_random(bias, cs.K, 3, 0.1f);
    CF(fhip_conv_init(&p, FHIP_IM2COL, packed, w, nullptr));
    CK(hipDeviceSynchronize());

    FlatConvParams g;
    memset(&g, 0, sizeof g);
    g.Wt = packed;
    g.in = in;
    g.out = out;
    g.bias = bias;
    g.C = cs.C;
    g.K = cs.K;
    g.Kdp = round_up(cs.C, 16);
    g.bm = cs.K <= 64 ? 64 : 128;
    g.HWin = cs.H * cs.H;
    g.W = cs.H;
    g.OW = p.output_w;
    g.OHW = p.output_h * p.output_w;
    g.SH = g.SW = cs.S;
    g.Ntot = cs.N * g.OHW;
    g.k_tiles = g.Kdp / 16;
    g.relu = 1;
    const bool vec4 = cs.S == 1 && (g.OHW % 4) == 0;

    std::vector<std::pair<std::string, std::function<void()>>> vars;
    vars.push_back({"product (C-ABI)", [&] { CF(fhip_conv_forward(&p, FHIP_IM2COL, cs.N, out_ref, in, packed, buf, bias, nullptr)); }});
#define V(NAME, SHAPE, TPB)                                                                         \
    vars.push_back({NAME, [&, g] {                                                                  \
                        if (vec4) launch_flat_conv<SHAPE, 4>(g, TPB);                               \
                        else launch_flat_conv<SHAPE, 1>(g, TPB);                                    \
                    }})
    using S128x64d3 = FlatShape<128, 64, 2, 2, 3, 4>;
    using S128x64d4 = FlatShape<128, 64, 2, 2, 4, 3>;
    using S128x128d3 = FlatShape<128, 128, 2, 2, 3, 3>;
    using S64x128d3 = FlatShape<64, 128, 1, 4, 3, 4>;
    using S64x128d4 = FlatShape<64, 128, 1, 4, 4, 3>;
    using S64x64d4 = FlatShape<64, 64, 2, 2, 4, 4>;
    if (cs.K > 64)
    {
        V("flat 128x64 D3 tpb1", S128x64d3, 1);
        V("flat 128x64 D3 tpb2", S128x64d3, 2);
        V("flat 128x64 D3 tpb4", S128x64d3, 4);
        V("flat 128x64 D3 tpb8", S128x64d3, 8);
        V("flat 128x64 D4 tpb2", S128x64d4, 2);
        V("flat 128x64 D4 tpb4", S128x64d4, 4);
        V("flat 128x128 D3 tpb1", S128x128d3, 1);
        V("flat 128x128 D3 tpb2", S128x128d3, 2);
        V("flat 128x128 D3 tpb4", S128x128d3, 4);
    }
    else
    {
        V("flat 64x128 D3 tpb1", S64x128d3, 1);
        V("flat 64x128 D3 tpb2", S64x128d3, 2);
        V("flat 64x128 D3 tpb4", S64x128d3, 4);
        V("flat 64x128 D3 tpb8", S64x128d3, 8);
        V("flat 64x128 D4 tpb4", S64x128d4, 4);
        V("flat 64x64 D4 tpb4", S64x64d4, 4);
        V("flat 64x64 D4 tpb8", S64x64d4, 8);
    }
#undef V
    if (getenv("FLAT_WAVE"))
    {
        vars.resize(1);
#define VW(NAME, SHAPE, TPB)                                                                        \
    vars.push_back({NAME, [&, g] {                                                                  \
                        if (vec4) launch_wave_conv<SHAPE, 4>(g, TPB);                               \
                        else launch_wave_conv<SHAPE, 1>(g, TPB);                                    \
                    }})
        using W16d2 = WaveShape<16, 2, 2>;
        using W16d3 = WaveShape<16, 3, 2>;
        using W8d2 = WaveShape<8, 2, 2>;
        using W8d3 = WaveShape<8, 3, 2>;
        using W8d4 = WaveShape<8, 4, 2>;
        VW("wave BK16 D2 tpb1", W16d2, 1);
        VW("wave BK16 D2 tpb2", W16d2, 2);
        VW("wave BK16 D2 tpb4", W16d2, 4);
        VW("wave BK16 D2 tpb8", W16d2, 8);
        VW("wave BK16 D3 tpb2", W16d3, 2);
        VW("wave BK8 D2 tpb1", W8d2, 1);
        VW("wave BK8 D2 tpb4", W8d2, 4);
        VW("wave BK8 D3 tpb1", W8d3, 1);
        VW("wave BK8 D3 tpb2", W8d3, 2);
        VW("wave BK8 D3 tpb4", W8d3, 4);
        VW("wave BK8 D3 tpb8", W8d3, 8);
        VW("wave BK8 D4 tpb4", W8d4, 4);
        if (vec4)
        {
            vars.push_back({"wave BK8 D3 tpb4 noread+nostore", [&, g] { launch_wave_conv<W8d3, 4, 3>(g, 4); }});
            vars.push_back({"wave BK16 D2 tpb4 noread+nostore", [&, g] { launch_wave_conv<W16d2, 4, 3>(g, 4); }});
            vars.push_back({"wave BK8 D3 tpb4 noMFMA", [&, g] { launch_wave_conv<W8d3, 4, 4>(g, 4); }});
        }
#undef VW
    }
    if (getenv("FLAT_CORE") && vec4 && cs.K > 64)
    {
        vars.resize(1);
        using C4 = GemmShape<128, 64, 16, 2, 2, 4>;
        using C6 = GemmShape<128, 64, 16, 2, 2, 6>;
        vars.push_back({"core 128x64 occ4 tune0 (r01)", [&, g] { launch_core_conv<C4, 2, 0, 0>(g, in, out, bias); }});
        vars.push_back({"core 128x64 occ4 tune1 prio", [&, g] { launch_core_conv<C4, 2, 0, 1>(g, in, out, bias); }});
        vars.push_back({"core 128x64 occ4 tune2 bias", [&, g] { launch_core_conv<C4, 2, 0, 2>(g, in, out, bias); }});
        vars.push_back({"core 128x64 occ4 tune3", [&, g] { launch_core_conv<C4, 2, 0, 3>(g, in, out, bias); }});
        vars.push_back({"core 128x64 occ6 tune3", [&, g] { launch_core_conv<C6, 2, 0, 3>(g, in, out, bias); }});
        vars.push_back({"core 64x64 occ8 tune3", [&, g] { launch_core_conv<GemmShape<64, 64, 16, 2, 2, 8>, 2, 0, 3>(g, in, out, bias); }});
        vars.push_back({"core 128x128 occ3 tune3", [&, g] { launch_core_conv<GemmShape<128, 128, 16, 2, 2, 3>, 2, 0, 3>(g, in, out, bias); }});
        static long long zero[64][16];
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_core_timeline), zero, sizeof zero));
        launch_core_conv<GemmShape<128, 64, 16, 2, 2, 4>, 2>(g, in, out, bias);
        launch_core_conv<GemmShape<128, 64, 16, 2, 2, 4>, 2>(g, in, out, bias);
        CK(hipDeviceSynchronize());
        launch_core_conv<GemmShape<128, 64, 16, 2, 2, 4>, 2, 32, 3>(g, in, out, bias);
        CK(hipDeviceSynchronize());
        static long long tl[64][16];
        CK(hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_core_timeline), sizeof tl));
        printf("core timeline %s (k_tiles %d): cycles since block start at [setup | k-tile 0 in LDS | after k-tiles 0..3 | k-loop done | store desc | transpose written | 4 stores | end]\n", cs.name, g.k_tiles);
        for (int b = 0; b < 64; b += 3)
        {
            if (!tl[b][0]) continue;
            printf("  blk %5d :", b * 97 + 5);
            for (int i : {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15}) printf(" %6lld", tl[b][i] ? tl[b][i] - tl[b][0] : -1);
            printf("\n");
        }
    }
    if (getenv("FLAT_TIMELINE") && vec4 && cs.K > 64)
    {
        // one launch of the flat kernel with stamps, dump the sampled blocks
        static long long zero[64][4][16];
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_flat_timeline), zero, sizeof zero));
        launch_flat_conv<S128x64d3, 4, 0>(g, 1);
        launch_flat_conv<S128x64d3, 4, 0>(g, 1);
        CK(hipDeviceSynchronize());
        launch_flat_conv<S128x64d3, 4, 32>(g, 1);
        CK(hipDeviceSynchronize());
        static long long tl[64][4][16];
        CK(hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_flat_timeline), sizeof tl));
        printf("timeline %s (k_tiles %d): per sampled block, wave 0: cycles since block start at [primed | s0: landed, barrier, mfma | s1: ... | end]\n", cs.name, g.k_tiles);
        long long base0 = 0;
        for (int b = 0; b < 64; ++b)
        {
            if (!tl[b][0][0]) continue;
            if (!base0) base0 = tl[b][0][0];
            printf("  blk %5d start@%8lld :", b * 97 + 5, tl[b][0][0] - base0);
            for (int i = 1; i < 14; ++i) printf(" %6lld", tl[b][0][i] ? tl[b][0][i] - tl[b][0][0] : -1);
            printf(" | end %6lld | waves end: %lld %lld %lld\n", tl[b][0][15] - tl[b][0][0], tl[b][1][15] - tl[b][0][0], tl[b][2][15] - tl[b][0][0], tl[b][3][15] - tl[b][0][0]);
        }
    }
    if (getenv("FLAT_CLK") && vec4)
    {
        vars.resize(1);
        if (cs.K > 64)
        {
            vars.push_back({"flat 128x64 D3 tpb1 +clk", [&, g] { launch_flat_conv<S128x64d3, 4, 8>(g, 1); }});
            vars.push_back({"flat 128x128 D3 tpb1 +clk", [&, g] { launch_flat_conv<S128x128d3, 4, 8>(g, 1); }});
        }
        else
            vars.push_back({"flat 64x128 D3 tpb1 +clk", [&, g] { launch_flat_conv<S64x128d3, 4, 8>(g, 1); }});
        vars.push_back({"wave BK16 D2 tpb1 +clk", [&, g] { launch_wave_conv<WaveShape<16, 2, 2>, 4, 8>(g, 1); }});
        vars.push_back({"wave BK16 D2 tpb1 +clk noread+nostore", [&, g] { launch_wave_conv<WaveShape<16, 2, 2>, 4, 11>(g, 1); }});
        vars.push_back({"wave BK16 D2 tpb1 +clk noMFMA", [&, g] { launch_wave_conv<WaveShape<16, 2, 2>, 4, 12>(g, 1); }});
        vars.push_back({"wave BK16 D2 tpb1 +clk norequests+nostore", [&, g] { launch_wave_conv<WaveShape<16, 2, 2>, 4, 26>(g, 1); }});
        if (cs.K > 64)
        {
            vars.push_back({"flat 128x64 D3 tpb1 +clk norequests+nostore", [&, g] { launch_flat_conv<S128x64d3, 4, 26>(g, 1); }});
            vars.push_back({"flat 128x128 D3 tpb1 +clk norequests+nostore", [&, g] { launch_flat_conv<S128x128d3, 4, 26>(g, 1); }});
            vars.push_back({"flat 128x64 D3 tpb1 +clk noread+nostore", [&, g] { launch_flat_conv<S128x64d3, 4, 11>(g, 1); }});
        }
    }
    if (getenv("FLAT_ABLATE") && vec4)
    {
        vars.resize(1);
#define VA(NAME, SHAPE, TPB, ABL) vars.push_back({NAME, [&, g] { launch_flat_conv<SHAPE, 4, ABL>(g, TPB); }})
        if (cs.K > 64)
        {
            VA("128x64 D3 tpb1 full", S128x64d3, 1, 0);
            VA("128x64 D3 tpb1 noHBMread", S128x64d3, 1, 1);
            VA("128x64 D3 tpb1 nostore", S128x64d3, 1, 2);
            VA("128x64 D3 tpb1 noMFMA", S128x64d3, 1, 4);
            VA("128x64 D3 tpb1 noread+nostore", S128x64d3, 1, 3);
            VA("128x64 D3 tpb1 noread+noMFMA", S128x64d3, 1, 5);
            VA("128x64 D3 tpb1 nostore+noMFMA", S128x64d3, 1, 6);
            VA("128x64 D3 tpb4 full", S128x64d3, 4, 0);
            VA("128x64 D3 tpb4 nostore", S128x64d3, 4, 2);
            VA("128x64 D3 tpb4 noMFMA", S128x64d3, 4, 4);
            VA("128x64 D3 tpb4 noread+nostore", S128x64d3, 4, 3);
        }
        else
        {
            VA("64x128 D3 tpb1 full", S64x128d3, 1, 0);
            VA("64x128 D3 tpb1 noHBMread", S64x128d3, 1, 1);
            VA("64x128 D3 tpb1 nostore", S64x128d3, 1, 2);
            VA("64x128 D3 tpb1 noMFMA", S64x128d3, 1, 4);
            VA("64x128 D3 tpb1 noread+nostore", S64x128d3, 1, 3);
            VA("64x128 D3 tpb1 nostore+noMFMA", S64x128d3, 1, 6);
        }
#undef VA
    }
    const double flops = 2.0 * cs.K * cs.C * (double)g.Ntot;
    const double bytes = 4.0 * ((double)cs.C * cs.N * (cs.S == 1 ? g.HWin : g.OHW) + (double)cs.K * g.Ntot);
    printf("conv %-14s C%4d K%4d H%3d s%d N%3d  Ntot %7d  %.2f GFLOP  %.1f MB  (peak %.1f us, 5 TB/s %.1f us)%s\n", cs.name, cs.C, cs.K, cs.H,
           cs.S, cs.N, g.Ntot, flops / 1e9, bytes / 1e6, flops / 157.3e6, bytes / 5e6, vec4 ? "" : "  [dword gathers]");
    std::vector<Result> res(vars.size());
    for (int r = 0; r < rounds; ++r)
        for (size_t vv = 0; vv < vars.size(); ++vv)
        {
            const size_t v = r == 0 ? vv : (vv + (size_t)r * 3) % vars.size(); // round 0 in order (the reference first), then rotated
            res[v].name = vars[v].first;
            if (r == 0 && v > 0) CK(hipMemset(out, 0xff, out_n * 4));
            res[v].ms.push_back(time_ms(vars[v].second));
            { const double c = read_clock_mhz(); if (c > 0) res[v].mhz = c; }
            if (r == 0 && v > 0) res[v].diff = compare(out, out_ref, out_n);
        }
    for (auto& r : res)
    {
        const double ms = median(r.ms), best = *std::min_element(r.ms.begin(), r.ms.end());
        printf("   %-24s %8.4f ms (best %8.4f)  %7.2f TF  %5.1f%%   diff %.1e%s", r.name.c_str(), ms, best, flops / ms / 1e9,
               flops / ms / 1e9 / 157.3 * 100, r.diff, (r.diff > 1e-5 && r.name.find("no") == std::string::npos) ? "  !!WRONG" : "");
        if (r.mhz > 0) printf("   shader clock %.0f MHz -> %.1f%% of the MFMA rate at that clock", r.mhz, flops / ms / 1e9 / (157.3 * r.mhz / 2400.0) * 100);
        printf("\n");
    }
    fflush(stdout);
    (void)hipFree(in);
    (void)hipFree(out_ref);
    (void)hipFree(out);
    (void)hipFree(w);
    (void)hipFree(packed);
    (void)hipFree(bias);
    if (buf) (void)hipFree(buf);
}

// ---------------------------------------------------------------------------------------------------------------------------
struct WinoCase
{
    const char* name;
    int C, K, H, N; // 3x3 s1 p1 convolution on H x H images
};

template <class Shape, int ABL = 0>
static void launch_flat_wino(const FlatWinoParams& g0, int P, int tpb)
{
    FlatWinoParams g = g0;
    g.m_tiles = (g.K + Shape::BM - 1) / Shape::BM;
    g.n_tiles = (P + Shape::BN - 1) / Shape::BN;
    g.tpb = tpb;
    g.batches = 64;
    const int groups = (g.n_tiles + tpb - 1) / tpb;
    hipLaunchKernelGGL((flat_gemm_kernel<Shape, FlatWinoPolicy, ABL>), dim3(64 * g.m_tiles * groups), dim3(256), 0, 0, g);
}

template <class Shape, int ABL = 0>
static void launch_wave_wino(const FlatWinoParams& g0, int P, int tpb)
{
    FlatWinoParams g = g0;
    g.m_tiles = (g.K + 63) / 64;
    g.n_tiles = (P + 63) / 64;
    g.tpb = tpb;
    g.batches = 64;
    const int groups = (g.n_tiles + tpb - 1) / tpb;
    hipLaunchKernelGGL((wave_gemm_kernel<Shape, FlatWinoPolicy, ABL>), dim3(64 * g.m_tiles * groups), dim3(64), 0, 0, g);
}

static void run_wino(const WinoCase& cs, int rounds)
{
    fhip_conv_param p;
    memset(&p, 0, sizeof p);
    p.input_channels = cs.C;
    p.output_channels = cs.K;
    p.input_h = p.input_w = cs.H;
    p.kernel_h = p.kernel_w = 3;
    p.stride_h = p.stride_w = 1;
    p.pad_left = p.pad_right = p.pad_top = p.pad_bottom = 1;
    p.group = 1;
    p.bias_term = 1;
    p.activation = FHIP_ACT_RELU;
    CF(fhip_conv_assign_output_dim(&p));
    fhip_winograd_plan pl;
    CF(fhip_winograd_f63_plan(&p, cs.N, &pl));
    float *U, *V, *M, *Mref, *w;
    // FLAT_PP=<columns>: the flat / wave variants use this row pitch for V and M instead of the product's (does the pitch matter?)
    const int pp_override = getenv("FLAT_PP") ? atoi(getenv("FLAT_PP")) : 0;
    if (pp_override && (pp_override < pl.columns || pp_override % 4)) { printf("bad FLAT_PP\n"); exit(1); }
    const size_t pp_scale_num = pp_override > pl.columns_padded ? pp_override : pl.columns_padded;
    CK(hipMalloc(&U, pl.u_bytes));
    CK(hipMalloc(&V, pl.v_bytes / pl.columns_padded * pp_scale_num));
    CK(hipMalloc(&M, pl.m_bytes / pl.columns_padded * pp_scale_num));
    CK(hipMalloc(&Mref, pl.m_bytes));
    CK(hipMalloc(&w, (size_t)cs.K * cs.C * 9 * 4));
    fill_random(w, (size_t)cs.K * cs.C * 9, 5, 1.f / std::sqrt(9.f * cs.C));
    fill_random(V, pl.v_bytes / 4 / pl.columns_padded * pp_scale_num, 6, 1.f);
    CF(fhip_winograd_f63_transform_kernel(&p, U, w, nullptr));
    CK(hipMemset(Mref, 0, pl.m_bytes));
    CK(hipDeviceSynchronize());

    FlatWinoParams g;
    memset(&g, 0, sizeof g);
    g.U = U;
    g.V = V;
    g.M = M;
    g.C = cs.C;
    g.K = cs.K;
    g.Cp = pl.in_channels_padded;
    g.Kp = pl.out_channels_padded;
    g.Pp = pp_override ? pp_override : pl.columns_padded;
    g.k_tiles = g.Cp / 16;
    const int P = pl.columns;

    std::vector<std::pair<std::string, std::function<void()>>> vars;
    vars.push_back({"product (C-ABI)", [&] { CF(fhip_winograd_f63_tile_gemm(&p, cs.N, Mref, U, V, nullptr)); }});
#define V_(NAME, SHAPE, TPB) vars.push_back({NAME, [&, g] { launch_flat_wino<SHAPE>(g, P, TPB); }})
    using S128x64d3 = FlatShape<128, 64, 2, 2, 3, 4>;
    using S128x64d4 = FlatShape<128, 64, 2, 2, 4, 3>;
    using S128x96d3 = FlatShape<128, 96, 4, 1, 3, 3>;
    using S128x128d3 = FlatShape<128, 128, 2, 2, 3, 3>;
    using S64x128d3 = FlatShape<64, 128, 1, 4, 3, 4>;
    using S64x128d4 = FlatShape<64, 128, 1, 4, 4, 3>;
    if (cs.K > 64)
    {
        V_("flat 128x64 D3 tpb1", S128x64d3, 1);
        V_("flat 128x64 D3 tpb2", S128x64d3, 2);
        V_("flat 128x64 D3 tpb4", S128x64d3, 4);
        V_("flat 128x64 D4 tpb2", S128x64d4, 2);
        V_("flat 128x64 D4 tpb4", S128x64d4, 4);
        V_("flat 128x96 D3 tpb1", S128x96d3, 1);
        V_("flat 128x96 D3 tpb3", S128x96d3, 3);
        V_("flat 128x128 D3 tpb1", S128x128d3, 1);
        V_("flat 128x128 D3 tpb2", S128x128d3, 2);
    }
    else
    {
        V_("flat 64x128 D3 tpb1", S64x128d3, 1);
        V_("flat 64x128 D3 tpb2", S64x128d3, 2);
        V_("flat 64x128 D3 tpb4", S64x128d3, 4);
        V_("flat 64x128 D4 tpb4", S64x128d4, 4);
    }
#undef V_
    if (getenv("FLAT_CLK"))
    {
        vars.resize(1);
        if (cs.K > 64)
        {
            vars.push_back({"flat 128x64 D3 tpb1 +clk", [&, g] { launch_flat_wino<S128x64d3, 8>(g, P, 1); }});
            vars.push_back({"flat 128x128 D3 tpb1 +clk", [&, g] { launch_flat_wino<S128x128d3, 8>(g, P, 1); }});
            vars.push_back({"flat 128x64 D3 tpb1 +clk noread+nostore", [&, g] { launch_flat_wino<S128x64d3, 11>(g, P, 1); }});
        }
        else
            vars.push_back({"flat 64x128 D3 tpb1 +clk", [&, g] { launch_flat_wino<S64x128d3, 8>(g, P, 1); }});
        vars.push_back({"wave BK16 D2 tpb1 +clk", [&, g] { launch_wave_wino<WaveShape<16, 2, 2>, 8>(g, P, 1); }});
        vars.push_back({"wave BK16 D2 tpb1 +clk noread+nostore", [&, g] { launch_wave_wino<WaveShape<16, 2, 2>, 11>(g, P, 1); }});
    }
    else if (getenv("FLAT_WAVE"))
    {
        vars.resize(1);
#define VW(NAME, SHAPE, TPB) vars.push_back({NAME, [&, g] { launch_wave_wino<SHAPE>(g, P, TPB); }})
        using W16d2 = WaveShape<16, 2, 2>;
        using W8d2 = WaveShape<8, 2, 2>;
        using W8d3 = WaveShape<8, 3, 2>;
        using W8d4 = WaveShape<8, 4, 2>;
        VW("wave BK16 D2 tpb1", W16d2, 1);
        VW("wave BK16 D2 tpb2", W16d2, 2);
        VW("wave BK16 D2 tpb4", W16d2, 4);
        VW("wave BK8 D2 tpb1", W8d2, 1);
        VW("wave BK8 D2 tpb4", W8d2, 4);
        VW("wave BK8 D3 tpb1", W8d3, 1);
        VW("wave BK8 D3 tpb2", W8d3, 2);
        VW("wave BK8 D3 tpb4", W8d3, 4);
        VW("wave BK8 D4 tpb4", W8d4, 4);
#undef VW
    }
    const double flops = 2.0 * 64 * cs.K * cs.C * (double)P;
    const double bytes = 4.0 * 64 * ((double)cs.C + cs.K) * P;
    printf("wino %-14s C%4d K%4d H%3d N%3d  P %6d (Pp %6d)  %.2f GFLOP  %.1f MB  (peak %.1f us, 5 TB/s %.1f us)\n", cs.name, cs.C, cs.K, cs.H, cs.N,
           P, g.Pp, flops / 1e9, bytes / 1e6, flops / 157.3e6, bytes / 5e6);
    std::vector<Result> res(vars.size());
    // compare the real columns only: [xi][K][Pp] -> pack P columns of planes 0, 31, 63
    auto diff_vs_ref = [&]() {
        double worst = 0;
        for (int xi : {0, 31, 63})
        {
            const size_t plane = (size_t)cs.K * g.Pp;
            std::vector<float> a(plane), b(plane);
            CK(hipMemcpy(a.data(), M + xi * plane, plane * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), Mref + xi * plane, plane * 4, hipMemcpyDeviceToHost));
            double w2 = 0, sc = 0;
            for (int k = 0; k < cs.K; ++k)
                for (int c = 0; c < P; ++c)
                {
                    const double d = std::abs((double)a[(size_t)k * g.Pp + c] - b[(size_t)k * g.Pp + c]);
                    if (!(d <= w2)) w2 = d;
                    sc = std::max(sc, (double)std::abs(b[(size_t)k * g.Pp + c]));
                }
            worst = std::max(worst, w2 / std::max(sc, 1e-30));
        }
        return worst;
    };
    for (int r = 0; r < rounds; ++r)
        for (size_t vv = 0; vv < vars.size(); ++vv)
        {
            const size_t v = r == 0 ? vv : (vv + (size_t)r * 3) % vars.size();
            res[v].name = vars[v].first;
            if (r == 0 && v > 0) CK(hipMemset(M, 0xff, pl.m_bytes));
            res[v].ms.push_back(time_ms(vars[v].second));
            { const double c = read_clock_mhz(); if (c > 0) res[v].mhz = c; }
            if (r == 0 && v > 0 && !pp_override) res[v].diff = diff_vs_ref();
        }
    for (auto& r : res)
    {
        const double ms = median(r.ms), best = *std::min_element(r.ms.begin(), r.ms.end());
        printf("   %-24s %8.4f ms (best %8.4f)  %7.2f TF  %5.1f%%   diff %.1e%s", r.name.c_str(), ms, best, flops / ms / 1e9,
               flops / ms / 1e9 / 157.3 * 100, r.diff, r.diff > 1e-5 ? "  !!WRONG" : "");
        if (r.mhz > 0) printf("   shader clock %.0f MHz -> %.1f%% of the MFMA rate at that clock", r.mhz, flops / ms / 1e9 / (157.3 * r.mhz / 2400.0) * 100);
        printf("\n");
    }
    fflush(stdout);
    (void)hipFree(U);
    (void)hipFree(V);
    (void)hipFree(M);
    (void)hipFree(Mref);
    (void)hipFree(w);
}

int main(int argc, char** argv)
{
    g_reps = argc > 1 ? atoi(argv[1]) : 20;
    const char* what = argc > 2 ? argv[2] : "all";
    const int rounds = argc > 3 ? atoi(argv[3]) : 3;
    CK(hipEventCreate(&g_a));
    CK(hipEventCreate(&g_b));
    char name[128];
    int cus = 0, ldsb = 0;
    CF(fhip_device_info(name, sizeof name, &cus, &ldsb));
    printf("%s, %d CUs; %s; reps %d rounds %d\n", name, cus, fhip_version(), g_reps, rounds);

    const ConvCase conv[] = {
        // ResNet-50 b64 (Caffe topology)
        {"r50 2a_proj/2c", 64, 256, 56, 1, 64},   {"r50 res2a_2a", 64, 64, 56, 1, 64},     {"r50 res2b_2a", 256, 64, 56, 1, 64},
        {"r50 res3a_2a", 256, 128, 56, 2, 64},    {"r50 res3a_proj", 256, 512, 56, 2, 64},  {"r50 res3x_2c", 128, 512, 28, 1, 64},
        {"r50 res3x_2a", 512, 128, 28, 1, 64},    {"r50 res4a_2a", 512, 256, 28, 2, 64},    {"r50 res4a_proj", 512, 1024, 28, 2, 64},
        {"r50 res4x_2c", 256, 1024, 14, 1, 64},   {"r50 res4x_2a", 1024, 256, 14, 1, 64},   {"r50 res5a_2a", 1024, 512, 14, 2, 64},
        {"r50 res5a_proj", 1024, 2048, 14, 2, 64}, {"r50 res5x_2c", 512, 2048, 7, 1, 64},   {"r50 res5x_2a", 2048, 512, 7, 1, 64},
        // MobileNet-V1 b256 pointwise layers
        {"mb conv2_pw", 32, 64, 112, 1, 256},     {"mb conv3_pw", 64, 128, 56, 1, 256},     {"mb conv4_pw", 128, 128, 56, 1, 256},
        {"mb conv5_pw", 128, 256, 28, 1, 256},    {"mb conv6_pw", 256, 256, 28, 1, 256},    {"mb conv7_pw", 256, 512, 14, 1, 256},
        {"mb conv8-12_pw", 512, 512, 14, 1, 256}, {"mb conv13_pw", 512, 1024, 7, 1, 256},   {"mb conv14_pw", 1024, 1024, 7, 1, 256},
    };
    const WinoCase wino[] = {
        // VGG-16 b32
        {"vgg conv1_2", 64, 64, 224, 32},   {"vgg conv2_1", 64, 128, 112, 32},  {"vgg conv2_2", 128, 128, 112, 32}, {"vgg conv3_1", 128, 256, 56, 32},
        {"vgg conv3_2", 256, 256, 56, 32},  {"vgg conv4_1", 256, 512, 28, 32},  {"vgg conv4_2", 512, 512, 28, 32},  {"vgg conv5_x", 512, 512, 14, 32},
        // ResNet-50 b64 3x3
        {"r50 res2x_2b", 64, 64, 56, 64},   {"r50 res3x_2b", 128, 128, 28, 64}, {"r50 res4x_2b", 256, 256, 14, 64}, {"r50 res5x_2b", 512, 512, 7, 64},
    };
    const char* only = getenv("FLAT_ONLY"); // substring filter on the case name
    if (!strcmp(what, "conv") || !strcmp(what, "all"))
        for (auto& c : conv)
            if (!only || strstr(c.name, only)) run_conv(c, rounds);
    if (!strcmp(what, "wino") || !strcmp(what, "all"))
        for (auto& c : wino)
            if (!only || strstr(c.name, only)) run_wino(c, rounds);
    return 0;
}
