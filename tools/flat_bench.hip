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

struct Result
{
    std::string name;
    std::vector<double> ms;
    double diff = 0;
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
    using S128x64d3 = FlatShape<128, 64, 2, 2, 3, 3>;
    using S128x64d4 = FlatShape<128, 64, 2, 2, 4, 2>;
    using S128x128d3 = FlatShape<128, 128, 2, 2, 3, 2>;
    using S64x128d3 = FlatShape<64, 128, 1, 4, 3, 3>;
    using S64x128d4 = FlatShape<64, 128, 1, 4, 4, 2>;
    using S64x64d4 = FlatShape<64, 64, 2, 2, 4, 3>;
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
        for (size_t v = 0; v < vars.size(); ++v)
        {
            res[v].name = vars[v].first;
            if (r == 0 && v > 0) CK(hipMemset(out, 0xff, out_n * 4));
            res[v].ms.push_back(time_ms(vars[v].second));
            if (r == 0 && v > 0) res[v].diff = compare(out, out_ref, out_n);
        }
    for (auto& r : res)
    {
        const double ms = median(r.ms), best = *std::min_element(r.ms.begin(), r.ms.end());
        printf("   %-24s %8.4f ms (best %8.4f)  %7.2f TF  %5.1f%%   diff %.1e%s\n", r.name.c_str(), ms, best, flops / ms / 1e9,
               flops / ms / 1e9 / 157.3 * 100, r.diff, (r.diff > 1e-5 && !getenv("FLAT_ABLATE")) ? "  !!WRONG" : "");
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

template <class Shape>
static void launch_flat_wino(const FlatWinoParams& g0, int P, int tpb)
{
    FlatWinoParams g = g0;
    g.m_tiles = (g.K + Shape::BM - 1) / Shape::BM;
    g.n_tiles = (P + Shape::BN - 1) / Shape::BN;
    g.tpb = tpb;
    g.batches = 64;
    const int groups = (g.n_tiles + tpb - 1) / tpb;
    hipLaunchKernelGGL((flat_gemm_kernel<Shape, FlatWinoPolicy>), dim3(64 * g.m_tiles * groups), dim3(256), 0, 0, g);
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
    CK(hipMalloc(&U, pl.u_bytes));
    CK(hipMalloc(&V, pl.v_bytes));
    CK(hipMalloc(&M, pl.m_bytes));
    CK(hipMalloc(&Mref, pl.m_bytes));
    CK(hipMalloc(&w, (size_t)cs.K * cs.C * 9 * 4));
    fill_random(w, (size_t)cs.K * cs.C * 9, 5, 1.f / std::sqrt(9.f * cs.C));
    fill_random(V, pl.v_bytes / 4, 6, 1.f);
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
    g.Pp = pl.columns_padded;
    g.k_tiles = g.Cp / 16;
    const int P = pl.columns;

    std::vector<std::pair<std::string, std::function<void()>>> vars;
    vars.push_back({"product (C-ABI)", [&] { CF(fhip_winograd_f63_tile_gemm(&p, cs.N, Mref, U, V, nullptr)); }});
#define V_(NAME, SHAPE, TPB) vars.push_back({NAME, [&, g] { launch_flat_wino<SHAPE>(g, P, TPB); }})
    using S128x64d3 = FlatShape<128, 64, 2, 2, 3, 3>;
    using S128x64d4 = FlatShape<128, 64, 2, 2, 4, 2>;
    using S128x96d3 = FlatShape<128, 96, 4, 1, 3, 3>;
    using S128x128d3 = FlatShape<128, 128, 2, 2, 3, 2>;
    using S64x128d3 = FlatShape<64, 128, 1, 4, 3, 3>;
    using S64x128d4 = FlatShape<64, 128, 1, 4, 4, 2>;
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
        for (size_t v = 0; v < vars.size(); ++v)
        {
            res[v].name = vars[v].first;
            if (r == 0 && v > 0) CK(hipMemset(M, 0xff, pl.m_bytes));
            res[v].ms.push_back(time_ms(vars[v].second));
            if (r == 0 && v > 0) res[v].diff = diff_vs_ref();
        }
    for (auto& r : res)
    {
        const double ms = median(r.ms), best = *std::min_element(r.ms.begin(), r.ms.end());
        printf("   %-24s %8.4f ms (best %8.4f)  %7.2f TF  %5.1f%%   diff %.1e%s\n", r.name.c_str(), ms, best, flops / ms / 1e9,
               flops / ms / 1e9 / 157.3 * 100, r.diff, r.diff > 1e-5 ? "  !!WRONG" : "");
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
