// tools/r50_probe.hip -- block life cycle of the LDS-tiled 1x1 GEMM on ResNet-50's layer shapes (batch 64): EVERY block of one launch
// logs the 100 MHz wall clock at [start | set-up done | k-tile 0 in LDS | k-loop done | end] and the CU it ran on
// (tools/experiments/gemm_core_probe.h, ABLATE bit 6); the host groups the blocks per CU and reports
//   * the phase durations (median / p90),
//   * for how much of the launch a CU had 0, 1, 2, 3, 4+ blocks inside their k-loops (the only phase that issues MFMAs),
//   * launch ramp: first start -> last start of the first wave of blocks, last end.
// Not part of the product.   usage: r50_probe [reps]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "conv_gemm_policy.h"
#include "gemm_core_probe.h"
#include "gemm_il.h"

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
#define CF(x)                                                                                   \
    do                                                                                          \
    {                                                                                           \
        int rc = (x);                                                                           \
        if (rc)                                                                                 \
        {                                                                                       \
            printf("fhip error %d (%s) at %s:%d\n", rc, fhip_last_error(), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

static hipEvent_t g_a, g_b;
template <class F>
static double time_us(F&& f, int reps)
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

static void fill_random(float* d, size_t n, unsigned seed, float scale)
{
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (auto& x : h)
    {
        s = s * 1664525u + 1013904223u;
        x = ((s >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f) * scale;
    }
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
}

struct Case
{
    const char* name;
    int C, K, H, S, N;
};

template <class Shape, int MODE, int ABL, bool PRODUCT = false>
static void launch_core(const fhip_conv_param& p, int batch, const float* packed, const float* in, float* out, const float* bias, const float* residual = nullptr)
{
    ConvGemmParams g;
    memset(&g, 0, sizeof g);
    g.batches = 1;
    g.Wt = packed;
    g.in = in;
    g.out = out;
    g.bias = bias;
    g.C = p.input_channels;
    g.K = p.output_channels;
    g.H = p.input_h;
    g.W = p.input_w;
    g.OH = p.output_h;
    g.OW = p.output_w;
    g.SH = g.SW = p.stride_h;
    g.KH = g.KW = 1;
    g.Kd = g.C;
    g.bm = g.K <= 64 ? 64 : 128;
    g.Kp = (g.K + g.bm - 1) / g.bm * g.bm;
    g.Kdp = round_up(g.C, 16);
    g.OHW = g.OH * g.OW;
    g.HW = g.H * g.W;
    g.KHW = 1;
    g.Ntot = batch * g.OHW;
    g.has_bias = 1;
    g.relu = 1;
    g.split_k = 1;
    g.has_residual = residual != nullptr;
    g.residual_delta = residual ? reinterpret_cast<const char*>(residual) - reinterpret_cast<const char*>(out) : 0;
    g.k_tiles = g.Kdp / 16;
    g.m_tiles = g.Kp / Shape::BM;
    g.n_tiles = (g.Ntot + Shape::BN - 1) / Shape::BN;
    if (PRODUCT && Shape::WTM == 32 && Shape::WTN == 128)
    {
        // round 4: the transpose-free epilogue (gemm_core.h gemm_mfma_il_kernel)
        if constexpr (Shape::WTM == 32 && Shape::WTN == 128)
            hipLaunchKernelGGL((gemm_mfma_il_kernel<Shape, ConvGemmPolicy<MODE>>), dim3(g.m_tiles * g.n_tiles), dim3(Shape::THREADS), 0, 0, g);
    }
    else if (PRODUCT)  // the library's kernel (residual operand requested ahead of the LDS transpose) instead of the instrumented copy (per-store request)
        hipLaunchKernelGGL((gemm_mfma_kernel<Shape, ConvGemmPolicy<MODE>>), dim3(g.m_tiles * g.n_tiles), dim3(Shape::THREADS), 0, 0, g);
    else
        hipLaunchKernelGGL((gemm_mfma_probe_kernel<Shape, ConvGemmPolicy<MODE>, ABL, 3>), dim3(g.m_tiles * g.n_tiles), dim3(Shape::THREADS), 0, 0, g);
}

static double pct(std::vector<double> v, double q)
{
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end());
    return v[std::min(v.size() - 1, (size_t)(q * v.size()))];
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    CK(hipEventCreate(&g_a));
    CK(hipEventCreate(&g_b));
    const Case cases[] = {
        {"res2 64->256 @56", 64, 256, 56, 1, 64},     {"res2 256->64 @56", 256, 64, 56, 1, 64},   {"res3 128->512 @28", 128, 512, 28, 1, 64},
        {"res3 512->128 @28", 512, 128, 28, 1, 64},   {"res4 256->1024 @14", 256, 1024, 14, 1, 64}, {"res4 1024->256 @14", 1024, 256, 14, 1, 64},
        {"res5 512->2048 @7", 512, 2048, 7, 1, 64},   {"res3a proj 256->512 s2", 256, 512, 56, 2, 64},
        {"res5 2048->512 @7", 2048, 512, 7, 1, 64},   {"res2a 64->64 @56", 64, 64, 56, 1, 64},       {"res3a 2a 256->128 s2", 256, 128, 56, 2, 64},
        {"res4a proj 512->1024 s2", 512, 1024, 28, 2, 64}, {"res4a 2a 512->256 s2", 512, 256, 28, 2, 64}, {"res5a proj 1024->2048 s2", 1024, 2048, 14, 2, 64},
        {"res5a 2a 1024->512 s2", 1024, 512, 14, 2, 64}, {"res2a proj 64->256 @56", 64, 256, 56, 1, 64},
    };
    long long* log_d = nullptr;
    const size_t max_blocks = 1 << 16;
    CK(hipMalloc(&log_d, max_blocks * 8 * sizeof(long long)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_core_blocklog), &log_d, sizeof log_d));
    long long* iter_d = nullptr;
    CK(hipMalloc(&iter_d, max_blocks * 64 * sizeof(long long)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_core_iterlog), &iter_d, sizeof iter_d));
    for (auto& cs : cases)
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
        float *in, *out, *w, *packed, *bias, *buf = nullptr;
        CK(hipMalloc(&in, in_n * 4));
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
        const bool small = cs.K <= 64;
        const bool vec = cs.S == 1 && (p.output_h * p.output_w) % 4 == 0;
        const double flops = 2.0 * cs.K * cs.C * p.output_h * p.output_w * cs.N;
        const double t_prod = time_us([&] { CF(fhip_conv_forward(&p, FHIP_IM2COL, cs.N, out, in, packed, buf, bias, nullptr)); }, reps);
        using Big = GemmShape<128, 64, 16, 2, 2, 4>;
        using SmallM = GemmShape<64, 128, 16, 1, 4, 4>;
        auto core = [&](bool log) {
            if (small)
            {
                if (vec) { if (log) launch_core<SmallM, 2, 192>(p, cs.N, packed, in, out, bias); else launch_core<SmallM, 2, 0>(p, cs.N, packed, in, out, bias); }
                else { if (log) launch_core<SmallM, 1, 192>(p, cs.N, packed, in, out, bias); else launch_core<SmallM, 1, 0>(p, cs.N, packed, in, out, bias); }
            }
            else
            {
                if (vec) { if (log) launch_core<Big, 2, 192>(p, cs.N, packed, in, out, bias); else launch_core<Big, 2, 0>(p, cs.N, packed, in, out, bias); }
                else { if (log) launch_core<Big, 1, 192>(p, cs.N, packed, in, out, bias); else launch_core<Big, 1, 0>(p, cs.N, packed, in, out, bias); }
            }
        };
        const double t_core = time_us([&] { core(false); }, reps);
        if (getenv("PROBE_RESIDUAL") && !small && vec)
        {
            // the fused residual add (ResNet's expand layers): residual operand requested per store (the instrumented copy = the round-2 epilogue)
            // against requested ahead of the LDS transpose (the library's kernel); interleaved rounds, median of 7
            float* res = nullptr;
            CK(hipMalloc(&res, out_n * 4));
            fill_random(res, std::min<size_t>(out_n, 1 << 24), 5, 1.f);
            using Sh = GemmShape<128, 64, 16, 2, 2, 4>;
            std::vector<double> ta, tb, tc;
            for (int r = 0; r < 7; ++r)
            {
                ta.push_back(time_us([&] { launch_core<Sh, 2, 0, false>(p, cs.N, packed, in, out, bias, res); }, reps));
                tb.push_back(time_us([&] { launch_core<Sh, 2, 0, true>(p, cs.N, packed, in, out, bias, res); }, reps));
                tc.push_back(time_us([&] { launch_core<Sh, 2, 0, true>(p, cs.N, packed, in, out, bias, nullptr); }, reps));
            }
            std::sort(ta.begin(), ta.end());
            std::sort(tb.begin(), tb.end());
            std::sort(tc.begin(), tc.end());
            printf("      residual add fused: per-store request %.1f us, requested ahead %.1f us; without residual %.1f us\n", ta[3], tb[3], tc[3]);
            (void)hipFree(res);
        }
        if (getenv("PROBE_SHAPES"))
        {
            // tile-shape scan on this layer: same main loop, same packed weights (a tile never crosses a 64- / 128-row weight panel);
            // interleaved rounds, median of 5 (box clocks drift by several % within a run)
            std::vector<std::pair<std::string, std::function<void()>>> vars;
            std::vector<int> nblk;
            auto shape = [&](const char* nm, auto tag, int bmv, int bnv) {
                using Sh = decltype(tag);
                if (bmv > (small ? 64 : 128)) return;
                vars.push_back({nm, [&, vec] { if (vec) launch_core<Sh, 2, 0>(p, cs.N, packed, in, out, bias); else launch_core<Sh, 1, 0>(p, cs.N, packed, in, out, bias); }});
                nblk.push_back((round_up(cs.K, bmv) / bmv) * ceil_div(cs.N * p.output_h * p.output_w, bnv));
            };
            shape("128x64  4 waves 2x2 (64x32 each)", GemmShape<128, 64, 16, 2, 2, 4>(), 128, 64);
            shape("128x128 4 waves 2x2 (64x64 each)", GemmShape<128, 128, 16, 2, 2, 4>(), 128, 128);
            shape("64x128  4 waves 1x4 (64x32 each)", GemmShape<64, 128, 16, 1, 4, 4>(), 64, 128);
            shape("64x64   2 waves 1x2 (64x32 each)", GemmShape<64, 64, 16, 1, 2, 5>(), 64, 64);
            shape("64x64   4 waves 2x2 (32x32 each)", GemmShape<64, 64, 16, 2, 2, 8>(), 64, 64);
            shape("128x32  2 waves 2x1 (64x32 each)", GemmShape<128, 32, 16, 2, 1, 5>(), 128, 32);
            shape("64x256  4 waves 1x4 (64x64 each)", GemmShape<64, 256, 16, 1, 4, 3>(), 64, 256);
            auto il = [&](const char* nm, auto tag, int bmv, int bnv) {
                using Sh = decltype(tag);
                if (bmv > (small ? 64 : 128)) return;
                vars.push_back({nm, [&, vec] { if (vec) launch_core<Sh, 2, 0, true>(p, cs.N, packed, in, out, bias); else launch_core<Sh, 1, 0, true>(p, cs.N, packed, in, out, bias); }});
                nblk.push_back((round_up(cs.K, bmv) / bmv) * ceil_div(cs.N * p.output_h * p.output_w, bnv));
            };
            il("IL 128x128 4 waves 4x1 (32x128 each)", GemmShape<128, 128, 16, 4, 1, 4>(), 128, 128);
            il("IL 64x128  2 waves 2x1 (32x128 each)", GemmShape<64, 128, 16, 2, 1, 8>(), 64, 128);
            il("IL 64x256  4 waves 2x2 (32x128 each)", GemmShape<64, 256, 16, 2, 2, 4>(), 64, 256);
            il("IL 128x256 8 waves 4x2 (32x128 each)", GemmShape<128, 256, 16, 4, 2, 2>(), 128, 256);
            if (!small) vars.push_back({"128x64  4 waves, loads 2 tiles deep", [&, vec] { if (vec) launch_core<GemmShape<128, 64, 16, 2, 2, 4>, 2, 256>(p, cs.N, packed, in, out, bias); else launch_core<GemmShape<128, 64, 16, 2, 2, 4>, 1, 256>(p, cs.N, packed, in, out, bias); }}), nblk.push_back(0);
            vars.push_back({"product (C-ABI)", [&] { CF(fhip_conv_forward(&p, FHIP_IM2COL, cs.N, out, in, packed, buf, bias, nullptr)); }});
            nblk.push_back(0);
            {
                // correctness of every variant against the product's result (bit-identical: same k order per output)
                std::vector<float> ref(out_n), got(out_n);
                CF(fhip_conv_forward(&p, FHIP_IM2COL, cs.N, out, in, packed, buf, bias, nullptr));
                CK(hipMemcpy(ref.data(), out, out_n * 4, hipMemcpyDeviceToHost));
                for (size_t v = 0; v + 1 < vars.size(); ++v)
                {
                    CK(hipMemset(out, 0xff, out_n * 4));
                    vars[v].second();
                    CK(hipMemcpy(got.data(), out, out_n * 4, hipMemcpyDeviceToHost));
                    double worst = 0;
                    for (size_t i = 0; i < out_n; ++i) worst = std::max(worst, (double)std::abs(got[i] - ref[i]));
                    if (!(worst <= 1e-5)) printf("      !! %s differs from the product: max |diff| %.3e\n", vars[v].first.c_str(), worst);
                }
            }
            std::vector<std::vector<double>> ts(vars.size());
            for (int r = 0; r < 5; ++r)
                for (size_t v = 0; v < vars.size(); ++v)
                {
                    const size_t k = (v + r) % vars.size();
                    ts[k].push_back(time_us(vars[k].second, reps));
                }
            for (size_t v = 0; v < vars.size(); ++v)
            {
                std::sort(ts[v].begin(), ts[v].end());
                const double t = ts[v][2];
                printf("      %-36s %7.1f us (min %.1f)  %5.1f%%   %6d blocks = %5.2f per CU\n", vars[v].first.c_str(), t, ts[v][0], flops / t / 1e6 / 1.573, nblk[v], nblk[v] / 256.0);
            }
        }
        const int bm = small ? 64 : 128, bn = small ? 128 : 64;
        const int blocks = (round_up(cs.K, bm) / bm) * ceil_div(cs.N * p.output_h * p.output_w, bn);
        printf("%-24s product %.1f us (%.1f TF = %.1f%%), LDS-tiled kernel alone %.1f us (%.1f%%), %d blocks = %.2f per CU, %d k-tiles\n", cs.name, t_prod,
               flops / t_prod / 1e6, flops / t_prod / 1e6 / 1.573, t_core, flops / t_core / 1e6 / 1.573, blocks, blocks / 256.0, round_up(cs.C, 16) / 16);
        if ((size_t)blocks <= max_blocks)
        {
            CK(hipMemset(log_d, 0, (size_t)blocks * 8 * sizeof(long long)));
            core(false);
            core(false);
            CK(hipDeviceSynchronize());
            core(true);
            CK(hipDeviceSynchronize());
            std::vector<long long> lg((size_t)blocks * 8);
            CK(hipMemcpy(lg.data(), log_d, lg.size() * sizeof(long long), hipMemcpyDeviceToHost));
            long long t0 = lg[1], t1 = lg[5];
            for (int b = 0; b < blocks; ++b)
            {
                t0 = std::min(t0, lg[b * 8 + 1]);
                t1 = std::max(t1, lg[b * 8 + 5]);
            }
            std::map<long long, std::vector<int>> cus;
            std::vector<double> setup, fill, kloop, epi, life;
            for (int b = 0; b < blocks; ++b)
            {
                const long long* e = &lg[b * 8];
                const long long cu = ((e[0] >> 32) << 16) | (e[0] & 0xff00); // XCC id | SE, SH, CU bits of HW_ID
                cus[cu].push_back(b);
                setup.push_back((e[2] - e[1]) * 0.01);
                fill.push_back((e[3] - e[2]) * 0.01);
                kloop.push_back((e[4] - e[3]) * 0.01);
                epi.push_back((e[5] - e[4]) * 0.01);
                life.push_back((e[5] - e[1]) * 0.01);
            }
            // per CU: time with n blocks in their k-loop
            double occ[6] = {0, 0, 0, 0, 0, 0}, resid[10] = {0};
            for (auto& kv : cus)
            {
                std::vector<std::pair<long long, int>> ev, ev2;
                for (int b : kv.second)
                {
                    ev.push_back({lg[b * 8 + 3], +1});
                    ev.push_back({lg[b * 8 + 4], -1});
                    ev2.push_back({lg[b * 8 + 1], +1});
                    ev2.push_back({lg[b * 8 + 5], -1});
                }
                std::sort(ev.begin(), ev.end());
                std::sort(ev2.begin(), ev2.end());
                long long prev = t0;
                int n = 0;
                for (auto& e : ev)
                {
                    occ[std::min(n, 5)] += (e.first - prev);
                    prev = e.first;
                    n += e.second;
                }
                occ[0] += t1 - prev;
                prev = t0;
                n = 0;
                for (auto& e : ev2)
                {
                    resid[std::min(n, 9)] += (e.first - prev);
                    prev = e.first;
                    n += e.second;
                }
                resid[0] += t1 - prev;
            }
            const double tot = (double)(t1 - t0) * cus.size();
            printf("   launch %.1f us over %zu CUs; phases us median/p90: set-up %.2f/%.2f  k-tile 0 -> LDS %.2f/%.2f  k-loop %.2f/%.2f  epilogue %.2f/%.2f  life %.2f/%.2f\n",
                   (t1 - t0) * 0.01, cus.size(), pct(setup, .5), pct(setup, .9), pct(fill, .5), pct(fill, .9), pct(kloop, .5), pct(kloop, .9), pct(epi, .5), pct(epi, .9),
                   pct(life, .5), pct(life, .9));
            printf("   CU time with n blocks inside their k-loop: 0: %.1f%%  1: %.1f%%  2: %.1f%%  3: %.1f%%  4: %.1f%%  5+: %.1f%%   | resident blocks: 0: %.1f%% 1: %.1f%% 2: %.1f%% 3: %.1f%% 4: %.1f%% 5+: %.1f%%\n",
                   100 * occ[0] / tot, 100 * occ[1] / tot, 100 * occ[2] / tot, 100 * occ[3] / tot, 100 * occ[4] / tot, 100 * occ[5] / tot, 100 * resid[0] / tot,
                   100 * resid[1] / tot, 100 * resid[2] / tot, 100 * resid[3] / tot, 100 * resid[4] / tot,
                   100 * (resid[5] + resid[6] + resid[7] + resid[8] + resid[9]) / tot);
            if (getenv("PROBE_DUMP"))
            {
                int shown = 0;
                for (auto& kv : cus)
                {
                    if (shown++ % 100 != 7) continue; // a few CUs
                    std::vector<int> bs = kv.second;
                    std::sort(bs.begin(), bs.end(), [&](int a, int b) { return lg[a * 8 + 1] < lg[b * 8 + 1]; });
                    printf("   CU %llx: blocks (start, k-loop begin, k-loop end, end) us:", (unsigned long long)kv.first);
                    for (int b : bs) printf("  [%.1f %.1f %.1f %.1f]", (lg[b * 8 + 1] - t0) * 0.01, (lg[b * 8 + 3] - t0) * 0.01, (lg[b * 8 + 4] - t0) * 0.01, (lg[b * 8 + 5] - t0) * 0.01);
                    printf("\n");
                    if (getenv("PROBE_ITERS"))
                    {
                        std::vector<long long> it((size_t)blocks * 64);
                        CK(hipMemcpy(it.data(), iter_d, it.size() * sizeof(long long), hipMemcpyDeviceToHost));
                        const int kt = round_up(cs.C, 16) / 16;
                        for (int b : bs)
                        {
                            printf("      blk %5d k-tile end times:", b);
                            for (int k = 0; k < std::min(kt, 64); ++k) printf(" %.1f", (it[(size_t)b * 64 + k] - t0) * 0.01);
                            printf("\n");
                        }
                    }
                }
            }
            // the k-loop alone at the MFMA rate: MFMAs per wave * 64 clk at 2.4 GHz
            const int tm = small ? 2 : 2, tn = 1;
            const double kloop_ideal = (double)tm * tn * 8 * (round_up(cs.C, 16) / 16) * 64 / 2400.0;
            printf("   k-loop of one wave alone on its SIMD at 2.4 GHz: %.2f us\n", kloop_ideal);
        }
        fflush(stdout);
        (void)hipFree(in);
        (void)hipFree(out);
        (void)hipFree(w);
        (void)hipFree(packed);
        (void)hipFree(bias);
        if (buf) (void)hipFree(buf);
    }
    return 0;
}
