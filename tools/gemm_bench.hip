// tools/gemm_bench.hip -- measurement harness for the shared MFMA main loop (gemm_core.h) on Winograd tile-GEMM
// shapes: times variants of tile shape / occupancy / ablations with HIP events. Not part of the product.
//   usage: gemm_bench [reps]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <cmath>
#include <algorithm>

#include "gemm_core.h"
#include "gemm_core_probe.h"
#include "wino_gemm_policy.h"
#include "wino_gemm_glds.h"
#include "wino_gemm_glds128.h"
#include "wino_gemm_c64.h"

using namespace fhip;
namespace fhip
{
int fail(int c, const char*) { return c; }
int fail_hip(hipError_t, const char*) { return -3; }
StageTimer::StageTimer(int, hipStream_t) {}
StageTimer::~StageTimer() {}
} // namespace fhip

#define CK(x)                                                                  \
    do                                                                         \
    {                                                                          \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess)                                                   \
        {                                                                      \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

struct Case
{
    int C, K, P;
};

static int g_cus = 256;
static int g_batches = 64;
static int g_ref_pp = 0;
static int g_bp = 0; // column block of V / M (GEMM_BP)
static int g_split_rounds = 0; // GEMM_SPLIT: wino_gemm_row_split's max_rounds for V0 == 3 (0 = whole tiles only)
static int g_c64_blocks = 2; // persistent blocks per CU of wino_gemm_c64_kernel
static long long* g_prof = nullptr;

template <class Shape, int ABLATE, int V0 = 0, int NT = 0>
double run(const char* name, const Case& cs, float* U, float* V, float* M, int reps)
{
    WinoGemmPolicy::Params g;
    g.batches = g_batches;
    g.U = U;
    g.V = V;
    g.M = M;
    g.C = cs.C;
    g.K = cs.K;
    g.Cp = round_up(cs.C, Shape::BK);
    g.Kp = round_up(cs.K, Shape::BM);
    g.Pp = round_up(cs.P, Shape::BN);
    g.k_tiles = g.Cp / Shape::BK;
    g.n_tiles = g.Pp / Shape::BN;
    g.m_tiles = g.Kp / Shape::BM;
    if (V0 == 7)
    {
        g.n_tiles = (cs.P + 95) / 96;
        g.Pp = std::max(g.Pp, g.n_tiles * 96);
    }
    // round 4: column-blocked V / M (wino_layout.h); GEMM_BP = 0 / unset: whole rows
    const int bp = g_bp && g_bp % Shape::BN == 0 && V0 != 7 ? g_bp : 0;
    if (bp) g.Pp = round_up(g.Pp, bp);
    g.Lv = wino_layout(g.C, g.Pp, bp);
    g.Lm = wino_layout(g.K, g.Pp, bp);
    const int tiles = g.batches * g.m_tiles * g.n_tiles;
    dim3 grid(tiles);
    if (V0 == 3 && g_split_rounds)
    {
        wino_gemm_row_split(tiles, g_cus, g_split_rounds, g.tail_first, g.tail_parts);
        grid = dim3(wino_gemm_row_split_grid(tiles, g.tail_first, g.tail_parts));
    }
    auto launch = [&]() {
        if constexpr (V0 == 3)
        {
            if (g.tail_parts > 1)
                hipLaunchKernelGGL((wino_gemm_glds_kernel<2, 16, 6, NT, true>), grid, dim3(256), 0, 0, g);
            else
                hipLaunchKernelGGL((wino_gemm_glds_kernel<2, 16, 6, NT>), grid, dim3(256), 0, 0, g);
        }
        else if constexpr (V0 == 9) // the product's main loop itself (not the probe copy), with the NT hints of WinoGemmPolicyT
            hipLaunchKernelGGL((gemm_mfma_kernel<Shape, WinoGemmPolicyT<NT>>), grid, dim3(Shape::THREADS), 0, 0, g);
        else if constexpr (V0 == 4)
            hipLaunchKernelGGL((wino_gemm_glds_kernel<2, 16, 5>), grid, dim3(256), 0, 0, g);
        else if constexpr (V0 == 5)
            hipLaunchKernelGGL((wino_gemm_glds_kernel<2, 16, 6>), grid, dim3(256), 0, 0, g);
        else if constexpr (V0 == 6)
            hipLaunchKernelGGL((wino_gemm_glds_kernel<2, 16, 3>), grid, dim3(256), 0, 0, g);
        else if constexpr (V0 == 7)
            hipLaunchKernelGGL(wino_gemm_glds96_kernel, grid, dim3(256), 0, 0, g);
        else if constexpr (V0 == 10)
            hipLaunchKernelGGL((wino_gemm_glds128_kernel<4, NT>), grid, dim3(256), 0, 0, g);
        else if constexpr (V0 == 11)
            hipLaunchKernelGGL((wino_gemm_glds128_kernel<5, NT>), grid, dim3(256), 0, 0, g);
        else if constexpr (V0 == 8)
            hipLaunchKernelGGL(wino_gemm_c64_kernel, dim3(std::min(tiles, g_cus * g_c64_blocks)), dim3(256), 0, 0, g); // persistent
        else
            hipLaunchKernelGGL((gemm_mfma_probe_kernel<Shape, WinoGemmPolicy, ABLATE>), grid, dim3(Shape::THREADS), 0, 0, g);
    };
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    const double tf = 2.0 * g_batches * cs.K * cs.C * (double)cs.P / ms / 1e9;
    // correctness against the first variant run for this case: planes xi = 0 and xi = batches-1, valid rows / columns only
    {
        static std::vector<float> ref[2];
        static Case ref_case{0, 0, 0};
        std::vector<float> got[2];
        const size_t plane = (size_t)cs.K * g.Pp;
        const bool fresh = ref_case.C != cs.C || ref_case.K != cs.K || ref_case.P != cs.P;
        for (int w = 0; w < 2; ++w)
        {
            got[w].resize(plane);
            CK(hipMemcpy(got[w].data(), M + (size_t)(w ? g_batches - 1 : 0) * plane, plane * 4, hipMemcpyDeviceToHost));
        }
        if (fresh && ABLATE == 0 && (V0 == 0 || (V0 == 3 && !g_split_rounds && getenv("GEMM_SPLIT"))) && !bp)
        {
            ref[0] = got[0];
            ref[1] = got[1];
            ref_case = cs;
            g_ref_pp = g.Pp;
        }
        else if (!fresh && ABLATE == 0 && !bp)
        {
            double worst = 0, scale = 0;
            for (int w = 0; w < 2; ++w)
                for (int m = 0; m < cs.K; ++m)
                    for (int p = 0; p < cs.P; ++p)
                    {
                        const double a = got[w][(size_t)m * g.Pp + p], b = ref[w][(size_t)m * g_ref_pp + p];
                        worst = std::max(worst, std::abs(a - b));
                        scale = std::max(scale, std::abs(b));
                    }
            if (worst > 1e-5 * scale) printf("    !! %s differs from the product kernel: max |diff| %.3e (scale %.3e)\n", name, worst, scale);
        }
    }
    printf("  %-34s C%4d K%4d P%6d grid %6d  %8.4f ms  %7.2f TF (%.1f%% of 157.3)\n", name, cs.C, cs.K, cs.P, grid.x, ms, tf, tf / 157.3 * 100);
    return tf;
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    // the Winograd tile-GEMM shapes of VGG-16 at batch 32 (C, K, P)
    // (GEMM_RESNET: ResNet-50's 3x3 layers at batch 64 instead)
    const Case vgg[] = {{64, 64, 46208}, {64, 128, 11552}, {128, 128, 11552}, {128, 256, 3200}, {256, 256, 3200}, {256, 512, 800}, {512, 512, 800}, {512, 512, 288}};
    const Case resnet[] = {{64, 64, 6400}, {256, 256, 576}, {512, 512, 256}, {128, 128, 1600}, {256, 256, 576}, {512, 512, 256}, {256, 256, 576}, {512, 512, 256}};
    // GEMM_TAIL: what a partial last round costs -- the LDS-DMA kernel on column counts that make whole rounds of 6 blocks per CU next to VGG-16's
    const Case tail[] = {{512, 512, 768}, {512, 512, 800}, {512, 512, 1152}, {256, 256, 3072}, {256, 256, 3200}, {256, 256, 4608}, {128, 128, 12288}, {128, 128, 11552}};
    // GEMM_SPLIT: the row split of the last tiles -- ResNet-50 b64's res5 (36 frequency points, set GEMM_BATCHES=36), res3, res4 on 64-column tiles, VGG-16's conv2_2
    const Case split[] = {{512, 512, 256}, {128, 128, 1600}, {256, 256, 576}, {128, 128, 11552}, {512, 512, 256}, {128, 128, 1600}, {256, 256, 576}, {128, 128, 11552}};
    if (getenv("GEMM_BATCHES")) g_batches = atoi(getenv("GEMM_BATCHES"));
    const Case* cases_p = getenv("GEMM_SPLIT") ? split : getenv("GEMM_TAIL") ? tail : getenv("GEMM_RESNET") ? resnet : vgg;
    Case cases[8];
    for (int i = 0; i < 8; ++i) cases[i] = cases_p[i];
    size_t maxU = 0, maxV = 0, maxM = 0;
    for (auto& c : cases)
    {
        maxU = std::max(maxU, (size_t)64 * round_up(c.C, 32) * round_up(c.K, 256));
        maxV = std::max(maxV, (size_t)64 * c.C * round_up(c.P, 1536));
        maxM = std::max(maxM, (size_t)64 * c.K * round_up(c.P, 1536));
    }
    float *U, *V, *M;
    CK(hipMalloc(&U, maxU * 4));
    CK(hipMalloc(&V, maxV * 4));
    CK(hipMalloc(&M, maxM * 4));
    std::vector<float> h(1 << 20);
    srand(1);
    for (auto& x : h) x = (rand() / (float)RAND_MAX) * 2 - 1;
    for (size_t off = 0; off < maxU; off += h.size()) CK(hipMemcpy(U + off, h.data(), std::min(h.size(), maxU - off) * 4, hipMemcpyHostToDevice));
    for (size_t off = 0; off < maxV; off += h.size()) CK(hipMemcpy(V + off, h.data(), std::min(h.size(), maxV - off) * 4, hipMemcpyHostToDevice));
    hipDeviceGetAttribute(&g_cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs %d\n", g_cus);
    if (argc > 2)
    {
        // occupancy scan: ONE 128x64 tile per block, `bpc` blocks per CU resident at once, deep K: what does the main loop reach
        // when only 1, 2, 3, 4 blocks share a CU?  (ablations: 1 = no global fetch, 2 = no store)
        g_batches = 1;
        for (int kt : {64, 256})
            for (int bpc : {1, 2, 3, 4, 8})
            {
                Case c{kt * 16, 128, 64 * g_cus * bpc};
                if ((size_t)c.C * round_up(c.P, 256) > maxV || (size_t)c.K * round_up(c.P, 256) > maxM) continue;
                printf("kt %d, %d block(s) per CU\n", kt, bpc);
                run<GemmShape<128, 64, 16, 2, 2, 4>, 0>("product", c, U, V, M, reps);
                run<GemmShape<128, 64, 16, 2, 2, 4>, 1>("no global fetch", c, U, V, M, reps);
            }
        return 0;
    }
    for (auto& c : cases)
    {
        if (getenv("GEMM_ONE") && !(c.C == 512 && c.K == 512 && c.P == 800)) continue;
        printf("case C=%d K=%d P=%d\n", c.C, c.K, c.P);
        if (getenv("GEMM_SPLIT"))
        {
            if (&c - cases >= 4) continue;
            for (int round = 0; round < 4; ++round)
            {
                g_split_rounds = 0;
                run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3, 0>("128x64 glds whole tiles", c, U, V, M, reps);
                g_split_rounds = 64;
                run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3, 0>("128x64 glds row split", c, U, V, M, reps);
                g_split_rounds = 0;
                if (c.P <= 576) run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 7>("128x96 glds", c, U, V, M, reps);
            }
            continue;
        }
        if (getenv("GEMM_TAIL"))
        {
            for (int round = 0; round < 3; ++round) run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3, 0>("128x64 glds (product)", c, U, V, M, reps);
            continue;
        }
        if (getenv("GEMM_R5"))
        {
            // round 5: non-temporal hints on the read-once / write-once streams, wider tiles for conv2_1, and what a partial last round costs
            for (int round = 0; round < 3; ++round)
            {
                if (c.K <= 64)
                {
                    for (int bp : {1024})
                    {
                        g_bp = bp;
                        run<GemmShape<64, 128, 16, 1, 4, 4>, 0, 9, 0>("64x128 reg plain BP1024", c, U, V, M, reps);
                        run<GemmShape<64, 128, 16, 1, 4, 4>, 0, 9, 1>("64x128 reg nt-load BP1024", c, U, V, M, reps);
                        run<GemmShape<64, 128, 16, 1, 4, 4>, 0, 9, 2>("64x128 reg nt-store BP1024", c, U, V, M, reps);
                        run<GemmShape<64, 128, 16, 1, 4, 4>, 0, 9, 3>("64x128 reg nt-both BP1024", c, U, V, M, reps);
                    }
                    g_bp = 0;
                }
                else if (c.C < 128)
                {
                    run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 9, 0>("128x64 reg plain (product)", c, U, V, M, reps);
                    run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 9, 1>("128x64 reg nt-load", c, U, V, M, reps);
                    run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 9, 2>("128x64 reg nt-store", c, U, V, M, reps);
                    run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 9, 3>("128x64 reg nt-both", c, U, V, M, reps);
                    run<GemmShape<128, 64, 16, 2, 2, 6>, 0, 9, 0>("128x64 reg occ 6", c, U, V, M, reps);
                    run<GemmShape<128, 128, 16, 2, 2, 3>, 0, 9, 0>("128x128 2x2 occ 3", c, U, V, M, reps);
                    run<GemmShape<128, 128, 16, 2, 2, 3>, 0, 9, 3>("128x128 2x2 occ 3 nt-both", c, U, V, M, reps);
                    run<GemmShape<128, 128, 16, 4, 1, 4>, 0, 9, 0>("128x128 4x1 occ 4", c, U, V, M, reps);
                    run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3, 0>("128x64 glds", c, U, V, M, reps);
                    run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3, 3>("128x64 glds nt-both", c, U, V, M, reps);
                    g_bp = 1024;
                    run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 9, 0>("128x64 reg plain BP1024", c, U, V, M, reps);
                    run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 9, 3>("128x64 reg nt-both BP1024", c, U, V, M, reps);
                    g_bp = 0;
                }
                else
                {
                    run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3, 0>("128x64 glds plain (product)", c, U, V, M, reps);
                    run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3, 1>("128x64 glds nt V loads", c, U, V, M, reps);
                    run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3, 2>("128x64 glds nt M stores", c, U, V, M, reps);
                    run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3, 3>("128x64 glds nt both", c, U, V, M, reps);
                }
            }
            continue;
        }
        if (getenv("GEMM_REGVSGLDS"))
        {
            // round 6: conv2_2-class launches (C = 128: 8 k-tiles) on the register-staged main loop against the LDS-DMA kernel the product picks from 8 k-tiles on
            if (c.C != 128) continue;
            for (int round = 0; round < 3; ++round)
            {
                run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3, 2>("128x64 glds nt M (product)", c, U, V, M, reps);
                run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 9, 2>("128x64 reg nt M", c, U, V, M, reps);
                run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 9, 3>("128x64 reg nt M + nt V", c, U, V, M, reps);
                run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3, 0>("128x64 glds plain", c, U, V, M, reps);
            }
            continue;
        }
        if (getenv("GEMM_128"))
        {
            // round 6: the LDS-DMA kernel with a 128 x 128 tile (tools/experiments/wino_gemm_glds128.h) against the product's launch, MFMA-bound shapes only
            if (c.C < 128) continue;
            run<GemmShape<128, 64, 16, 2, 2, 4>, 0>("128x64 reg (reference values)", c, U, V, M, 2);
            for (int round = 0; round < 3; ++round)
            {
                if (c.P == 288) run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 7>("128x96 glds (product)", c, U, V, M, reps);
                else run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3, 2>("128x64 glds nt M (product)", c, U, V, M, reps);
                run<GemmShape<128, 128, 16, 2, 2, 4>, 0, 10, 2>("128x128 glds occ 4 nt M", c, U, V, M, reps);
                run<GemmShape<128, 128, 16, 2, 2, 4>, 0, 11, 2>("128x128 glds occ 5 nt M", c, U, V, M, reps);
            }
            continue;
        }
        if (getenv("GEMM_96"))
        {
            if (c.C < 128) continue;
            for (int round = 0; round < 5; ++round)
            {
                run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3>("128x64 glds (product)", c, U, V, M, reps);
                run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 7>("128x96 glds", c, U, V, M, reps);
            }
            continue;
        }
        if (getenv("GEMM_C64"))
        {
            // round 4: the persistent streaming kernel for C, K <= 64 against the product's one-tile-per-block kernel, whole rows and column blocks
            if (c.C > 64 || c.K > 64) continue;
            for (int round = 0; round < 3; ++round)
                for (int bp : {0, 1024})
                {
                    g_bp = bp;
                    char nm[64];
                    snprintf(nm, sizeof nm, "64x128 reg-staged (product) BP %d", bp);
                    run<GemmShape<64, 128, 16, 1, 4, 4>, 0>(nm, c, U, V, M, reps);
                    for (int bpc : {1, 2})
                    {
                        g_c64_blocks = bpc;
                        snprintf(nm, sizeof nm, "c64 persistent %d/CU BP %d", bpc, bp);
                        run<GemmShape<64, 128, 16, 1, 4, 4>, 0, 8>(nm, c, U, V, M, reps);
                    }
                }
            g_bp = 0;
            continue;
        }
        if (getenv("GEMM_LAYOUT"))
        {
            // whole rows against column blocks of 64 ... 512, the product's kernel for the shape, interleaved rounds
            for (int round = 0; round < 3; ++round)
                for (int bp : {0, 64, 128, 256, 512})
                {
                    g_bp = bp;
                    char nm[64];
                    snprintf(nm, sizeof nm, "BP %d", bp);
                    if (c.K <= 64)
                    {
                        if (bp == 64) continue;
                        run<GemmShape<64, 128, 16, 1, 4, 4>, 0>((std::string("64x128 reg-staged ") + nm).c_str(), c, U, V, M, reps);
                    }
                    else if (c.C < 128)
                        run<GemmShape<128, 64, 16, 2, 2, 4>, 0>((std::string("128x64 reg-staged ") + nm).c_str(), c, U, V, M, reps);
                    else
                        run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3>((std::string("128x64 glds ") + nm).c_str(), c, U, V, M, reps);
                }
            g_bp = 0;
            continue;
        }
        for (int round = 0; round < 3; ++round)
        {
            run<GemmShape<128, 64, 16, 2, 2, 4>, 0>("128x64x16 2x2 (product)", c, U, V, M, reps);
            run<GemmShape<128, 64, 16, 2, 2, 4>, 0, 3>("128x64x16 glds 2 buffers", c, U, V, M, reps);
            run<GemmShape<64, 128, 16, 1, 4, 4>, 0>("64x128 1x4 (product small-M)", c, U, V, M, reps);
            run<GemmShape<64, 64, 16, 2, 2, 8>, 0>("64x64 2x2 occ 8", c, U, V, M, reps);
            run<GemmShape<64, 64, 16, 1, 2, 8>, 0>("64x64 1x2 (2 waves) occ 8", c, U, V, M, reps);
            run<GemmShape<64, 128, 16, 1, 4, 6>, 0>("64x128 1x4 occ 6", c, U, V, M, reps);
            run<GemmShape<64, 256, 16, 1, 4, 3>, 0>("64x256 1x4 occ 3", c, U, V, M, reps);
        }
    }
    return 0;
}
