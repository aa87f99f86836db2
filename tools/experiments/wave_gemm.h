// wave_gemm.h -- fp32-MFMA GEMM with WAVE-PRIVATE tiles: one wavefront = one workgroup = one 64 x 64 output tile, its own
// LDS ring of k-tile stages filled by LDS-DMA, no s_barrier anywhere.  Serves the Winograd tile GEMM (reference TensorGEMM,
// src/booster/avx/winograd_kernels_F63.cpp:518-692) and the 1x1 implicit-GEMM convolution (IM2COL_Forward +
// packed_sgemm_activation, avx/booster.cpp:83-102, avx/sgemm.cpp:377-433) through the policies of flat_gemm.h.
//
// Why (round-2 ablations, tools/flat_bench.hip): with every HBM read and every store removed, the four-wave 128 x 64 block
// kernels (gemm_core.h, wino_gemm_glds.h, flat_gemm.h) still run ResNet-50's 1x1 layers at 45-53 % of the fp32-MFMA peak:
// the loss is inside the CU.  Their waves read operands from LDS right in front of the MFMAs that use them (hipcc: ds_read,
// s_waitcnt lgkmcnt(0), 2 x v_mfma -- ~100 exposed cycles per 128 of matrix work unless 4-6 other waves cover them), every
// k-tile ends in a barrier that couples four SIMDs, and a layer of 6.6 GFLOP has only 3-12 such blocks per CU to hide all
// that with.  Here:
//   * a wave owns a whole 64 x 64 tile (4 accumulators): 4 operand reads feed 4 MFMAs (256 matrix cycles), and the reads of
//     k-step kp+1 go to a SECOND register set before the MFMAs of k-step kp are issued -- the LDS latency sits under matrix
//     work of the same wave, so one or two waves per SIMD are enough;
//   * operands arrive by global_load_lds into the wave's own ring; the only synchronisation is the wave's own counted
//     s_waitcnt vmcnt (in-order retirement, stores included -- same bookkeeping as flat_gemm.h); the wait for stage s+1 and
//     the read of its first fragments happen before the LAST k-step of stage s, so stage boundaries do not expose latency;
//   * the scheduler balances single-wave workgroups at the finest grain (a 128 x 64 block is 4 waves that start and end together);
//   * price: A and B panels are fetched per wave (64 + 64 rows per 4 MFMAs = 256 B per MFMA against 192 B for the shared
//     128 x 64 tile) -- L2 / L1 traffic, not HBM; LDS capacity bounds residency at 8-10 waves per CU.
#pragma once

#include "flat_gemm.h"

namespace fhip
{

template <int BK_, int D_, int WPS_>
struct WaveShape
{
    static constexpr int BM = 64, BN = 64, BK = BK_, D = D_, WPS = WPS_;
    static constexpr int STAGE = BK * (BM + BN); // floats: A [BK][64] then B [BK][64]
    static constexpr int EPI_LD = 36;
    static constexpr int EPI = 16 * EPI_LD;
    static constexpr int LDS_FLOATS = D * STAGE + EPI + BM;
    static_assert(BK % 4 == 0 && BK >= 4 && D >= 2 && D <= 4, "stage shape");
};

template <class Shape, class Policy, int ABLATE = 0>
__global__ __launch_bounds__(64, Shape::WPS) void wave_gemm_kernel(const typename Policy::Params prm)
{
    constexpr int BM = 64, BN = 64, BK = Shape::BK, D = Shape::D, STAGE = Shape::STAGE, EPI_LD = Shape::EPI_LD;
    constexpr int VEC = Policy::VEC;
    constexpr int GA = BK / 4;                    // A requests per stage: 4 rows of 64 floats each
    constexpr int GB = VEC == 4 ? BK / 4 : BK;    // B requests per stage
    constexpr int G = GA + GB;
    constexpr int KS = BK / 2;                    // MFMA k-steps per stage
    __shared__ __attribute__((aligned(16))) float lds[Shape::LDS_FLOATS];
    float* const scr = lds + D * STAGE;
    float* const bias_s = scr + Shape::EPI;

    const int n_groups = (prm.n_tiles + prm.tpb - 1) / prm.tpb;
    const int nwg = prm.batches * prm.m_tiles * n_groups;
    int vid = xcd_remap(blockIdx.x, nwg);
    const int mt = vid % prm.m_tiles;
    vid /= prm.m_tiles;
    const int grp = vid % n_groups;
    const int batch = vid / n_groups;
    const int m0 = mt * BM;
    const int t0 = grp * prm.tpb;
    const int ntl = min(prm.tpb, prm.n_tiles - t0);
    const int k_tiles = prm.k_tiles;
    const int total = ntl * k_tiles;

    const long long probe_c0 = (ABLATE & 8) ? clock64() : 0, probe_w0 = (ABLATE & 8) ? wall_clock64() : 0;
    const int lane = threadIdx.x;
    const int l31 = lane & 31, half = lane >> 5;

    {
        const float* bp = Policy::bias(prm);
        bias_s[lane] = (bp && m0 + lane < Policy::rows(prm)) ? bp[m0 + lane] : 0.f;
    }

    // ---- request descriptors: piece g of A covers rows 4g .. 4g+3 (16 lanes x 16 B per row)
    const int r4 = lane >> 4, c4 = (lane & 15) * 4;
    const float* a_src = Policy::a_ptr(prm, batch, m0 + c4, r4);
    const size_t lda = Policy::lda(prm), ldb = Policy::ldb(prm);
    const int krows = Policy::krows(prm);
    const float* b_src = nullptr;
    auto set_issue_tile = [&](int t) {
        const int n = (t0 + ((ABLATE & 1) ? 0 : t)) * BN + (VEC == 4 ? c4 : lane);
        b_src = Policy::b_ptr(prm, batch, n);
    };
    int it_i = 0, kt_i = 0, buf_i = 0;
    set_issue_tile(0);
    auto issue_next = [&]() {
        float* base = lds + buf_i * STAGE;
        const float* a = a_src + (size_t)(kt_i * BK) * lda;
#pragma unroll
        for (int g = 0; g < GA; ++g) flat_request<4>(a + (size_t)(4 * g) * lda, base + g * 256);
#pragma unroll
        for (int g = 0; g < GB; ++g)
        {
            const int r = min(kt_i * BK + (VEC == 4 ? 4 * g + r4 : g), krows - 1);
            flat_request<VEC>(b_src + (size_t)r * ldb, base + BK * BM + g * (64 * VEC));
        }
        buf_i = buf_i + 1 == D ? 0 : buf_i + 1;
        if (++kt_i == k_tiles)
        {
            kt_i = 0;
            if (++it_i < ntl) set_issue_tile(it_i);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int f = 0; f < D - 1; ++f)
        if (f < total) issue_next();

    const int a_off = half * BM + l31;
    const int b_off = BK * BM + half * BN + l31;
    const int e_row = lane >> 3, e_c4 = (lane & 7) * 4;
    const int out_rows = Policy::rows(prm), out_cols = Policy::cols(prm);

    int hist[D - 1]; // hist[i]: store instructions issued in iteration s-1-i
#pragma unroll
    for (int i = 0; i < D - 1; ++i) hist[i] = 0;

    // stage 0 landed?  (behind it: the D-2 other prologue stages)
    flat_wait(min(total - 1, D - 2) * G);
    float fa[2][2], fb[2][2]; // [register set][tile]
    {
        const float* as = lds + a_off;
        const float* bs = lds + b_off;
        fa[0][0] = as[0];
        fa[0][1] = as[32];
        fb[0][0] = bs[0];
        fb[0][1] = bs[32];
    }

    int cur = 0, kt = 0, it = 0;
    for (int s = 0; s < total; ++s)
    {
        // the stage refilled now was read during iteration s-1; all of those reads fed MFMAs that are already issued
        if (s + D - 1 < total && !(ABLATE & 16)) issue_next();
        const float* as = lds + cur * STAGE + a_off;
        const float* bs = lds + cur * STAGE + b_off;
        const int nxt = cur + 1 == D ? 0 : cur + 1;
#pragma unroll
        for (int kp = 0; kp < KS; ++kp)
        {
            const int x = kp & 1, y = x ^ 1;
            if (kp + 1 < KS)
            {
                fa[y][0] = as[(2 * kp + 2) * BM];
                fa[y][1] = as[(2 * kp + 2) * BM + 32];
                fb[y][0] = bs[(2 * kp + 2) * BN];
                fb[y][1] = bs[(2 * kp + 2) * BN + 32];
            }
            else if (s + 1 < total)
            {
                // before the LAST k-step: stage s+1 must have landed (behind its loads: the stages s+2 .. s+D-1 and the stores of
                // the epilogues since), then its first fragments are requested under this stage's last MFMAs
                int behind = min(total - 2 - s, D - 2) * G;
#pragma unroll
                for (int i = 0; i < D - 2; ++i) behind += hist[i];
                flat_wait(behind);
                const float* an = lds + nxt * STAGE + a_off;
                const float* bn = lds + nxt * STAGE + b_off;
                fa[y][0] = an[0];
                fa[y][1] = an[32];
                fb[y][0] = bn[0];
                fb[y][1] = bn[32];
            }
            // pin the order: the operand reads above are issued BEFORE this k-step's MFMAs (hipcc would sink them below and then wait
            // for them at once, exposing the LDS latency once per k-step)
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABLATE & 4))
            {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x][0], fb[x][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x][0], fb[x][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x][1], fb[x][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x][1], fb[x][1], acc[1][1], 0, 0, 0);
            }
        }
        // KS is even: after the loop the fragments of the next stage sit in set 0 again
        static_assert(KS % 2 == 0, "register sets alternate per k-step");

        int stores = 0;
        const bool end = kt == k_tiles - 1;
        if (end)
        {
            const int n0 = (t0 + it) * BN;
#pragma unroll
            for (int j = 0; j < 2; ++j)
            {
                const int nj = n0 + j * 32;
                const typename Policy::Out st(prm, batch, nj + e_c4);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                    {
#pragma unroll
                        for (int r = 0; r < 8; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * half) * EPI_LD + l31] = acc[i][j][8 * h + r];
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                        {
                            const int mq = i * 32 + h * 16 + q * 8;
                            if (nj < out_cols && m0 + mq < out_rows) // wave-uniform; lane (row 0, column 0) is active
                            {
                                const float* sp = &scr[(q * 8 + e_row) * EPI_LD + e_c4]; // float reads: see flat_gemm.h
                                const float4 v = make_float4(sp[0], sp[1], sp[2], sp[3]);
                                if (ABLATE & 2)
                                    asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                                else
                                {
                                    st.put4(prm, m0 + mq + e_row, v, bias_s[mq + e_row]);
                                    ++stores;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
            }
            ++it;
            kt = 0;
        }
        else
            ++kt;
#pragma unroll
        for (int i = D - 2; i > 0; --i) hist[i] = hist[i - 1];
        hist[0] = stores;
        cur = nxt;
    }
    if ((ABLATE & 8) && threadIdx.x == 0 && (blockIdx.x & 63) == 0)
    {
        atomicAdd(&g_flat_clock_probe[0], (unsigned long long)(clock64() - probe_c0));
        atomicAdd(&g_flat_clock_probe[1], (unsigned long long)(wall_clock64() - probe_w0));
    }
}

} // namespace fhip
