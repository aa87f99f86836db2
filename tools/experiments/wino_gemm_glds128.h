// tools/experiments/wino_gemm_glds128.h -- round 6 experiment: the LDS-DMA Winograd tile GEMM (wino_gemm_glds.h) with a 128 x 128 tile.
// 4 waves as 2 x 2, a wave owns 64 rows x 64 columns = four 32x32 accumulators: 4 fragment reads per 4 MFMAs (1.0 per MFMA against 1.5 with the
// 128 x 64 tile), 32 MFMAs per wave between two barriers instead of 16, half as many blocks (prologue / epilogue amortised over twice the matrix
// work).  Two LDS buffers of [16][128] + [16][128] floats = 32 KB per block, 90 VGPRs: OCC blocks per CU (4 or 5).  V / M are padded to 128 columns
// already (kWinoColTile), so the only extra work is the padding a 128-column tile adds over a 64-column one (P = 800: 896 against 832 columns).
// Measured (tools/gemm_bench.hip GEMM_128=1, 20 launches x 3 rounds, same call; bit-identical M): conv2_2 302 -> 290 us, conv3_1 124 -> 122,
// conv3_2 / 3_3 219.5 -> 217.5, conv4_1 115 -> 121, conv4_2 / 4_3 219 -> 232, conv5 (96-column tile) 81 -> 103.  Wired into the library for the
// launches it wins (>= 8 tiles per CU, columns a multiple of 128) it changed VGG-16 b32 by nothing measurable in the net -- 10 108 / 9 943 / 9 991 img/s
// with it against 9 978 / 9 958 / 10 116 without, tile GEMM 1.985 / 2.011 / 1.987 ms against 2.007 / 2.001 / 1.977 (three interleaved rounds) -- and
// was taken out again (EXPERIMENTS.md R6.7).  Not part of the product.
#pragma once

#include "wino_gemm_glds.h"

namespace fhip
{

template <int OCC, int NT = 0>
__global__ __launch_bounds__(256, OCC) void wino_gemm_glds128_kernel(const WinoGemmPolicy::Params prm)
{
    constexpr int BM = 128, BN = 128, BK = 16, EPI_LD = 36, NBUF = 2;
    constexpr int BUF_FLOATS = BK * (BM + BN);
    constexpr int LDSF = NBUF * BUF_FLOATS > 4 * 32 * EPI_LD ? NBUF * BUF_FLOATS : 4 * 32 * EPI_LD;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];

    const int nwg = prm.batches * prm.m_tiles * prm.n_tiles;
    int vid = xcd_remap(blockIdx.x, nwg);
    const int mt = vid % prm.m_tiles;
    vid /= prm.m_tiles;
    const int nt = vid % prm.n_tiles;
    const int xi = vid / prm.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int k_tiles = prm.k_tiles;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    // A and B alike: wave w, piece i covers k rows 4w + 2i, 4w + 2i + 1 (32 lanes x 16 B per row of 128 floats)
    const float* srcA = prm.U + (size_t)xi * prm.Cp * prm.Kp + (size_t)(wave * 4 + half) * prm.Kp + m0 + l31 * 4;
    const float* srcB = prm.V + (size_t)xi * prm.Lv.xis + prm.Lv.col(n0) + l31 * 4; // (whole rows or column blocks that are multiples of 128)
    const int brow = wave * 4 + half;
    const size_t a_step = (size_t)BK * prm.Kp;

    auto issue = [&](int kt, int buf) {
        float* base = lds + buf * BUF_FLOATS;
        const float* a = srcA + (size_t)kt * a_step;
        __builtin_amdgcn_global_load_lds(a, (lds_void*)(base + (wave * 4) * BM), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(a + (size_t)2 * prm.Kp, (lds_void*)(base + (wave * 4 + 2) * BM), 16, 0, 0);
        float* bb = base + BK * BM;
#pragma unroll
        for (int i = 0; i < 2; ++i)
        {
            const int r = min(kt * BK + brow + 2 * i, prm.C - 1);
            __builtin_amdgcn_global_load_lds(srcB + (size_t)r * prm.Lv.bp, (lds_void*)(bb + (wave * 4 + 2 * i) * BN), 16, 0, (NT & 1) ? 2 : 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int a_off = half * BM + wm * 64 + l31;
    const int b_off = BK * BM + half * BN + wn * 64 + l31;
    int cur = 0;
    for (int kt = 0; kt < k_tiles; ++kt)
    {
        if (kt + 1 < k_tiles) issue(kt + 1, cur ^ 1);
        const float* as = lds + cur * BUF_FLOATS + a_off;
        const float* bs = lds + cur * BUF_FLOATS + b_off;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp)
        {
            const float fa0 = as[(2 * kp) * BM], fa1 = as[(2 * kp) * BM + 32], fb0 = bs[(2 * kp) * BN], fb1 = bs[(2 * kp) * BN + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1, acc[1][1], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }

    float* const scr = lds + wave * (32 * EPI_LD);
    const int e_row = lane >> 3, e_c4 = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j)
    {
        float* mbase = prm.M + (size_t)xi * prm.Lm.xis + prm.Lm.col(n0 + wn * 64 + j * 32) + e_c4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * half) * EPI_LD + l31] = acc[i][j][r];
            const int mrow = m0 + wm * 64 + i * 32 + e_row;
#pragma unroll
            for (int q = 0; q < 4; ++q)
            {
                const float4 v = *reinterpret_cast<const float4*>(&scr[(q * 8 + e_row) * EPI_LD + e_c4]);
                const int m = mrow + q * 8;
                if (m < prm.K)
                {
                    if constexpr ((NT & 2) != 0)
                        stg4_nt(mbase + (size_t)m * prm.Lm.bp, v);
                    else
                        *reinterpret_cast<float4*>(mbase + (size_t)m * prm.Lm.bp) = v;
                }
            }
        }
    }
}

} // namespace fhip
