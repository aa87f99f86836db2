// gemm_il.h -- EXPERIMENT (tools only, round 4, VERDICT r03 #1b): the LDS-tiled fp32-MFMA GEMM of gemm_core.h with a TRANSPOSE-FREE epilogue.
// Measured on all 16 ResNet-50 b64 1x1 shapes (tools/r50_probe.hip PROBE_SHAPES=1, interleaved rounds, median of 5, results checked against the
// product): slower than the product's 128 x 64 / 64 x 128 tiles with the LDS transpose on EVERY shape -- e.g. 256 -> 1024 @14x14: 81.8 - 86.0 us
// (128x128 / 64x128 / 64x256 / 128x256 tiles) vs 71.4; 128 -> 512 @28x28: 85.8 - 98.0 vs 73.9; 64 -> 256 @56x56: 99.9 - 128.7 vs 86.7; 512 -> 2048
// @7x7: 96.2 - 115.4 vs 85.0.  A wave that owns 32 x 128 outputs reads one A fragment per four MFMAs but halves the number of waves a tile
// is spread over (or doubles the tile): the epilogue was never what limited these launches.
#pragma once
#include "gemm_core.h"

namespace fhip
{

// ---- the same main loop with a TRANSPOSE-FREE epilogue (round 4 experiment, VERDICT r03 #1b; instantiated by tools/r50_probe.hip only) ----
// A wave owns 32 rows x 128 columns as the four INTERLEAVED column sets {4 l + t} (stream_gemm.h's arrangement): one ds_read_b128 of the
// k-major B tile gives lane l columns 4 l .. 4 l + 3 of a k row, i.e. the B operands of four MFMAs, and accumulator register r of the four
// accumulators then holds four CONSECUTIVE columns of one output row -- the tile leaves as 16 dwordx4 stores per wave with no LDS round trip,
// no transpose scratch and no second wait.  Shape: WTM = 32, WTN = 128 (block tiles 128 x 128 with 4 x 1 waves, 64 x 128 with 2 x 1, ...).
// Everything before the epilogue is gemm_mfma_kernel's code.  Measured: DESIGN.md 3.9.
template <class Shape, class Policy>
__global__ __launch_bounds__(Shape::THREADS, Shape::BLOCKS_PER_CU* Shape::THREADS / 256) void gemm_mfma_il_kernel(const typename Policy::Params prm)
{
    static_assert(Shape::WTM == 32 && Shape::WTN == 128, "interleaved epilogue: 32 x 128 wave tiles");
    constexpr int BM = Shape::BM, BN = Shape::BN, BK = Shape::BK;
    __shared__ __attribute__((aligned(16))) float lds[Shape::OPERAND_FLOATS + Policy::EXTRA_LDS_FLOATS];
    float* const extra = lds + Shape::OPERAND_FLOATS;
    float* const As0 = lds;
    float* const Bs0 = lds + 2 * BK * BM;

    __builtin_amdgcn_s_setprio(3);
    const int nwg = prm.batches * prm.m_tiles * prm.n_tiles;
    int vid = xcd_remap(blockIdx.x, nwg);
    const int mt = vid % prm.m_tiles;
    vid /= prm.m_tiles;
    const int nt = vid % prm.n_tiles;
    const int batch = vid / prm.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int k_tiles = Policy::k_count(prm, batch);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Shape::WAVES_N, wn = wave % Shape::WAVES_N;
    const int l31 = lane & 31, half = lane >> 5;
    const int a_c4 = tid % Shape::A_F4_PER_ROW, a_r = tid / Shape::A_F4_PER_ROW;
    const int b_c4 = tid % Shape::B_F4_PER_ROW, b_r = tid / Shape::B_F4_PER_ROW;
    const typename Policy::ALoad aload(prm, batch, m0 + a_c4 * 4);
    const typename Policy::BLoad bload(prm, batch, n0 + b_c4 * 4);

    typedef typename Policy::BLoad::Raw BRaw;
    float4 pa[Shape::A_PASSES];
    BRaw pb[Shape::B_PASSES];
    unsigned pok[Shape::B_PASSES];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i) pa[i] = aload.load(prm, kt * BK + a_r + i * Shape::A_ROWS_PER_PASS);
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i) pb[i] = bload.load(prm, kt * BK + b_r + i * Shape::B_ROWS_PER_PASS, pok[i]);
    };
    auto stash = [&](int buf, int kt) {
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i)
            *reinterpret_cast<float4*>(&As0[buf * (BK * BM) + (a_r + i * Shape::A_ROWS_PER_PASS) * BM + a_c4 * 4]) = pa[i];
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i)
        {
            float4 v = bload.finish(prm, pb[i], kt * BK + b_r + i * Shape::B_ROWS_PER_PASS, extra);
            v.x = (pok[i] & 1u) ? v.x : 0.f;
            v.y = (pok[i] & 2u) ? v.y : 0.f;
            v.z = (pok[i] & 4u) ? v.z : 0.f;
            v.w = (pok[i] & 8u) ? v.w : 0.f;
            *reinterpret_cast<float4*>(&Bs0[buf * (BK * BN) + (b_r + i * Shape::B_ROWS_PER_PASS) * BN + b_c4 * 4]) = v;
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    fetch(0);
    // bias of the 16 rows this lane stores: row = (r & 3) + 8 (r >> 2) + 4 half
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = Policy::bias_at(prm, m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half);
    if (k_tiles > 1)
    {
        float4 qa[Shape::A_PASSES];
        BRaw qb[Shape::B_PASSES];
        unsigned qok[Shape::B_PASSES];
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i) qa[i] = aload.load(prm, BK + a_r + i * Shape::A_ROWS_PER_PASS);
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i) qb[i] = bload.load(prm, BK + b_r + i * Shape::B_ROWS_PER_PASS, qok[i]);
        stash(0, 0);
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i) pa[i] = qa[i];
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i)
        {
            pb[i] = qb[i];
            pok[i] = qok[i];
        }
    }
    else
        stash(0, 0);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();

    const int a_off = half * BM + wm * 32 + l31;
    const int b_off = half * BN + wn * 128 + 4 * l31;
    int cur = 0;
    for (int kt = 0; kt < k_tiles; ++kt)
    {
        if (kt + 1 < k_tiles) stash(cur ^ 1, kt + 1);
        if (kt + 2 < k_tiles) fetch(kt + 2);
        const float* as = As0 + cur * (BK * BM) + a_off;
        const float* bs = Bs0 + cur * (BK * BN) + b_off;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp)
        {
            const float fa = as[(2 * kp) * BM];
            const float4 fb = *reinterpret_cast<const float4*>(&bs[(2 * kp) * BN]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb.w, acc[3], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: register r of the four accumulators = columns 4 l .. 4 l + 3 of row (r & 3) + 8 (r >> 2) + 4 half: straight to memory, the
    // residual operand requested four rows ahead
    const typename Policy::Store st(prm, batch, n0 + wn * 128 + 4 * l31);
    const int mrow = m0 + wm * 32 + 4 * half;
    float4 res[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) res[q] = st.residual4(prm, mrow + q);
#pragma unroll
    for (int g = 0; g < 4; ++g)
    {
        float4 nxt[4];
        if (g + 1 < 4)
        {
#pragma unroll
            for (int q = 0; q < 4; ++q) nxt[q] = st.residual4(prm, mrow + 8 * (g + 1) + q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            const int r = 4 * g + q;
            st.put4b(prm, mrow + 8 * g + q, make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]), bias_r[r], res[q]);
        }
        if (g + 1 < 4)
        {
#pragma unroll
            for (int q = 0; q < 4; ++q) res[q] = nxt[q];
        }
    }
}

} // namespace fhip
