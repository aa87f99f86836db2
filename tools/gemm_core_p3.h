// gemm_core_p3.h -- EXPERIMENT: the main loop of gemm_core.h with three LDS buffers, the per-k-tile barrier in the middle of
// the tile's MFMAs and the next tile's first fragments prefetched before the tile ends.  Same policies, same epilogue.
#pragma once

#include "gemm_core.h"

namespace fhip
{

template <class Shape, class Policy, int ABLATE = 0>
__global__ __launch_bounds__(Shape::THREADS, Shape::BLOCKS_PER_CU* Shape::THREADS / 256) void gemm_mfma_kernel_p3(
    const typename Policy::Params prm)
{
    constexpr int BM = Shape::BM, BN = Shape::BN, BK = Shape::BK;
    // ONE LDS object (a second __shared__ object de-pipelines hipcc's waits)
    constexpr int OPER3 = 3 * BK * (BM + BN);
    constexpr int LDS3 = OPER3 > Shape::EPI_FLOATS ? OPER3 : Shape::EPI_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS3];
    float* const As0 = lds;               // As[buf] = As0 + buf * BK*BM, buf in 0..2
    float* const Bs0 = lds + 3 * BK * BM; // Bs[buf] = Bs0 + buf * BK*BN

    const int nwg = prm.batches * prm.m_tiles * prm.n_tiles;
    int vid = xcd_remap(blockIdx.x, nwg);
    const int mt = vid % prm.m_tiles;
    vid /= prm.m_tiles;
    const int nt = vid % prm.n_tiles;
    const int batch = vid / prm.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int k_tiles = prm.k_tiles;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Shape::WAVES_N, wn = wave % Shape::WAVES_N;
    const int l31 = lane & 31, half = lane >> 5;

    // loader mapping: a thread always fetches the same 4 consecutive m (n) of rows r, r + ROWS_PER_PASS, ...
    const int a_c4 = tid % Shape::A_F4_PER_ROW, a_r = tid / Shape::A_F4_PER_ROW;
    const int b_c4 = tid % Shape::B_F4_PER_ROW, b_r = tid / Shape::B_F4_PER_ROW;
    const typename Policy::ALoad aload(prm, batch, m0 + a_c4 * 4);
    const typename Policy::BLoad bload(prm, batch, n0 + b_c4 * 4);

    float4 pa[Shape::A_PASSES], pb[Shape::B_PASSES];
    unsigned pok[Shape::B_PASSES];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i) pa[i] = aload.load(prm, kt * BK + a_r + i * Shape::A_ROWS_PER_PASS);
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i) pb[i] = bload.load(prm, kt * BK + b_r + i * Shape::B_ROWS_PER_PASS, pok[i]);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i)
            *reinterpret_cast<float4*>(&As0[buf * (BK * BM) + (a_r + i * Shape::A_ROWS_PER_PASS) * BM + a_c4 * 4]) = pa[i];
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i)
        {
            float4 v = pb[i];
            v.x = (pok[i] & 1u) ? v.x : 0.f;
            v.y = (pok[i] & 2u) ? v.y : 0.f;
            v.z = (pok[i] & 4u) ? v.z : 0.f;
            v.w = (pok[i] & 8u) ? v.w : 0.f;
            *reinterpret_cast<float4*>(&Bs0[buf * (BK * BN) + (b_r + i * Shape::B_ROWS_PER_PASS) * BN + b_c4 * 4]) = v;
        }
    };

    f32x16 acc[Shape::TM][Shape::TN];
#pragma unroll
    for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
        for (int j = 0; j < Shape::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: k-tile 0 -> LDS buffer 0, k-tile 1 -> registers.  Both tiles' loads are issued back to back
    // (a second register set for a moment), so the block pays ONE global round trip before its first MFMA, not two.
    fetch(0);
    if (k_tiles > 1)
    {
        float4 qa[Shape::A_PASSES], qb[Shape::B_PASSES];
        unsigned qok[Shape::B_PASSES];
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i) qa[i] = aload.load(prm, BK + a_r + i * Shape::A_ROWS_PER_PASS);
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i) qb[i] = bload.load(prm, BK + b_r + i * Shape::B_ROWS_PER_PASS, qok[i]);
        stash(0); // waits for tile 0's loads only (vmcnt counts in order)
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
        stash(0);
    __syncthreads();

    const int a_off = half * BM + wm * Shape::WTM + l31;
    const int b_off = half * BN + wn * Shape::WTN + l31;
    // Three LDS buffers, ONE barrier per k-tile placed in the MIDDLE of the tile's MFMAs:
    //   iteration t: stash tile t+1 -> buf[(t+1)%3] | request tile t+2 | MFMA k-steps 0..3 of tile t | barrier |
    //                MFMA k-steps 4..7, and before the last one the first fragments of tile t+1 are fetched.
    // The barrier publishes tile t+1 while tile t's MFMAs are still in flight, and the next tile's first operands are in
    // registers before its iteration starts: no LDS round trip and no barrier between the last MFMA of a tile and the first of
    // the next.  (Two buffers would need a second barrier: a fast wave's stash of tile t+1 could overwrite what a slow wave
    // still reads in the second half of tile t-1.)
    float fa_n[Shape::TM], fb_n[Shape::TN];
    {
        const float* as = As0 + a_off;
        const float* bs = Bs0 + b_off;
#pragma unroll
        for (int i = 0; i < Shape::TM; ++i) fa_n[i] = as[i * 32];
#pragma unroll
        for (int j = 0; j < Shape::TN; ++j) fb_n[j] = bs[j * 32];
    }
    int cur = 0;
    for (int kt = 0; kt < k_tiles; ++kt)
    {
        const int nxt = cur == 2 ? 0 : cur + 1;
        if (kt + 1 < k_tiles) stash(nxt);
        if (kt + 2 < k_tiles && !(ABLATE & 1)) fetch(kt + 2);

        const float* as = As0 + cur * (BK * BM) + a_off;
        const float* bs = Bs0 + cur * (BK * BN) + b_off;
        float fa[Shape::TM], fbv[Shape::TN];
#pragma unroll
        for (int i = 0; i < Shape::TM; ++i) fa[i] = fa_n[i];
#pragma unroll
        for (int j = 0; j < Shape::TN; ++j) fbv[j] = fb_n[j];
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp)
        {
            if (kp == BK / 4) __syncthreads();
            float ga[Shape::TM], gb[Shape::TN];
            if (kp + 1 < BK / 2)
            {
#pragma unroll
                for (int i = 0; i < Shape::TM; ++i) ga[i] = as[(2 * (kp + 1)) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < Shape::TN; ++j) gb[j] = bs[(2 * (kp + 1)) * BN + j * 32];
            }
            else
            {
                // last k-step: its operands are already here; fetch the first fragments of the next tile instead
                const float* an = As0 + nxt * (BK * BM) + a_off;
                const float* bn = Bs0 + nxt * (BK * BN) + b_off;
#pragma unroll
                for (int i = 0; i < Shape::TM; ++i) ga[i] = an[i * 32];
#pragma unroll
                for (int j = 0; j < Shape::TN; ++j) gb[j] = bn[j * 32];
            }
#pragma unroll
            for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
                for (int j = 0; j < Shape::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fbv[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < Shape::TM; ++i) fa[i] = ga[i];
#pragma unroll
            for (int j = 0; j < Shape::TN; ++j) fbv[j] = gb[j];
        }
#pragma unroll
        for (int i = 0; i < Shape::TM; ++i) fa_n[i] = fa[i];
#pragma unroll
        for (int j = 0; j < Shape::TN; ++j) fb_n[j] = fbv[j];
        cur = nxt;
    }
    __syncthreads(); // the epilogue reuses the operand buffers as scratch

    // ---- epilogue.  After the last barrier nobody reads the operand buffers any more: each wave transposes its
    // 32x32 MFMA tiles through a private piece of them (C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2)
    // + 4 * (lane >> 5)) and stores 4 consecutive columns per lane.  Wave-private + in-order LDS queue: no barrier.
    if (ABLATE & 2)
    {
#pragma unroll
        for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
            for (int j = 0; j < Shape::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[i][j][r]));
        return;
    }
    float* const scr = lds + wave * (32 * Shape::EPI_LD);
    const int e_row = lane >> 3, e_c4 = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < Shape::TN; ++j)
    {
        const typename Policy::Store st(prm, batch, n0 + wn * Shape::WTN + j * 32 + e_c4);
#pragma unroll
        for (int i = 0; i < Shape::TM; ++i)
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * half) * Shape::EPI_LD + l31] = acc[i][j][r];
            const int mbase = m0 + wm * Shape::WTM + i * 32 + e_row;
#pragma unroll
            for (int q = 0; q < 4; ++q)
            {
                const float4 v = *reinterpret_cast<const float4*>(&scr[(q * 8 + e_row) * Shape::EPI_LD + e_c4]);
                st.put4(prm, mbase + q * 8, v);
            }
        }
    }
}

} // namespace fhip
