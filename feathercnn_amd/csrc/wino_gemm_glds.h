// wino_gemm_glds.h -- the Winograd tile GEMM (K > 64) with LDS-DMA operand staging: global_load_lds_dwordx4 writes the operand
// tiles straight into LDS (no staging registers, no ds_write pass), counted s_waitcnt and raw s_barrier instead of
// __syncthreads (whose fence drains the DMA queue).  128x64x16 tile, 4 waves (2x2), same wave tiling, fragment reads and
// epilogue as gemm_core.h.  NBUF LDS buffers, loads issued NBUF-1 k-tiles ahead; the product uses NBUF = 2 (measured, 40
// repetitions per VGG-16 shape in tools/gemm_bench.hip: +1.3 ... +3.3 % over the register-staged kernel; 3 buffers gave less).
// Ordering rules (cdna_hip_programming.md, LDS-DMA): a wave waits vmcnt for ITS pieces of tile t+1, then the barrier; readers
// touch that buffer only after the barrier (RAW); a buffer is re-filled only after every wave waited lgkmcnt(0) for its reads
// of it and passed the barrier (WAR).
// No masks are needed: U is zero padded in both dimensions, V rows past C are clamped to row C-1 (finite data) and meet zero
// rows of U; V columns past P only feed M columns nobody reads.
#pragma once

#include "gemm_core.h"
#include "wino_gemm_policy.h"

namespace fhip
{

typedef __attribute__((address_space(3))) void lds_void;

template <int NBUF, int BK = 16, int OCC = 6>
__global__ __launch_bounds__(256, BK == 16 ? OCC : 3) void wino_gemm_glds_kernel(const WinoGemmPolicy::Params prm)
{
    constexpr int BM = 128, BN = 64, EPI_LD = 36;
    constexpr int RPW = BK / 4; // k rows per wave and tile
    constexpr int BUF_FLOATS = BK * (BM + BN); // A [16][128] then B [16][64]
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

    // per-lane sources.  A: wave w, piece i covers rows 4w + 2i, 4w + 2i + 1 (32 lanes x 16 B per row)
    const float* srcA = prm.U + (size_t)xi * prm.Cp * prm.Kp + (size_t)(wave * RPW + half) * prm.Kp + m0 + l31 * 4;
    // B: wave w covers rows RPW*w .. RPW*w + RPW-1 (16 lanes x 16 B per row, 4 rows per piece)
    const int brow = wave * RPW + (lane >> 4);
    const float* srcB = prm.V + (size_t)xi * prm.C * prm.Pp + n0 + (lane & 15) * 4;
    const size_t a_step = (size_t)BK * prm.Kp;

    auto issue = [&](int kt, int buf) {
        float* base = lds + buf * BUF_FLOATS;
        const float* a = srcA + (size_t)kt * a_step;
#pragma unroll
        for (int i = 0; i < RPW / 2; ++i)
            __builtin_amdgcn_global_load_lds(a + (size_t)(2 * i) * prm.Kp, (lds_void*)(base + (wave * RPW + 2 * i) * BM), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < RPW / 4; ++i)
        {
            const int r = min(kt * BK + brow + 4 * i, prm.C - 1);
            __builtin_amdgcn_global_load_lds(srcB + (size_t)r * prm.Pp, (lds_void*)(base + BK * BM + (wave * RPW + 4 * i) * BN), 16, 0, 0);
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // prologue: NBUF-1 tiles in flight
#pragma unroll
    for (int p = 0; p < NBUF - 1; ++p)
        if (p < k_tiles) issue(p, p);
    if (NBUF == 3 && k_tiles > 1)
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int a_off = half * BM + wm * 64 + l31;
    const int b_off = BK * BM + half * BN + wn * 32 + l31;
    int cur = 0;
    for (int kt = 0; kt < k_tiles; ++kt)
    {
        const int ahead = kt + NBUF - 1;
        int nb = cur + NBUF - 1;
        if (nb >= NBUF) nb -= NBUF;
        const bool more = ahead < k_tiles;
        if (more) issue(ahead, nb);

        const float* as = lds + cur * BUF_FLOATS + a_off;
        const float* bs = lds + cur * BUF_FLOATS + b_off;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp)
        {
            const float fa0 = as[(2 * kp) * BM], fa1 = as[(2 * kp) * BM + 32], fb = bs[(2 * kp) * BN];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb, acc[1], 0, 0, 0);
        }
        // my pieces of the NEXT tile have landed (the tile just issued may stay in flight), my LDS reads are done
        if (NBUF == 3 && more)
            asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur + 1 == NBUF ? 0 : cur + 1;
    }

    // epilogue (gemm_core.h): per-wave LDS transpose, 16-byte row stores
    float* const scr = lds + wave * (32 * EPI_LD);
    const int e_row = lane >> 3, e_c4 = (lane & 7) * 4;
    float* mbase = prm.M + (size_t)xi * prm.K * prm.Pp + n0 + wn * 32 + e_c4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
    {
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * half) * EPI_LD + l31] = acc[i][r];
        const int mrow = m0 + wm * 64 + i * 32 + e_row;
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            const float4 v = *reinterpret_cast<const float4*>(&scr[(q * 8 + e_row) * EPI_LD + e_c4]);
            const int m = mrow + q * 8;
            if (m < prm.K) *reinterpret_cast<float4*>(mbase + (size_t)m * prm.Pp) = v;
        }
    }
}

} // namespace fhip
