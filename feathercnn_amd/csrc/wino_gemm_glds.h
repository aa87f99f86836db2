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

#ifndef FHIP_GLDS_NT
#define FHIP_GLDS_NT 0 // default of the NT template parameters below (tools); the product's launches pass FHIP_M_NT_BIG / FHIP_M_NT_SMALL
#endif
// NT bit 0: the V pieces are requested with the non-temporal hint (aux = 2 of global_load_lds); bit 1: M leaves through `nt` stores
// SPLIT: the launch carries row pieces behind its whole tiles (prm.tail_first / tail_parts); false = the plain kernel, untouched
template <int NBUF, int BK = 16, int OCC = 6, int NT = FHIP_GLDS_NT, bool SPLIT = false>
__global__ __launch_bounds__(256, BK == 16 ? OCC : 3) void wino_gemm_glds_kernel(const WinoGemmPolicy::Params prm)
{
    constexpr int BM = 128, BN = 64, EPI_LD = 36;
    constexpr int RPW = BK / 4; // k rows per wave and tile
    constexpr int BUF_FLOATS = BK * (BM + BN); // A [16][128] then B [16][64]
    constexpr int LDSF = NBUF * BUF_FLOATS > 4 * 32 * EPI_LD ? NBUF * BUF_FLOATS : 4 * 32 * EPI_LD;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];

    const int nwg = prm.batches * prm.m_tiles * prm.n_tiles;
    // whole tiles first (XCD-aware order); behind them the row pieces of the last tiles, the pieces of one tile on one XCD (block b runs on
    // XCD b % 8 and tail_first is a multiple of the CU count)
    const int parts = SPLIT ? prm.tail_parts : 1;
    const bool whole = !SPLIT || parts == 1 || (int)blockIdx.x < prm.tail_first;
    int vid, part = 0;
    if (!SPLIT || parts == 1)
        vid = xcd_remap(blockIdx.x, nwg);
    else if (whole)
        vid = xcd_remap(blockIdx.x, prm.tail_first);
    else
    {
        const int e = blockIdx.x - prm.tail_first, grp = e >> 3;
        part = grp % parts;
        vid = prm.tail_first + (grp / parts) * 8 + (e & 7);
        if (vid >= nwg) return; // (tail tiles) not a multiple of 8: the surplus blocks of the last group
    }
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
    const float* srcB = prm.V + (size_t)xi * prm.Lv.xis + prm.Lv.col(n0) + (lane & 15) * 4; // the 64-column tile lies inside one column block
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
            __builtin_amdgcn_global_load_lds(srcB + (size_t)r * prm.Lv.bp, (lds_void*)(base + BK * BM + (wave * RPW + 4 * i) * BN), 16, 0, (NT & 1) ? 2 : 0);
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

    // rows of this wave inside the 128-row tile: a whole tile gives wave (wm, wn) rows 64 wm .. + 63 (two accumulators); a half piece
    // (64 rows) gives it rows 64 part + 32 wm .. + 31, a quarter piece (32 rows) rows 32 part .. + 31 on the waves wm = 0 only (one accumulator)
    const int row_off = whole ? wm * 64 : parts == 2 ? part * 64 + wm * 32 : part * 32;
    const bool idle = SPLIT && !whole && parts == 4 && wm == 1; // still issues its share of the operand loads and meets every barrier
    const int a_off = half * BM + row_off + l31;
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
        if (whole)
        {
#pragma unroll
            for (int kp = 0; kp < BK / 2; ++kp)
            {
                const float fa0 = as[(2 * kp) * BM], fa1 = as[(2 * kp) * BM + 32], fb = bs[(2 * kp) * BN];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb, acc[1], 0, 0, 0);
            }
        }
        else if (!idle)
        {
#pragma unroll
            for (int kp = 0; kp < BK / 2; ++kp)
            {
                const float fa0 = as[(2 * kp) * BM], fb = bs[(2 * kp) * BN];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb, acc[0], 0, 0, 0);
            }
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
    if (idle) return;
    float* const scr = lds + wave * (32 * EPI_LD);
    const int e_row = lane >> 3, e_c4 = (lane & 7) * 4;
    float* mbase = prm.M + (size_t)xi * prm.Lm.xis + prm.Lm.col(n0) + wn * 32 + e_c4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
    {
        if (i == 1 && !whole) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * half) * EPI_LD + l31] = acc[i][r];
        const int mrow = m0 + row_off + i * 32 + e_row;
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            const float4 v = *reinterpret_cast<const float4*>(&scr[(q * 8 + e_row) * EPI_LD + e_c4]);
            const int m = mrow + q * 8;
            if (m < prm.K)
            {
                if constexpr (NT & 2)
                    stg4_nt(mbase + (size_t)m * prm.Lm.bp, v);
                else
                    *reinterpret_cast<float4*>(mbase + (size_t)m * prm.Lm.bp) = v;
            }
        }
    }
}

// The same kernel with a 128 x 96 tile, for column counts that 64-column tiles pad badly (round 3).  VGG-16's conv5 layers at batch 32 have
// P = 9 tiles x 32 images = 288 columns = 4.5 tiles of 64: a ninth of the executed MFMAs multiplied padding (100 TF algorithmic = 111 TF
// executed).  288 = 3 x 96 exactly, and 4 x 3 x 64 blocks are 3 per CU -- one round, everything resident.  Waves 4 x 1: a wave owns 32
// rows x 96 columns = three 32x32 accumulators sharing one A fragment (1 A + 3 B LDS reads per 3 MFMAs).  B tile [16][96] is six 1-KB
// LDS-DMA pieces (lane -> row e / 24, 16-byte column e % 24), A eight; two buffers (28 KB), 5 blocks per CU.
template <int NT = FHIP_GLDS_NT>
__global__ __launch_bounds__(256, 5) void wino_gemm_glds96_kernel(const WinoGemmPolicy::Params prm)
{
    constexpr int BM = 128, BN = 96, BK = 16, EPI_LD = 36, NBUF = 2;
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
    const int l31 = lane & 31, half = lane >> 5;

    // A: wave w, piece i covers rows 4w + 2i, 4w + 2i + 1 (32 lanes x 16 B per row), as in the 64-column kernel
    const float* srcA = prm.U + (size_t)xi * prm.Cp * prm.Kp + (size_t)(wave * 4 + half) * prm.Kp + m0 + l31 * 4;
    const size_t a_step = (size_t)BK * prm.Kp;
    // B: piece j = wave (and wave + 4 for waves 0, 1) is float4 elements 64 j .. 64 j + 63 of the [16][24 float4] tile
    // (a 96-column tile may straddle column blocks of V / M: every 16-byte piece is addressed through the layout; whole rows otherwise)
    const float* srcB = prm.V + (size_t)xi * prm.Lv.xis;
    const int e0 = wave * 64 + lane, e1 = (wave + 4) * 64 + lane;
    const int b0_row = e0 / 24, b1_row = e1 / 24;
    const size_t b0_col = prm.Lv.col(n0 + (e0 - b0_row * 24) * 4), b1_col = prm.Lv.col(n0 + (e1 - b1_row * 24) * 4);

    auto issue = [&](int kt, int buf) {
        float* base = lds + buf * BUF_FLOATS;
        const float* a = srcA + (size_t)kt * a_step;
        __builtin_amdgcn_global_load_lds(a, (lds_void*)(base + (wave * 4) * BM), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(a + (size_t)2 * prm.Kp, (lds_void*)(base + (wave * 4 + 2) * BM), 16, 0, 0);
        float* bb = base + BK * BM;
        {
            const int r = min(kt * BK + b0_row, prm.C - 1);
            __builtin_amdgcn_global_load_lds(srcB + (size_t)r * prm.Lv.bp + b0_col, (lds_void*)(bb + wave * 256), 16, 0, (NT & 1) ? 2 : 0);
        }
        if (wave < 2)
        {
            const int r = min(kt * BK + b1_row, prm.C - 1);
            __builtin_amdgcn_global_load_lds(srcB + (size_t)r * prm.Lv.bp + b1_col, (lds_void*)(bb + (wave + 4) * 256), 16, 0, (NT & 1) ? 2 : 0);
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int a_off = half * BM + wave * 32 + l31;
    const int b_off = BK * BM + half * BN + l31;
    int cur = 0;
    for (int kt = 0; kt < k_tiles; ++kt)
    {
        if (kt + 1 < k_tiles) issue(kt + 1, cur ^ 1);
        const float* as = lds + cur * BUF_FLOATS + a_off;
        const float* bs = lds + cur * BUF_FLOATS + b_off;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp)
        {
            const float fa = as[(2 * kp) * BM], fb0 = bs[(2 * kp) * BN], fb1 = bs[(2 * kp) * BN + 32], fb2 = bs[(2 * kp) * BN + 64];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb2, acc[2], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }

    float* const scr = lds + wave * (32 * EPI_LD);
    const int e_row = lane >> 3, e_c4 = (lane & 7) * 4;
    const int mrow = m0 + wave * 32 + e_row;
#pragma unroll
    for (int j = 0; j < 3; ++j)
    {
        float* mbase = prm.M + (size_t)xi * prm.Lm.xis + prm.Lm.col(n0 + j * 32 + e_c4);
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * half) * EPI_LD + l31] = acc[j][r];
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

// Row split of the last tiles (round 5).  A launch of `tiles` whole 128 x 64 tiles on `cus` CUs leaves tiles % cus CUs with one tile more
// than the others; when that remainder is at most a quarter (half) of the CUs, its tiles are cut into 4 (2) row pieces of 32 (64) rows --
// one block each, so every CU gets at most one PIECE more.  ResNet-50 b64's res5 layers (F(4x4,3x3): 36 x 4 x 4 = 576 tiles = 2.25 per
// CU) run 2 tiles + 1 quarter per CU instead of 3 tiles on a quarter of the CUs.  Only where a tile is a large share of a CU's work
// (<= max_rounds tiles per CU): measured with tools/gemm_bench.hip GEMM_SPLIT=1.  -> tail_first (== tiles: no split), tail_parts
inline void wino_gemm_row_split(int tiles, int cus, int max_rounds, int& tail_first, int& tail_parts)
{
    tail_first = tiles;
    tail_parts = 1;
    const int r = tiles % cus;
    if (r == 0 || tiles < cus || tiles > max_rounds * cus || (cus & 7)) return;
    if (r * 4 <= cus)
        tail_parts = 4;
    else if (r * 2 <= cus)
        tail_parts = 2;
    else
        return;
    tail_first = tiles - r;
}
inline int wino_gemm_row_split_grid(int tiles, int tail_first, int tail_parts)
{
    return tail_parts == 1 ? tiles : tail_first + round_up(tiles - tail_first, 8) * tail_parts;
}

// does the 96-column tile pad fewer columns than the 64-column one?  (pure function of the column count)
// ... or, with neither padding, are the columns so few that whole 96-column tiles give one balanced round of fewer, larger blocks?
// (ResNet-50 b64's 14-pixel 3x3 layers: P = 576 = 9 x 64 = 6 x 96 -- 1152 blocks, 4.5 per CU, against 768, 3 per CU: 51.3 -> 47.8 us in
// tools/gemm_bench.hip (GEMM_RESNET=1 GEMM_96=1); P = 1600 and P = 256 stay faster on 64 columns)
inline bool wino_gemm_prefers_96(int columns)
{
    return ceil_div(columns, 96) * 96 < ceil_div(columns, 64) * 64 || (columns % 96 == 0 && columns <= 576);
}

} // namespace fhip
