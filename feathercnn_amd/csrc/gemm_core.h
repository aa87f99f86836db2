// gemm_core.h -- the one fp32-MFMA GEMM main loop shared by
//   * the Winograd tile GEMM  M_xi[K x P] = U_xi[K x C] * V_xi[C x P]   (reference TensorGEMM,
//     src/booster/avx/winograd_kernels_F63.cpp:518-692), batch index = frequency point xi (64 of them);
//   * the implicit-GEMM convolution out[K x N*Ho*Wo] = W[K x C*kh*kw] * im2col(in)  (reference
//     IM2COL_Forward = booster::im2col + packed_sgemm_activation, avx/booster.cpp:83-102,
//     avx/generic_kernels.cpp:50-85, avx/sgemm.cpp:377-433) where the im2col matrix is never materialised:
//     the B-operand loader gathers straight from the NCHW input.
//
// CDNA4 mapping (not a translation of the reference's 6x16 AVX micro-kernel):
//   * v_mfma_f32_32x32x2_f32: exact fp32, 64 cycles / SIMD, one VGPR per operand
//     (A: lane l holds A[i = l & 31][k = l >> 5]; B: B[k = l >> 5][j = l & 31]).
//   * Both operand tiles live in LDS k-major ([BK][BM] / [BK][BN], the m / n index contiguous), so an
//     operand fetch is one conflict-free ds_read_b32 per lane (two 32-lane halves read two k rows) and a
//     global->LDS copy is a straight 16-byte-per-lane row copy.  fp32 MFMA is 1/16 the bf16 rate, so LDS
//     bandwidth is nowhere near the limit (16 B/clk/CU used of 128); what decides the rate is whether the
//     matrix pipe is issued back to back.  Measured on MI355X (tools/gemm_bench.hip, tools/loop_probe.hip):
//     this loop structure (LDS operands + one barrier and one LDS write pass per k-tile) sustains ~91 % of
//     the 157 TF peak with >= 2 waves per SIMD; what takes it away is whatever stalls a wave in front of its
//     MFMAs.  The rules this kernel follows, each from a measurement:
//       - one output tile per block, 4 blocks per CU, hardware block scheduling.  A persistent variant with
//         a static or an atomic tile scheduler was built and measured SLOWER (52-66 % vs 58-70 %): with ~12
//         tiles per CU the static split loses 17 % to unlucky CUs, a contended atomic costs microseconds per
//         tile, and lock-stepped blocks collide in the CU's single VMEM address pipe;
//       - global loads are issued TWO k-tiles ahead (registers); the LDS write of k-tile t+1 happens at the top
//         of iteration t, a full k-tile after its loads were issued, so its vmcnt wait is free; one barrier
//         per k-tile;
//       - every global load is UNCONDITIONAL (clamped address + validity mask applied at LDS-write time): a
//         load under a branch makes hipcc wait vmcnt(0) right behind it (the merge with the zero needs the
//         value), which serialises the pipeline on the global round trip (measured 66 % -> 50 %);
//       - accumulators leave through a per-wave LDS transpose (in the operand buffers, free by then) as 16-byte
//         row stores: a VMEM instruction costs the CU's address pipe ~16 clk whatever its width, so 64 dword
//         stores per lane were 4096 clk of a C=64 tile's 8192 MFMA clk; 16 dwordx4 stores are 1024.
//   * XCD-aware: block b runs on XCD b % 8 (observed; speed only, never correctness).  The 1-D grid is remapped
//     so consecutive virtual ids land on one XCD and walk m-tiles fastest: blocks that share a B panel / an A
//     panel hit the same private 4 MiB L2 at the same time.
#pragma once
#include <type_traits>

#include "common.h"

namespace fhip
{

template <int BM_, int BN_, int BK_, int WAVES_M_, int WAVES_N_, int BLOCKS_PER_CU_ = 4>
struct GemmShape
{
    static constexpr int BM = BM_, BN = BN_, BK = BK_;
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
    static constexpr int WAVES = WAVES_M * WAVES_N;
    static constexpr int BLOCKS_PER_CU = BLOCKS_PER_CU_;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N; // per-wave output tile
    static constexpr int TM = WTM / 32, TN = WTN / 32;           // 32x32 MFMA tiles per wave
    static constexpr int A_F4_PER_ROW = BM / 4, B_F4_PER_ROW = BN / 4;
    static constexpr int A_ROWS_PER_PASS = THREADS / A_F4_PER_ROW, B_ROWS_PER_PASS = THREADS / B_F4_PER_ROW;
    static constexpr int A_PASSES = BK / A_ROWS_PER_PASS, B_PASSES = BK / B_ROWS_PER_PASS;
    static constexpr int OPERAND_FLOATS = 2 * BK * (BM + BN); // double-buffered A and B tiles
    static constexpr int EPI_LD = 36;                         // row pitch of the 32x32 transpose scratch (16-B aligned rows)
    static constexpr int EPI_FLOATS = WAVES * 32 * EPI_LD;    // one scratch per wave, overlaid on the operand buffers
    static constexpr int LDS_FLOATS = OPERAND_FLOATS > EPI_FLOATS ? OPERAND_FLOATS : EPI_FLOATS;
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA");
    static_assert(THREADS % A_F4_PER_ROW == 0 && THREADS % B_F4_PER_ROW == 0, "loader mapping");
    static_assert(A_PASSES >= 1 && B_PASSES >= 1 && BK % A_ROWS_PER_PASS == 0 && BK % B_ROWS_PER_PASS == 0, "BK too small");
    static_assert(BK % 2 == 0, "MFMA k-depth is 2");
};

// Policy concept:
//   struct Params { int batches, m_tiles, n_tiles, k_tiles; ... };
//   struct ALoad { ALoad(const Params&, int batch, int m4); float4 load(const Params&, int krow) const; };
//   struct BLoad { BLoad(const Params&, int batch, int n4); typedef ... Raw; Raw load(const Params&, int krow, unsigned& ok) const;
//                  float4 finish(const Params&, const Raw&, int krow, const float* extra_lds) const; };
//       (Raw = what the global loads of one request return -- float4 for a plain operand; `finish` turns it into the 4 operand values at
//        LDS-write time, a k-tile after the loads were issued: identity for a plain operand, the 3x3 depthwise arithmetic for the fused
//        depthwise + pointwise route)
//   static constexpr int EXTRA_LDS_FLOATS; static void stage_extra(const Params&, float* extra_lds, int tid, int threads);
//       (block-wide constants the B loader wants in LDS -- the depthwise taps; 0 / no-op otherwise)
//       (m4 / n4 = first of the 4 consecutive rows / columns this thread always fetches; loads are unconditional
//        from clamped addresses, `ok` bit e = element e is real data, zero-fill happens at LDS-write time)
//   struct Store { Store(const Params&, int batch, int n4); void put4(const Params&, int m, float4 v) const;
//                  float4 residual4(const Params&, int m) const;   (the fused residual operand of row m's 4 columns, requested early; zeros if none)
//                  void put4b(const Params&, int m, float4 v, float bias_m, float4 res) const; }   (bias and residual come from the caller)
//   static float bias_at(const Params&, int m);   bias of output row m (0 when there is none / the row does not exist)
//       (4 consecutive output columns n4..n4+3 of row m)
//   static int k_count(const Params&, int batch);   k-tiles this batch entry reduces over (the loaders know where they start)
//
// Block life cycle (round 2, from per-block clock stamps of the instrumented copy of this loop, tools/experiments/gemm_core_probe.h:
// a 128x64x64 tile of ResNet-50's 1x1 layers spent ~7000 cycles in its index set-up, ~16000 in the k-loop and 10000-30000 in the epilogue):
//   * the block's set-up runs at raised wave priority.  The SIMD arbitrates issue slots by priority, then AGE: a freshly
//          launched wave is the youngest on its SIMD and only gets the slots the older, MFMA-issuing waves leave over, so the few
//          hundred VALU instructions in front of its first load took microseconds -- with every other block's prologue latency
//          behind them;
//   * the lane's bias values are requested in the prologue (Policy::bias_at) instead of one dependent global load in front of
//          every accumulator store (8 serialised L2 round trips per wave and tile).
// Measured on ResNet-50 / MobileNet 1x1 layers: +3 ... +7 % on the shallow ones (C <= 128), nothing on the deep ones.  Two other
// epilogues were measured and dropped: storing straight from the accumulators of an MFMA with swapped operand
// roles (32-byte store pieces: -20 %) and batching the LDS transpose of a whole 32-column piece (block latency -3000 cycles,
// throughput unchanged).
template <class P, class = void>
struct gemm_has_decode : std::false_type
{
};
template <class P>
struct gemm_has_decode<P, std::void_t<decltype(&P::decode)>> : std::true_type
{
};

template <class Shape, class Policy>
__global__ __launch_bounds__(Shape::THREADS, Shape::BLOCKS_PER_CU* Shape::THREADS / 256) void gemm_mfma_kernel(
    const typename Policy::Params prm)
{
    constexpr int BM = Shape::BM, BN = Shape::BN, BK = Shape::BK;
    // ONE LDS object (a second __shared__ object de-pipelines hipcc's waits)
    __shared__ __attribute__((aligned(16))) float lds[Shape::LDS_FLOATS + Policy::EXTRA_LDS_FLOATS];
    float* const extra = lds + Shape::LDS_FLOATS; // Policy::stage_extra's block-wide constants (behind the operand / epilogue area)
    float* const As0 = lds;               // As[buf] = As0 + buf * BK*BM
    float* const Bs0 = lds + 2 * BK * BM; // Bs[buf] = Bs0 + buf * BK*BN

    __builtin_amdgcn_s_setprio(3);
    int mt, nt, batch;
    if constexpr (gemm_has_decode<Policy>::value)
        Policy::decode(prm, mt, nt, batch); // a policy with a block order of its own (ConvGemmPolicy: the tail split)
    else
    {
        int vid = xcd_remap(blockIdx.x, prm.batches * prm.m_tiles * prm.n_tiles);
        mt = vid % prm.m_tiles;
        vid /= prm.m_tiles;
        nt = vid % prm.n_tiles;
        batch = vid / prm.n_tiles;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int k_tiles = Policy::k_count(prm, batch); // tiles of THIS batch entry (a split-K piece may be uneven)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Shape::WAVES_N, wn = wave % Shape::WAVES_N;
    const int l31 = lane & 31, half = lane >> 5;

    // loader mapping: a thread always fetches the same 4 consecutive m (n) of rows r, r + ROWS_PER_PASS, ...
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
    if (Policy::EXTRA_LDS_FLOATS > 0)
    {
        Policy::stage_extra(prm, extra, tid, Shape::THREADS); // behind the first operand requests; the first finish() needs it
        __syncthreads();
    }
    // bias of the rows this lane will store (row = .. + i*32 + q*8 + (lane >> 3)), requested behind the first operand tile
    float bias_r[Shape::TM][4];
#pragma unroll
    for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            bias_r[i][q] = Policy::bias_at(prm, m0 + wm * Shape::WTM + i * 32 + q * 8 + (lane >> 3));
    if (k_tiles > 1)
    {
        float4 qa[Shape::A_PASSES];
        BRaw qb[Shape::B_PASSES];
        unsigned qok[Shape::B_PASSES];
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i) qa[i] = aload.load(prm, BK + a_r + i * Shape::A_ROWS_PER_PASS);
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i) qb[i] = bload.load(prm, BK + b_r + i * Shape::B_ROWS_PER_PASS, qok[i]);
        stash(0, 0); // waits for tile 0's loads only (vmcnt counts in order)
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

    const int a_off = half * BM + wm * Shape::WTM + l31;
    const int b_off = half * BN + wn * Shape::WTN + l31;
    int cur = 0;
    for (int kt = 0; kt < k_tiles; ++kt)
    {
        // k-tile kt+1 (in registers since the previous iteration) -> the other LDS buffer; k-tile kt+2 -> registers
        if (kt + 1 < k_tiles) stash(cur ^ 1, kt + 1);
        if (kt + 2 < k_tiles) fetch(kt + 2);

        const float* as = As0 + cur * (BK * BM) + a_off;
        const float* bs = Bs0 + cur * (BK * BN) + b_off;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp)
        {
            float fa[Shape::TM], fbv[Shape::TN];
#pragma unroll
            for (int i = 0; i < Shape::TM; ++i) fa[i] = as[(2 * kp) * BM + i * 32];
#pragma unroll
            for (int j = 0; j < Shape::TN; ++j) fbv[j] = bs[(2 * kp) * BN + j * 32];
#pragma unroll
            for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
                for (int j = 0; j < Shape::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fbv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue.  After the last barrier nobody reads the operand buffers any more: each wave transposes its
    // 32x32 MFMA tiles through a private piece of them (C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2)
    // + 4 * (lane >> 5)) and stores 4 consecutive columns per lane.  Wave-private + in-order LDS queue: no barrier.
    float* const scr = lds + wave * (32 * Shape::EPI_LD);
    const int e_row = lane >> 3, e_c4 = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < Shape::TN; ++j)
    {
        const typename Policy::Store st(prm, batch, n0 + wn * Shape::WTN + j * 32 + e_c4);
        // The fused residual operand (zeros when the layer has none) is requested one 32-row tile AHEAD: tile i + 1's four float4 go out
        // right after tile i's accumulators went to the transpose scratch -- into the registers those accumulators just left, so the
        // request costs no occupancy (requesting every tile up front did: 86 -> 104 VGPRs, 5 -> 4 waves per SIMD) -- and their round trip
        // runs under tile i's LDS reads and stores instead of starting only behind them: one exposed HBM / L2 latency per wave and column
        // piece instead of TM (round 4; round 3 had already moved the four requests of a tile in front of its transpose).
        float4 res[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) res[q] = st.residual4(prm, m0 + wm * Shape::WTM + e_row + q * 8);
#pragma unroll
        for (int i = 0; i < Shape::TM; ++i)
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * half) * Shape::EPI_LD + l31] = acc[i][j][r];
            const int mbase = m0 + wm * Shape::WTM + i * 32 + e_row;
            float4 nxt[4];
            if (i + 1 < Shape::TM)
            {
#pragma unroll
                for (int q = 0; q < 4; ++q) nxt[q] = st.residual4(prm, mbase + 32 + q * 8);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
            {
                const float4 v = *reinterpret_cast<const float4*>(&scr[(q * 8 + e_row) * Shape::EPI_LD + e_c4]);
                st.put4b(prm, mbase + q * 8, v, bias_r[i][q], res[q]);
            }
            if (i + 1 < Shape::TM)
            {
#pragma unroll
                for (int q = 0; q < 4; ++q) res[q] = nxt[q];
            }
        }
    }
}

} // namespace fhip
