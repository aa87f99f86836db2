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
//     global->LDS copy is a straight 16-byte-per-lane row copy.  fp32 MFMA is 1/16 the bf16 rate, so
//     LDS bandwidth is nowhere near the limit (16 B/clk/CU used of 128); what matters is keeping the
//     matrix pipe issued back to back: 2x2 independent 32x32 accumulators per wave, register-prefetched
//     global loads one k-tile ahead, LDS double buffer with ONE barrier per k-tile, >= 2 blocks per CU.
//   * 1-D grid, XCD-aware remap: consecutive virtual block ids run on one XCD and walk m-tiles fastest,
//     so the blocks that share a B panel / an A panel hit the same private L2.
#pragma once
// FROZEN copy of the round-1 first-version main loop (one tile per block), kept only for A/B runs in tools/gemm_bench.hip.

#include "common.h"

namespace fhip
{

template <int BM_, int BN_, int BK_, int WAVES_M_, int WAVES_N_>
struct GemmShapeV0
{
    static constexpr int BM = BM_, BN = BN_, BK = BK_;
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
    static constexpr int THREADS = 64 * WAVES_M * WAVES_N;
    static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N; // per-wave output tile
    static constexpr int TM = WTM / 32, TN = WTN / 32;           // 32x32 MFMA tiles per wave
    static constexpr int A_F4_PER_ROW = BM / 4, B_F4_PER_ROW = BN / 4;
    static constexpr int A_ROWS_PER_PASS = THREADS / A_F4_PER_ROW, B_ROWS_PER_PASS = THREADS / B_F4_PER_ROW;
    static constexpr int A_PASSES = BK / A_ROWS_PER_PASS, B_PASSES = BK / B_ROWS_PER_PASS;
    static constexpr int LDS_FLOATS = 2 * BK * (BM + BN);
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA");
    static_assert(THREADS % A_F4_PER_ROW == 0 && THREADS % B_F4_PER_ROW == 0, "loader mapping");
    static_assert(A_PASSES >= 1 && B_PASSES >= 1 && BK % A_ROWS_PER_PASS == 0 && BK % B_ROWS_PER_PASS == 0, "BK too small");
    static_assert(BK % 2 == 0, "MFMA k-depth is 2");
};

// Policy concept:
//   struct Params { int batches, m_tiles, n_tiles, k_tiles; ... };
//   struct ALoad { ALoad(const Params&, int batch, int m4); float4 load(const Params&, int krow) const; };
//   struct BLoad { BLoad(const Params&, int batch, int n4); float4 load(const Params&, int krow) const; };
//       (m4 / n4 = first of the 4 consecutive rows / columns this thread always fetches)
//   struct Store { Store(const Params&, int batch, int n); void put(const Params&, int m, float v) const; };
// ABLATE (measurement builds only, tools/gemm_bench.hip; the product always uses 0):
//   bit 0: no global fetch inside the k loop (MFMA + LDS only), bit 1: no accumulator store.
template <class Shape, class Policy, int ABLATE = 0, int MIN_WAVES = 2>
__global__ __launch_bounds__(Shape::THREADS, MIN_WAVES) void gemm_mfma_kernel_v0(const typename Policy::Params prm)
{
    constexpr int BM = Shape::BM, BN = Shape::BN, BK = Shape::BK;
    __shared__ __attribute__((aligned(16))) float lds[Shape::LDS_FLOATS];
    // As[buf] = lds + buf * BK*BM ; Bs[buf] = lds + 2*BK*BM + buf * BK*BN  (plain arithmetic: a runtime-indexed
    // pointer array would be demoted to scratch)
    float* const As0 = lds;
    float* const Bs0 = lds + 2 * BK * BM;

    const int nwg = prm.batches * prm.m_tiles * prm.n_tiles;
    int vid = xcd_remap(blockIdx.x, nwg);
    const int mt = vid % prm.m_tiles;
    vid /= prm.m_tiles;
    const int nt = vid % prm.n_tiles;
    const int batch = vid / prm.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

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
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i) pa[i] = aload.load(prm, kt * BK + a_r + i * Shape::A_ROWS_PER_PASS);
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i) { unsigned okm; pb[i] = bload.load(prm, kt * BK + b_r + i * Shape::B_ROWS_PER_PASS, okm); }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i)
            *reinterpret_cast<float4*>(&As0[buf * (BK * BM) + (a_r + i * Shape::A_ROWS_PER_PASS) * BM + a_c4 * 4]) = pa[i];
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i)
            *reinterpret_cast<float4*>(&Bs0[buf * (BK * BN) + (b_r + i * Shape::B_ROWS_PER_PASS) * BN + b_c4 * 4]) = pb[i];
    };

    f32x16 acc[Shape::TM][Shape::TN];
#pragma unroll
    for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
        for (int j = 0; j < Shape::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    fetch(0);
    stash(0);
    __syncthreads();

    const int a_off = half * BM + wm * Shape::WTM + l31;
    const int b_off = half * BN + wn * Shape::WTN + l31;
    int cur = 0;
    for (int kt = 0; kt < prm.k_tiles; ++kt)
    {
        const bool more = (kt + 1 < prm.k_tiles) && !(ABLATE & 1);
        if (more) fetch(kt + 1); // global loads for the next k-tile fly under this tile's MFMAs
        const float* as = As0 + cur * (BK * BM) + a_off;
        const float* bs = Bs0 + cur * (BK * BN) + b_off;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2)
        {
            float a[Shape::TM], b[Shape::TN];
#pragma unroll
            for (int i = 0; i < Shape::TM; ++i) a[i] = as[kk * BM + i * 32];
#pragma unroll
            for (int j = 0; j < Shape::TN; ++j) b[j] = bs[kk * BN + j * 32];
#pragma unroll
            for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
                for (int j = 0; j < Shape::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#pragma unroll
    for (int j = 0; j < Shape::TN; ++j)
    {
        const int n = n0 + wn * Shape::WTN + j * 32 + l31;
        float* const stb = prm.M + (size_t)batch * prm.K * prm.Pp + n;
#pragma unroll
        for (int i = 0; i < Shape::TM; ++i)
        {
            const int mbase = m0 + wm * Shape::WTM + i * 32 + 4 * half;
            if (ABLATE & 2)
            {
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[i][j][r]));
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { const int m_ = mbase + (r & 3) + 8 * (r >> 2); if (m_ < prm.K) stb[(size_t)m_ * prm.Pp] = acc[i][j][r]; }
        }
    }
}

} // namespace fhip
