// EXPERIMENT, NOT IN THE LIBRARY (round 3, measured negative -- DESIGN.md 3.4).  Built, wired into fhip_conv_forward / _chained, parity-green
// (7 geometries against the oracle, M within 2e-6 of input transform + tile GEMM), and SLOWER: VGG-16 conv1_2 b32 1.09 ms with every wave
// doing both jobs (251 VGPRs), 1.44 ms with the producer / consumer split below (one block per CU), against 0.24 + 0.29 = 0.53 ms for
// the input-transform kernel + tile GEMM it replaces; ResNet-50's 56-pixel 3x3 layers 0.17 vs 0.08 ms.  The 8 blocks of a column group
// re-read every patch through L2 (4-byte-aligned dwordx4 rows: ~13 cache lines per wave instruction for 1 KB of data, ~10 GB of L2 -> L1
// line traffic per launch) with one producer wave per SIMD to hide that latency.  Kept as evidence; to build it, include it from
// winograd_f63.hip behind bt8 and call wino_fused_in_gemm_kernel (512 threads, kWinoFusedLdsFloats * 4 bytes of dynamic LDS).
//
// wino_fused_in.h -- Winograd F(6x6,3x3) input transform INSIDE the tile GEMM for layers with K <= 64 output channels (VGG-16 conv1_2,
// ResNet-50's 56-pixel 3x3 layers): V never reaches HBM.  SURVEY.md 8(f) rank 4, scoped the way VERDICT r02 #5 asked: the input
// transform (reference avx/winograd_kernels_F63.cpp:327-513) moves into the B-operand producer of TensorGEMM (:518-692); M stays in HBM
// and the output transform / chained transform are unchanged.
//
// Why only now (round 3): such layers are HBM-bound (C = K = 64: 16 FLOP per byte of V + M), K2 writes 64/36 of the activation and the
// GEMM reads it back -- 1.5 GB of the 2.7 GB conv1_2 moves at batch 32.  Holding all 64 frequency points of a tile panel on chip does not
// fit, but a block does not need all 64: out[i][j] = sum_a sum_b Bt[i][a] d[a][b] Bt[j][b], so ONE row i of the 8 x 8
// frequency grid needs the whole 8 x 8 patch but only ONE output of the column butterfly per patch column (t[b] = sum_a Bt[i][a] d[a][b])
// followed by one 8-point butterfly (bt8 over b) -- 8 of the 64 V values for an eighth of the arithmetic.  So:
//   * a block owns 64 consecutive columns (tiles) x all K <= 64 output channels x the 8 frequency points xi = 8 i + j of ONE row i;
//   * per k-tile of 16 input channels every producer lane loads the 8 x 8 patches of 4 (channel, tile) pairs (two 4-byte-aligned dwordx4 per patch
//     row), reduces them to the 8 V values of row i and writes those into eight [16][64] B tiles in LDS; the eight [16][64] A tiles
//     (U_xi, k contiguous) arrive as plain 16-byte copies; 64 MFMAs per wave follow (8 xi x 8 k-steps, one 32x32 accumulator per xi);
//   * the 8 blocks (i = 0 .. 7) of a column group are consecutive virtual ids on one XCD, so the patches they all read come from HBM once
//     and from that XCD's L2 seven times (L2 -> CU rate measured at 25-30 TB/s, tools/l2_probe.hip);
//   * M leaves through the per-wave LDS transpose as 16-byte row stores, layout [64][K][Pp] as the output transforms expect.
// Row i of the first stage is evaluated with the expression bt8 uses for that output and the second stage IS bt8, the MFMAs accumulate the
// reduction in the same order as the tile GEMM: M is meant to be bit-identical to input transform + tile GEMM (tests compare the two);
// every entry point that runs a qualifying layer takes this route.
#pragma once

#include "gemm_core.h"

namespace fhip
{

struct WinoFusedInParams
{
    const float* in; // [N][C][H][W]
    const float* U;  // [64][Cp][Kp], Kp = 64
    float* M;        // [64][K][Pp]
    int C, K, H, W, PL, PT;
    int TX, T, P, Pp, Cp, Kp;
    int k_tiles; // Cp / 16
    int groups;  // ceil(P / 64)
};

// output ROW of the bt8 butterfly of winograd_f63.hip, expression for expression
template <int ROW>
__device__ __forceinline__ float bt8_row(float r0, float r1, float r2, float r3, float r4, float r5, float r6, float r7)
{
    if (ROW == 0) return (r0 - r6) + 5.25f * (r4 - r2);
    if (ROW == 1) return ((r2 + r6) - 4.25f * r4) + ((r1 + r5) - 4.25f * r3);
    if (ROW == 2) return ((r2 + r6) - 4.25f * r4) - ((r1 + r5) - 4.25f * r3);
    if (ROW == 3) return (r6 + (0.25f * r2 - 1.25f * r4)) + ((0.5f * r1 - 2.5f * r3) + 2.f * r5);
    if (ROW == 4) return (r6 + (0.25f * r2 - 1.25f * r4)) - ((0.5f * r1 - 2.5f * r3) + 2.f * r5);
    if (ROW == 5) return (r6 + 4.f * (r2 - 1.25f * r4)) + ((2.f * r1 - 2.5f * r3) + 0.5f * r5);
    if (ROW == 6) return (r6 + 4.f * (r2 - 1.25f * r4)) - ((2.f * r1 - 2.5f * r3) + 0.5f * r5);
    return (r7 - r1) + 5.25f * (r3 - r5);
}
template <int ROW>
__device__ __forceinline__ void bt8_rows(const float (&d)[8][8], float (&t8)[8])
{
#pragma unroll
    for (int b = 0; b < 8; ++b) t8[b] = bt8_row<ROW>(d[0][b], d[1][b], d[2][b], d[3][b], d[4][b], d[5][b], d[6][b], d[7][b]);
}

typedef float wf4u __attribute__((ext_vector_type(4), aligned(4))); // a 16-byte load from a 4-byte-aligned address (global_load_dwordx4)

// Block = 8 waves, specialised: waves 0-3 CONSUME (the 64 MFMAs per k-tile, 8 accumulators = 128 registers each), waves 4-7 PRODUCE the next
// k-tile's operands meanwhile (patch loads, transform, LDS writes; no accumulators, so two or three patches are in flight per lane).  One
// barrier per k-tile, operands double buffered: 128 KB of LDS, one block per CU -- one producer and one consumer wave on every SIMD.  (The
// first version did both jobs in every wave: 251 registers, the patches of a lane loaded one after the other with the matrix pipe idle in
// between -- 1.09 ms for VGG-16's conv1_2 against 0.53 ms for the two kernels it replaces.)
constexpr int kWinoFusedLdsFloats = 2 * 2 * 8 * 16 * 64; // {B, A} x 2 buffers x 8 frequency points x [16][64]

__global__ __launch_bounds__(512, 2) void wino_fused_in_gemm_kernel(const WinoFusedInParams q)
{
    constexpr int BK = 16, EPI_LD = 36, TILE = 8 * BK * 64; // floats of the eight [16][64] tiles of one operand
    extern __shared__ __attribute__((aligned(16))) float lds[]; // Bs[2][8][16][64] then As[2][8][16][64]
    float* const Bs0 = lds;
    float* const As0 = lds + 2 * TILE;

    const int vid = xcd_remap(blockIdx.x, q.groups * 8);
    const int tg = vid >> 3, fi = vid & 7; // column group, frequency row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int pw = wave & 3, ptid = tid & 255;
    const int l31 = lane & 31, half = lane >> 5, wm = pw >> 1, wn = pw & 1;

    // producer lanes: this lane's tile
    const int p = tg * 64 + lane;
    const bool pvalid = p < q.P;
    const int pc = pvalid ? p : q.P - 1;
    const int n = pc / q.T, t = pc - n * q.T;
    const int ty = t / q.TX, tx = t - ty * q.TX;
    const int y0 = ty * 6 - q.PT, x0 = tx * 6 - q.PL;
    const bool interior = (y0 >= 0) && (x0 >= 0) && (y0 + 8 <= q.H) && (x0 + 8 <= q.W);
    const size_t HW = (size_t)q.H * q.W;
    const float* const img = q.in + (size_t)n * q.C * HW;

    auto produce = [&](int kt, int buf) {
        float* const As = As0 + buf * TILE;
        float* const Bs = Bs0 + buf * TILE;
        // A tiles: U[8 fi + j][kt*16 + k][0 .. 63] -> As[j][k][.]; 2048 float4, 8 per producer lane
        {
            const int k = ptid >> 4, m4 = ptid & 15;
            f32x4 av[8]; // (native vectors: an array of HIP float4 structs ends up in scratch memory)
#pragma unroll
            for (int e = 0; e < 8; ++e) av[e] = *reinterpret_cast<const f32x4*>(q.U + ((size_t)(fi * 8 + e) * q.Cp + kt * BK + k) * q.Kp + m4 * 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) *reinterpret_cast<f32x4*>(As + (e * BK + k) * 64 + m4 * 4) = av[e];
        }
        // B tiles: 4 (channel, tile) pairs per lane: channel kt*16 + pw + 4 r, tile = lane; two patches in flight
#pragma unroll 2
        for (int r = 0; r < 4; ++r)
        {
            const int kk = pw + 4 * r;
            const int c = min(kt * BK + kk, q.C - 1); // channels past C (C % 16 != 0) meet zero rows of U
            const float* ip = img + (size_t)c * HW;
            float d[8][8];
            if (interior)
            {
#pragma unroll
                for (int a = 0; a < 8; ++a)
                {
                    const float* row = ip + (size_t)(y0 + a) * q.W + x0;
                    const wf4u lo = *reinterpret_cast<const wf4u*>(row), hi = *reinterpret_cast<const wf4u*>(row + 4);
                    d[a][0] = lo.x;
                    d[a][1] = lo.y;
                    d[a][2] = lo.z;
                    d[a][3] = lo.w;
                    d[a][4] = hi.x;
                    d[a][5] = hi.y;
                    d[a][6] = hi.z;
                    d[a][7] = hi.w;
                }
            }
            else
            {
#pragma unroll
                for (int a = 0; a < 8; ++a)
                {
                    const int y = y0 + a;
                    const bool yok = (unsigned)y < (unsigned)q.H;
#pragma unroll
                    for (int b = 0; b < 8; ++b)
                    {
                        const int x = x0 + b;
                        const bool ok = yok && ((unsigned)x < (unsigned)q.W);
                        d[a][b] = ok ? ip[(size_t)y * q.W + x] : 0.f;
                    }
                }
            }
            // row fi of B^T d for every patch column -- the SAME expression bt8 evaluates for that output (so V, and with it M, has the
            // bits of the input-transform kernel) --, then the whole 8-point butterfly along the row
            float t8[8];
            switch (fi) // block-uniform
            {
                case 0: bt8_rows<0>(d, t8); break;
                case 1: bt8_rows<1>(d, t8); break;
                case 2: bt8_rows<2>(d, t8); break;
                case 3: bt8_rows<3>(d, t8); break;
                case 4: bt8_rows<4>(d, t8); break;
                case 5: bt8_rows<5>(d, t8); break;
                case 6: bt8_rows<6>(d, t8); break;
                default: bt8_rows<7>(d, t8); break;
            }
            bt8(t8[0], t8[1], t8[2], t8[3], t8[4], t8[5], t8[6], t8[7]);
#pragma unroll
            for (int j = 0; j < 8; ++j) Bs[(j * BK + kk) * 64 + lane] = pvalid ? t8[j] : 0.f;
        }
    };

    // the two roles run separate loops with the same number of barriers, so the producers carry no accumulators and the consumers no patches
    if (producer)
    {
        produce(0, 0);
        __syncthreads();
        for (int kt = 0; kt < q.k_tiles; ++kt)
        {
            if (kt + 1 < q.k_tiles) produce(kt + 1, (kt & 1) ^ 1);
            __syncthreads();
        }
        return;
    }
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    __syncthreads();
    for (int kt = 0; kt < q.k_tiles; ++kt)
    {
        // 8 frequency points x 8 k-steps
        const float* as = As0 + (kt & 1) * TILE + half * 64 + wm * 32 + l31;
        const float* bs = Bs0 + (kt & 1) * TILE + half * 64 + wn * 32 + l31;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp)
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                const float fa = as[(j * BK + 2 * kp) * 64], fb = bs[(j * BK + 2 * kp) * 64];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[j], 0, 0, 0);
            }
        __syncthreads();
    }

    // ---- epilogue (consumer waves): per-wave LDS transpose (gemm_core.h), 16-byte row stores of M[8 fi + j][m][p]
    float* const scr = lds + pw * (32 * EPI_LD);
    const int e_row = lane >> 3, e_c4 = (lane & 7) * 4;
    const int col = tg * 64 + wn * 32 + e_c4;
#pragma unroll
    for (int j = 0; j < 8; ++j)
    {
        float* mbase = q.M + (size_t)(fi * 8 + j) * q.K * q.Pp + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * half) * EPI_LD + l31] = acc[j][r];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
        {
            const float4 v = *reinterpret_cast<const float4*>(&scr[(qd * 8 + e_row) * EPI_LD + e_c4]);
            const int m = wm * 32 + qd * 8 + e_row;
            if (m < q.K) *reinterpret_cast<float4*>(mbase + (size_t)m * q.Pp) = v;
        }
    }
}

} // namespace fhip
