// wino_gemm_out.h -- PROTOTYPE (round 3): Winograd tile GEMM + output transform (+ bias, ReLU, 2x2 max pooling) in one kernel for C = K = 64
// (VGG-16's conv1_2): M never exists.  A block owns 16 consecutive tiles and all 64 output channels; a wave owns 16 channels x 16 tiles =
// one v_mfma_f32_16x16x4_f32 accumulator per frequency point; operands come straight from global memory in MFMA order (V and U laid out
// for it by their producers: no LDS, no barriers); the 36 outputs of each (channel, tile) are accumulated in registers as the 64
// frequency points go by:  R[b] += At[b][j] m(i, j);  after row i:  Y[a][b] += At[a][i] R[b].
//
// MEASURED (tools/g4_bench.hip, synthetic operands, VGG-16 conv1_2 b32, 24.2 GF): 478 us with 16 tiles per wave (232 VGPRs, 2 waves per
// SIMD: 51 TF = 0.32 of the MFMA peak), 540 us with 32 tiles per wave (one wave per SIMD); an eight-slot operand ring at one wave per SIMD
// (slot j re-requested for the next row of frequency points as soon as its MFMAs are issued) 906 us -- hipcc sinks the ring's loads to
// their uses; with stream_gemm.h's inline-asm loads and counted waits (4 slots, scalar bases) 519 us: above 256 registers the 144
// accumulators move to AGPRs and every update pays accvgpr moves.  The tile GEMM + chained transform it would replace take
// 312 + 223 = 535 us, and it would add an input transform for conv2_1 (~60 us): no gain as it stands.  Every wave re-streams its share of
// U (1 MB per block of 16 tiles, 2.9 GB of L2 -> CU traffic per launch) and a frequency point is 512 clk of matrix work against ~2.9 k clk
// of L2 latency: the form needs either the ring or operand sharing through LDS (barriers per frequency point).  Not pursued.
#pragma once

#include "common.h"

namespace fhip
{

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct WinoGemmOutParams
{
    const float* V4; // [blocks][64 xi][4 q][64 lanes][4 e]: V[xi][c = 16 q + 4 e + lane / 16][tile = 16 block + lane % 16]
    const float* U4; // [64 xi][4 w][4 q][64 lanes][4 e]:   U[xi][k = 16 w + lane % 16][c = 16 q + 4 e + lane / 16]
    const float* bias;
    float* out;      // [N][64][OHp][OWp] pooled (or [N][64][OH][OW])
    int TX, T, P;    // tiling of the layer, columns
    int OH, OW;      // the layer's output image
    int relu, has_bias;
};

// At (the NNPACK / reference F(6,3) variant, winograd_f63.hip at6), stored by column: kWgAt[i][a] = At[a][i]; wave-uniform scalar loads
__constant__ float kWgAt[8][8] = {{1, 0, 0, 0, 0, 0, 0, 0},      {1, 1, 1, 1, 1, 1, 0, 0},        {1, -1, 1, -1, 1, -1, 0, 0},  {1, 2, 4, 8, 16, 32, 0, 0},
                                  {1, -2, 4, -8, 16, -32, 0, 0}, {32, 16, 8, 4, 2, 1, 0, 0},      {32, -16, 8, -4, 2, -1, 0, 0}, {0, 0, 0, 0, 0, 1, 0, 0}};

template <bool POOL, int NT>
__global__ __launch_bounds__(256, NT == 1 ? 2 : 1) void wino_gemm_out_kernel(const WinoGemmOutParams p)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const f32x4v* vb = reinterpret_cast<const f32x4v*>(p.V4) + (size_t)blk * 64 * 256 * NT + lane; // [blk][xi][NT][q][lane]
    const f32x4v* ub = reinterpret_cast<const f32x4v*>(p.U4) + (size_t)w * 256 + lane;
    constexpr int E = 4 * NT;
    float Y[E][6][6], R[E][6];
#pragma unroll
    for (int r = 0; r < E; ++r)
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) Y[r][a][b] = 0.f;
    f32x4v av[4], bv[NT][4], an[4], bn[NT][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
        av[q] = ub[q * 64];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t][q] = vb[(t * 4 + q) * 64];
    }
    // (runtime loops: with the 64 frequency points unrolled hipcc hoists the operand loads and spills 1.4 - 2.4 KB per lane)
#pragma unroll 1
    for (int i = 0; i < 8; ++i)
    {
#pragma unroll
        for (int r = 0; r < E; ++r)
#pragma unroll
            for (int b = 0; b < 6; ++b) R[r][b] = 0.f;
#pragma unroll 1
        for (int j = 0; j < 8; ++j)
        {
            const int xi = 8 * i + j, nx = xi < 63 ? xi + 1 : 63;
#pragma unroll
            for (int q = 0; q < 4; ++q)
            {
                an[q] = ub[(size_t)nx * 1024 + q * 64];
#pragma unroll
                for (int t = 0; t < NT; ++t) bn[t][q] = vb[((size_t)nx * NT + t) * 256 + q * 64];
            }
            f32x4v acc0[NT], acc1[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc0[t] = acc1[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; q += 2)
            {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                    {
                        acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][e], bv[t][q][e], acc0[t], 0, 0, 0);
                        acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q + 1][e], bv[t][q + 1][e], acc1[t], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < E; ++r)
            {
                const float m = acc0[r >> 2][r & 3] + acc1[r >> 2][r & 3];
#pragma unroll
                for (int b = 0; b < 6; ++b) R[r][b] = fmaf(kWgAt[j][b], m, R[r][b]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
            {
                av[q] = an[q];
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[t][q] = bn[t][q];
            }
        }
#pragma unroll
        for (int r = 0; r < E; ++r)
#pragma unroll
            for (int a = 0; a < 6; ++a)
            {
                const float c = kWgAt[i][a];
#pragma unroll
                for (int b = 0; b < 6; ++b) Y[r][a][b] = fmaf(c, R[r][b], Y[r][a][b]);
            }
    }
    // epilogue: lane = (tile column lane % 16, channel rows 4 * (lane / 16) + r of this wave's 16)
    const float lo = p.relu ? 0.f : -__builtin_huge_valf();
#pragma unroll
    for (int r = 0; r < E; ++r)
    {
        const int pcol = (blk * NT + (r >> 2)) * 16 + (lane & 15);
        if (pcol >= p.P) continue;
        const int n = pcol / p.T, t = pcol - n * p.T;
        const int ty = t / p.TX, tx = t - ty * p.TX;
        const int k = 16 * w + 4 * (lane >> 4) + (r & 3);
        const float bb = p.has_bias ? p.bias[k] : 0.f;
        if (POOL)
        {
            const int OHp = p.OH >> 1, OWp = p.OW >> 1;
            float* o = p.out + ((size_t)n * 64 + k) * OHp * OWp;
#pragma unroll
            for (int a = 0; a < 3; ++a)
            {
                const int oy = 3 * ty + a;
                if (oy >= OHp) continue;
#pragma unroll
                for (int b = 0; b < 3; ++b)
                {
                    const int ox = 3 * tx + b;
                    if (ox >= OWp) continue;
                    const float v = fmaxf(fmaxf(Y[r][2 * a][2 * b], Y[r][2 * a][2 * b + 1]), fmaxf(Y[r][2 * a + 1][2 * b], Y[r][2 * a + 1][2 * b + 1]));
                    o[(size_t)oy * OWp + ox] = fmaxf(v + bb, lo);
                }
            }
        }
        else
        {
            float* o = p.out + ((size_t)n * 64 + k) * p.OH * p.OW;
#pragma unroll
            for (int a = 0; a < 6; ++a)
            {
                const int oy = 6 * ty + a;
                if (oy >= p.OH) continue;
#pragma unroll
                for (int b = 0; b < 6; ++b)
                {
                    const int ox = 6 * tx + b;
                    if (ox < p.OW) o[(size_t)oy * p.OW + ox] = fmaxf(Y[r][a][b] + bb, lo);
                }
            }
        }
    }
}

} // namespace fhip
