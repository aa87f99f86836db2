// conv_stream3.h -- the first layer of the benchmark nets (3 input channels, 3x3 kernel, stride 1 or 2, 32 or 64 output channels) as a
// register-streamed implicit GEMM: no LDS, no barriers, no im2col.
//
// Same contraction as IM2COL_Forward + packed_sgemm_activation of the reference (avx/booster.cpp:83-102, avx/generic_kernels.cpp:50-85,
// avx/sgemm.cpp:381-432) and as conv_smallc_kernel, laid out like stream_gemm.h: a lane owns 4 consecutive output pixels of one row, a
// wave 32 such groups, and the B operand of reduction row (c, u, w) is the float4 in[c][oy*S + u - pad][ox0*S + w - pad ...] -- one
// (unaligned) 16-byte load per reduction-row pair and lane, two when the stride is 2 (the even elements of 8) -- whose components feed
// the MFMAs of the four interleaved pixel sets.  Padding is a mask on the loaded components (a row out of range masks all four); the
// loads themselves stay inside the tensor except for the handful of lanes at its very first and last elements, whose wave takes a
// scalar path.  The 27 x K weights sit in registers (14 A operands per 32 output channels, read from the k-major packed matrix of
// igemm_init), the result leaves as dwordx4 stores of 4 consecutive pixels with bias + ReLU: the kernel is bound by its output
// writes (VGG-16 conv1_1 b32: 411 MB), which the LDS-staged form (conv_smallc_kernel: patch -> LDS -> gather -> MFMA -> LDS
// transpose -> store, 2 barriers per 128 pixels) reached only 3.2 TB/s of.
#pragma once
#include <type_traits>

#include "common.h"

namespace fhip
{
typedef float s3_f32x16 __attribute__((ext_vector_type(16)));

struct Stream3Params
{
    const float* in;   // [N][3][H][W]
    const float* Wt;   // packed k-major [32 rows][Kp]: Wt[r * Kp + k] = W[k][r], rows >= 27 zero (igemm_init)
    const float* bias; // [K] (read when BIAS)
    float* out;        // [N][K][OH][OW]
    int H, W, OH, OW, K, Kp, PL, PT;
    int groups_per_row;     // OW / 4
    long long groups;       // N * OH * OW / 4
    long long in_floats;    // N * 3 * H * W
};

// S = stride (1, 2); MGROUPS = K / 32 (1, 2): a wave owns ONE 32-channel group of one 128-pixel tile (64 accumulator registers: with two
// groups per wave the kernel needs 270 registers and runs one wave per SIMD, measured 20 % slower than the LDS-staged kernel);
// the waves of the two groups of a tile sit in the same block and share the input through L1
template <int S, int MGROUPS, bool BIAS, bool RELU>
__global__ __launch_bounds__(256) void conv_stream3_kernel(const Stream3Params q)
{
    constexpr int J = 14; // reduction rows 0 .. 26 (+ one zero row) in pairs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    constexpr int MG = 1;
    // 32-bit index arithmetic throughout (the host checks groups < 2^31 and in-image offsets < 2^28): 64-bit divisions cost a wave
    // more than its 56 MFMAs
    const unsigned wid = blockIdx.x * 4u + wave;
    const unsigned tile = wid / MGROUPS;
    const int mg = (int)(wid - tile * MGROUPS);
    const unsigned g = tile * 32u + l31; // this lane's group of 4 output pixels
    if (tile * 32u >= (unsigned)q.groups) return; // whole wave beyond the tensor
    const bool ok = g < (unsigned)q.groups;
    const unsigned gc = ok ? g : 0u;
    const unsigned rowid = gc / (unsigned)q.groups_per_row;
    const int gx = (int)(gc - rowid * (unsigned)q.groups_per_row);
    const unsigned nn = rowid / (unsigned)q.OH;
    const int oy = (int)(rowid - nn * (unsigned)q.OH), n = (int)nn;
    const int ox0 = gx * 4;

    // weights: A operand of step j for m-group m = W[32 m + l31][2 j + half]
    float aw[J][MG];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int m = 0; m < MG; ++m) aw[j][m] = q.Wt[(size_t)(2 * j + half) * q.Kp + 32 * (mg + m) + l31];

    // geometry of the three input rows and three column offsets this lane reads
    const size_t img = (size_t)n * 3 * q.H * q.W;
    int rowoff[3];  // (clamped iy) * W
    bool rowok[3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
    {
        const int iy = oy * S + u - q.PT;
        rowok[u] = (unsigned)iy < (unsigned)q.H;
        rowoff[u] = min(max(iy, 0), q.H - 1) * q.W;
    }
    const int ixb = ox0 * S - q.PL; // column of tap w = 0 of pixel 0
    unsigned colmask[3];            // bit t: pixel t's tap w is inside the row
#pragma unroll
    for (int w = 0; w < 3; ++w)
    {
        colmask[w] = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) colmask[w] |= ((unsigned)(ixb + w + t * S) < (unsigned)q.W) ? (1u << t) : 0u;
    }
    // may every vector load of this lane stay inside the tensor?  (only the first / last few elements of the whole input can fail)
    const long long lo = (long long)img + rowoff[0] + ixb, hi = (long long)img + 2LL * q.H * q.W + rowoff[2] + ixb + 2 + (S == 1 ? 4 : 8);
    const bool fast = !__builtin_amdgcn_ballot_w64(lo < 0 || hi > q.in_floats);

    s3_f32x16 acc[MG][4];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

    const float* ip = q.in + img;
    // The two load forms are two separate loops: a load under a branch makes hipcc wait for it on the spot (vmcnt(0) at the join),
    // which would serialise the 14 steps on 14 memory round trips.
    auto steps = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
        for (int j = 0; j < J; ++j)
        {
            // reduction row of this half-wave: r = 2 j + half = (c, u, w); r = 27 is the zero row (any address, zero weights)
            const int r0 = 2 * j, r1 = min(2 * j + 1, 26);
            const int c = half ? r1 / 9 : r0 / 9, u = half ? (r1 % 9) / 3 : (r0 % 9) / 3, w = half ? r1 % 3 : r0 % 3;
            const int ro = u == 0 ? rowoff[0] : (u == 1 ? rowoff[1] : rowoff[2]);
            const bool rk = u == 0 ? rowok[0] : (u == 1 ? rowok[1] : rowok[2]);
            const unsigned cm = rk ? (w == 0 ? colmask[0] : (w == 1 ? colmask[1] : colmask[2])) : 0u;
            const int off = c * q.H * q.W + ro + ixb + w;
            float b0, b1, b2, b3;
            if (FAST)
            {
                const float4 v0 = *reinterpret_cast<const float4*>(ip + off);
                if (S == 1)
                {
                    b0 = v0.x;
                    b1 = v0.y;
                    b2 = v0.z;
                    b3 = v0.w;
                }
                else
                {
                    const float4 v1 = *reinterpret_cast<const float4*>(ip + off + 4);
                    b0 = v0.x;
                    b1 = v0.z;
                    b2 = v1.x;
                    b3 = v1.z;
                }
            }
            else
            {
                const long long first = -(long long)img, last = q.in_floats - 1 - (long long)img;
                b0 = ip[min(max((long long)off, first), last)];
                b1 = ip[min(max((long long)off + S, first), last)];
                b2 = ip[min(max((long long)off + 2 * S, first), last)];
                b3 = ip[min(max((long long)off + 3 * S, first), last)];
            }
            b0 = (cm & 1u) ? b0 : 0.f;
            b1 = (cm & 2u) ? b1 : 0.f;
            b2 = (cm & 4u) ? b2 : 0.f;
            b3 = (cm & 8u) ? b3 : 0.f;
#pragma unroll
            for (int m = 0; m < MG; ++m)
            {
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[j][m], b0, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[j][m], b1, acc[m][1], 0, 0, 0);
                acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[j][m], b2, acc[m][2], 0, 0, 0);
                acc[m][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[j][m], b3, acc[m][3], 0, 0, 0);
            }
        }
    };
    if (fast)
        steps(std::true_type());
    else
        steps(std::false_type());
    if (!ok) return;
    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const size_t plane = (size_t)q.OH * q.OW;
#pragma unroll
    for (int m = 0; m < MG; ++m)
    {
        float* op = q.out + ((size_t)n * q.K + 32 * (mg + m) + 4 * half) * plane + (size_t)oy * q.OW + ox0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
            const int row = (r & 3) + 8 * (r >> 2);
            const float bs = BIAS ? q.bias[32 * (mg + m) + 4 * half + row] : 0.f;
            float4 v = make_float4(acc[m][0][r] + bs, acc[m][1][r] + bs, acc[m][2][r] + bs, acc[m][3][r] + bs);
            if (RELU)
            {
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
            }
            *reinterpret_cast<float4*>(op + (size_t)row * plane) = v;
        }
    }
}
} // namespace fhip
