// winograd_f63.hip -- Winograd F(6x6,3x3) for gfx950: filter / input / output transforms and the 64-way
// batched tile GEMM.  Replaces the reference's non-fused AVX pipeline
//   transformKernel_F6x6_3x3          src/booster/avx/winograd_kernels_F63.cpp:222-271
//   pad_input + winogradInputFrameTransformSeq   avx/generic_kernels.cpp:31-48, winograd_kernels_F63.cpp:327-513
//   TensorGEMM                        winograd_kernels_F63.cpp:518-692
//   winogradOutputTransform<relu,bias> winograd_kernels_F63.cpp:1088-1269
// with the same transform matrices (G :224-234, B^T :297-324, A^T :1039-1046) and the same edge rules
// (zero-filled edge tiles :378-414, clipped stores :1200-1232), but none of its layouts:
//
//   U[xi][Cp][Kp]   filters, xi = 8*i + j frequency point, k (output channel) contiguous, zero padded
//   V[xi][C][Pp]    transformed input, column p = n*T + ty*TX + tx (tile of image n) contiguous
//   M[xi][K][Pp]    tile-GEMM output
//
// i.e. every xi is a plain k-major GEMM operand pair for gemm_core.h, every transform kernel maps one
// tile to one lane with lanes running along p, so all 64 V stores / M loads of a wave are 256-byte
// coalesced rows, and the padding (pad_input) is folded into the input transform's bounds checks
// instead of a padded copy of the input.
#include <algorithm>

#include "gemm_core.h"
#include "wino_gemm_policy.h"
#include "wino_gemm_glds.h"

namespace fhip
{

// ---------------------------------------------------------------------------------------------------
// B^T d (8 -> 8) and A^T m (8 -> 6) butterflies (the NNPACK/ncnn F(6,3) variant the reference uses).
__device__ __forceinline__ void bt8(float& r0, float& r1, float& r2, float& r3, float& r4, float& r5, float& r6, float& r7)
{
    const float o0 = (r0 - r6) + 5.25f * (r4 - r2);
    const float o7 = (r7 - r1) + 5.25f * (r3 - r5);
    const float t1 = (r2 + r6) - 4.25f * r4;
    const float t2 = (r1 + r5) - 4.25f * r3;
    const float p1 = r6 + (0.25f * r2 - 1.25f * r4);
    const float p2 = (0.5f * r1 - 2.5f * r3) + 2.f * r5;
    const float q1 = r6 + 4.f * (r2 - 1.25f * r4);
    const float q2 = (2.f * r1 - 2.5f * r3) + 0.5f * r5;
    r0 = o0;
    r1 = t1 + t2;
    r2 = t1 - t2;
    r3 = p1 + p2;
    r4 = p1 - p2;
    r5 = q1 + q2;
    r6 = q1 - q2;
    r7 = o7;
}

__device__ __forceinline__ void at6(float m0, float m1, float m2, float m3, float m4, float m5, float m6, float m7,
                                    float& s0, float& s1, float& s2, float& s3, float& s4, float& s5)
{
    const float a12 = m1 + m2, d12 = m1 - m2;
    const float a34 = m3 + m4, d34 = m3 - m4;
    const float a56 = m5 + m6, d56 = m5 - m6;
    s0 = (m0 + a12) + (a34 + 32.f * a56);
    s1 = (d12 + 2.f * d34) + 16.f * d56;
    s2 = (a12 + 4.f * a34) + 8.f * a56;
    s3 = (d12 + 8.f * d34) + 4.f * d56;
    s4 = (a12 + 16.f * a34) + 2.f * a56;
    s5 = ((d12 + 32.f * d34) + d56) + m7;
}

// ---------------------------------------------------------------------------------------------------
// K1: U = G g G^T, one (k, c) filter per lane, lanes along k so the 64 stores per lane are coalesced rows
// of U[xi][c][.].  One-time work (ConvBooster::Init).
__global__ __launch_bounds__(256) void wino_filter_transform_kernel(float* __restrict__ U, const float* __restrict__ w,
                                                                   int C, int K, int Cp, int Kp)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (k >= K) return;
    const float* g = w + ((size_t)k * C + c) * 9;
    float gg[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) gg[i][j] = g[i * 3 + j];
    const float G[8][3] = {{1.0f, 0.0f, 0.0f},
                           {-2.0f / 9, -2.0f / 9, -2.0f / 9},
                           {-2.0f / 9, 2.0f / 9, -2.0f / 9},
                           {1.0f / 90, 1.0f / 45, 2.0f / 45},
                           {1.0f / 90, -1.0f / 45, 2.0f / 45},
                           {1.0f / 45, 1.0f / 90, 1.0f / 180},
                           {1.0f / 45, -1.0f / 90, 1.0f / 180},
                           {0.0f, 0.0f, 1.0f}};
    float mid[8][3];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) mid[i][j] = G[i][0] * gg[0][j] + G[i][1] * gg[1][j] + G[i][2] * gg[2][j];
    const size_t xi_stride = (size_t)Cp * Kp;
    float* up = U + (size_t)c * Kp + k;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            up[(size_t)(i * 8 + j) * xi_stride] = mid[i][0] * G[j][0] + mid[i][1] * G[j][1] + mid[i][2] * G[j][2];
}

// ---------------------------------------------------------------------------------------------------
// K2: V = B^T d B on 8x8 tiles at stride 6.  One tile per lane; padding and edge tiles are bounds checks.
struct WinoXformParams
{
    int C, K, H, W, OH, OW, PL, PT;
    int TX, T;   // tiles per row, tiles per image
    int P, Pp;   // columns, padded columns
    int N;
    WinoLayout Lv, Lm; // where V (rows = C) and M (rows = K) live (wino_layout.h)
};

__global__ __launch_bounds__(256) void wino_input_transform_kernel(float* __restrict__ V, const float* __restrict__ in,
                                                                  const WinoXformParams q)
{
    // flattened (channel, column) index, column fastest: no partially filled blocks when P is small (14x14 images
    // at batch 32 have P = 288: a per-channel grid would run its second block of 256 lanes with 32 of them)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)q.C * q.P) return;
    const int c = (int)(idx / q.P);
    const int p = (int)(idx - (long long)c * q.P);
    const int n = p / q.T, t = p - n * q.T;
    const int ty = t / q.TX, tx = t - ty * q.TX;
    const int y0 = ty * 6 - q.PT, x0 = tx * 6 - q.PL;
    const float* ip = in + ((size_t)n * q.C + c) * q.H * q.W;

    float d[8][8];
    const bool interior = (y0 >= 0) && (x0 >= 0) && (y0 + 8 <= q.H) && (x0 + 8 <= q.W);
    if (interior)
    {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) d[i][j] = ip[(size_t)(y0 + i) * q.W + x0 + j];
    }
    else
    {
#pragma unroll
        for (int i = 0; i < 8; ++i)
        {
            const int y = y0 + i;
            const bool yok = (unsigned)y < (unsigned)q.H;
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                const int x = x0 + j;
                const bool ok = yok && ((unsigned)x < (unsigned)q.W);
                d[i][j] = ok ? ip[(size_t)y * q.W + x] : 0.f;
            }
        }
    }
    // B^T d : along the row index, for every column
#pragma unroll
    for (int j = 0; j < 8; ++j) bt8(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], d[6][j], d[7][j]);
    // (.) B : along the column index, for every row
#pragma unroll
    for (int i = 0; i < 8; ++i) bt8(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5], d[i][6], d[i][7]);

    const size_t xi_stride = q.Lv.xis;
    float* vp = V + (size_t)c * q.Lv.bp + q.Lv.col(p);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) st_scratch(vp + (size_t)(i * 8 + j) * xi_stride, d[i][j]);
}

} // namespace fhip
#include "wino_first.h" // K2 with the net's first (<= 4-channel) convolution computed in place of the loads
namespace fhip
{

// (Round 1 measured a one-shot LDS-staged K2 -- row loads into LDS, a barrier, windows from LDS -- slower than the direct form above on
// VGG-16's planes (1.37 vs 0.87 ms per step).  Round 4's wino_input_staged_kernel below is the form that wins, on ResNet-50's 56 / 28 / 14-px
// planes: whole planes as 16-byte vectors, persistent blocks, the next unit's vectors in flight while this one's windows are transformed.)
struct WinoStaged
{
    int UB;    // units per block
    int LDW;   // LDS row pitch (floats): >= R*TX, multiple of 4
    int units; // N * TY
    int TY;
    int R;     // staged rows per unit: 6, or 3 when the 2x2 max pooling is fused
};

// ---------------------------------------------------------------------------------------------------
// K4: Y = A^T m A, + bias, ReLU, clipped 6x6 store.
template <bool HAS_BIAS, bool RELU>
__global__ __launch_bounds__(256) void wino_output_transform_kernel(float* __restrict__ out, const float* __restrict__ M,
                                                                   const float* __restrict__ bias, const WinoXformParams q)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (p >= q.P) return;
    const int n = p / q.T, t = p - n * q.T;
    const int ty = t / q.TX, tx = t - ty * q.TX;

    const size_t xi_stride = q.Lm.xis;
    const float* mp = M + (size_t)k * q.Lm.bp + q.Lm.col(p);
    float m[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) m[i][j] = ld_scratch(mp + (size_t)(i * 8 + j) * xi_stride);

    float tmp[6][8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        at6(m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j], m[6][j], m[7][j], tmp[0][j], tmp[1][j], tmp[2][j],
            tmp[3][j], tmp[4][j], tmp[5][j]);
    const float b = HAS_BIAS ? bias[k] : 0.f;
    const int oy0 = ty * 6, ox0 = tx * 6;
    float* op = out + (((size_t)n * q.K + k) * q.OH + oy0) * q.OW + ox0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
    {
        float y[6];
        at6(tmp[a][0], tmp[a][1], tmp[a][2], tmp[a][3], tmp[a][4], tmp[a][5], tmp[a][6], tmp[a][7], y[0], y[1], y[2], y[3],
            y[4], y[5]);
        if (oy0 + a < q.OH)
        {
#pragma unroll
            for (int bb = 0; bb < 6; ++bb)
                if (ox0 + bb < q.OW)
                {
                    float v = y[bb] + b;
                    if (RELU) v = fmaxf(v, 0.f);
                    op[(size_t)a * q.OW + bb] = v;
                }
        }
    }
}

// K4, LDS-staged form (the one normally used).  A block owns UB consecutive TILE ROWS ("units": unit u = image n,
// tile row ty; its tiles are the TX consecutive columns p = u*TX .. u*TX+TX-1) of one output channel k, one tile per
// lane -- so the 64 M loads of a wave are still 256-byte coalesced rows.  The 6x6 results go to LDS laid out as the
// output image rows they are ([unit][6][6*TX]); a unit's 6 x OW output floats are ONE contiguous run of the
// NCHW tensor, so the block then copies LDS -> global with 16-byte coalesced stores.  (The direct form above
// writes 6-float pieces at a 24-byte lane stride: 36 store instructions per lane, each touching 12 cache lines;
// measured 3.3 TB/s effective on VGG conv1_2 against ~5 TB/s for the coalesced input transform.)

// POOL: the 6x6 tile is reduced by a 2x2 / stride-2 max in registers before it is staged, so the block writes 3 x OW/2
// pooled rows per unit and the full-resolution output never exists (OH, OW even: pooled cells never straddle the edge).
template <bool HAS_BIAS, bool RELU, bool POOL>
__global__ __launch_bounds__(256) void wino_output_transform_staged_kernel(float* __restrict__ out, const float* __restrict__ M,
                                                                          const float* __restrict__ bias,
                                                                          const WinoXformParams q, const WinoStaged g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const tile = smem;                                             // [UB][R][LDW], R = 6 (3 when pooled)
    long long* const ubase = reinterpret_cast<long long*>(smem + (size_t)g.UB * g.R * g.LDW); // [UB] output offset of the unit's first row
    int* const urows = reinterpret_cast<int*>(ubase + g.UB);              // [UB] valid rows (0 = unit beyond the tensor)

    const int tid = threadIdx.x;
    const int k = blockIdx.y;
    const int u0 = blockIdx.x * g.UB;
    if (tid < g.UB)
    {
        const int u = u0 + tid;
        int rows = 0;
        long long base = 0;
        if (u < g.units)
        {
            const int n = u / g.TY, ty = u - n * g.TY;
            rows = min(6, q.OH - 6 * ty);
            base = (((long long)n * q.K + k) * q.OH + 6 * ty) * q.OW;
            if (POOL)
            {
                rows >>= 1;
                base = (((long long)n * q.K + k) * (q.OH >> 1) + 3 * ty) * (q.OW >> 1);
            }
        }
        ubase[tid] = base;
        urows[tid] = rows;
    }

    const int unit_l = tid / q.TX, tx = tid - unit_l * q.TX;
    const bool active = unit_l < g.UB && (u0 + unit_l) < g.units;
    if (active)
    {
        const int p = (u0 + unit_l) * q.TX + tx;
        const size_t xi_stride = q.Lm.xis;
        const float* mp = M + (size_t)k * q.Lm.bp + q.Lm.col(p);
        float m[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) m[i][j] = ld_scratch(mp + (size_t)(i * 8 + j) * xi_stride);
        float tmp[6][8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            at6(m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j], m[6][j], m[7][j], tmp[0][j], tmp[1][j], tmp[2][j],
                tmp[3][j], tmp[4][j], tmp[5][j]);
        const float b = HAS_BIAS ? bias[k] : 0.f;
        float* tp = tile + ((size_t)unit_l * g.R) * g.LDW + (POOL ? 3 : 6) * tx;
        float prev0 = 0.f, prev1 = 0.f, prev2 = 0.f;
#pragma unroll
        for (int a = 0; a < 6; ++a)
        {
            float y[6];
            at6(tmp[a][0], tmp[a][1], tmp[a][2], tmp[a][3], tmp[a][4], tmp[a][5], tmp[a][6], tmp[a][7], y[0], y[1], y[2], y[3],
                y[4], y[5]);
#pragma unroll
            for (int bb = 0; bb < 6; ++bb)
            {
                float v = y[bb] + b;
                if (RELU) v = fmaxf(v, 0.f);
                y[bb] = v;
            }
            if (POOL)
            {
                const float h0 = fmaxf(y[0], y[1]), h1 = fmaxf(y[2], y[3]), h2 = fmaxf(y[4], y[5]);
                if ((a & 1) == 0)
                {
                    prev0 = h0;
                    prev1 = h1;
                    prev2 = h2;
                }
                else
                {
                    float* t = tp + (size_t)(a >> 1) * g.LDW;
                    t[0] = fmaxf(prev0, h0);
                    t[1] = fmaxf(prev1, h1);
                    t[2] = fmaxf(prev2, h2);
                }
                continue;
            }
            // 6*tx floats = 24*tx bytes: 8-byte aligned
            float2* t2 = reinterpret_cast<float2*>(tp + (size_t)a * g.LDW);
            t2[0] = make_float2(y[0], y[1]);
            t2[1] = make_float2(y[2], y[3]);
            t2[2] = make_float2(y[4], y[5]);
        }
    }
    __syncthreads();

    // LDS -> global: every valid output row of the block, OW floats each, consecutive lanes on consecutive addresses
    const int R = POOL ? 3 : 6, OWo = POOL ? q.OW >> 1 : q.OW;
    if ((OWo & 3) == 0)
    {
        const int w4 = OWo >> 2;
        const int total = g.UB * R * w4;
        for (int idx = tid; idx < total; idx += 256)
        {
            const int row = idx / w4, x4 = idx - row * w4;
            const int ul = row / R, a = row - ul * R;
            if (a < urows[ul])
            {
                const float4 v = *reinterpret_cast<const float4*>(tile + ((size_t)ul * R + a) * g.LDW + 4 * x4);
                stg4_act<2>(out + ubase[ul] + (long long)a * OWo + 4 * x4, v);
            }
        }
    }
    else
    {
        const int total = g.UB * R * OWo;
        for (int idx = tid; idx < total; idx += 256)
        {
            const int row = idx / OWo, x = idx - row * OWo;
            const int ul = row / R, a = row - ul * R;
            if (a < urows[ul]) out[ubase[ul] + (long long)a * OWo + x] = tile[((size_t)ul * R + a) * g.LDW + x];
        }
    }
}

// K4, LDS-staged and PERSISTENT (round 5): the staged kernel above as a block that walks work items -- item = (output channel k, group of
// UB tile rows), k-major so that a block's consecutive items are consecutive column runs of M -- with the NEXT item's 64 M values per lane
// requested right after the barrier, before this item's rows leave the LDS: loads and stores of one block overlap, the form that took the
// chained transform from 0.49 to 0.55 of the HBM rate (wino_chain_kernel).  Same butterflies on the same values in the same order: the output
// is bit-identical to the one-shot kernel's.  Used where the one-shot grid is more than one round of resident blocks (ResNet-50 b64's 56-
// and 28-pixel 3x3 layers); a grid that is resident at once keeps the one-shot kernel.
template <bool HAS_BIAS, bool RELU, bool POOL>
__global__ __launch_bounds__(256, 3) void wino_output_transform_persist_kernel(float* __restrict__ out, const float* __restrict__ M,
                                                                              const float* __restrict__ bias, const WinoXformParams q,
                                                                              const WinoStaged g, const int xblocks, const int items)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const tile = smem;
    long long* const ubase = reinterpret_cast<long long*>(smem + (size_t)g.UB * g.R * g.LDW);
    int* const urows = reinterpret_cast<int*>(ubase + g.UB);

    const int tid = threadIdx.x;
    const int unit_l = tid / q.TX, tx = tid - unit_l * q.TX;
    const bool lane_in = unit_l < g.UB;
    // the XCD blockIdx % 8 owns a contiguous eighth of the items, its blocks take neighbouring items at the same time
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, xcd_blocks = (gridDim.x + 7 - xcd) >> 3;
    const int i_lo = (int)((long long)items * xcd / 8), i_hi = (int)((long long)items * (xcd + 1) / 8);
    const size_t xi_stride = q.Lm.xis;

    float m[8][8];
    auto fetch = [&](int item) {
        // clamped and unconditional (a load under a branch is waited for on the spot)
        const int k = item / xblocks, bx = item - k * xblocks;
        const int u = min(bx * g.UB + (lane_in ? unit_l : 0), g.units - 1);
        const float* mp = M + (size_t)k * q.Lm.bp + q.Lm.col(u * q.TX + tx);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) m[i][jj] = ld_scratch(mp + (size_t)(i * 8 + jj) * xi_stride);
    };
    int item = i_lo + j;
    if (item < i_hi) fetch(item);
    for (; item < i_hi; item += xcd_blocks)
    {
        const int k = item / xblocks, u0 = (item - k * xblocks) * g.UB;
        if (tid < g.UB)
        {
            const int u = u0 + tid;
            int rows = 0;
            long long base = 0;
            if (u < g.units)
            {
                const int n = u / g.TY, ty = u - n * g.TY;
                rows = min(6, q.OH - 6 * ty);
                base = (((long long)n * q.K + k) * q.OH + 6 * ty) * q.OW;
                if (POOL)
                {
                    rows >>= 1;
                    base = (((long long)n * q.K + k) * (q.OH >> 1) + 3 * ty) * (q.OW >> 1);
                }
            }
            ubase[tid] = base;
            urows[tid] = rows;
        }
        if (lane_in && (u0 + unit_l) < g.units)
        {
            float tmp[6][8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                at6(m[0][jj], m[1][jj], m[2][jj], m[3][jj], m[4][jj], m[5][jj], m[6][jj], m[7][jj], tmp[0][jj], tmp[1][jj], tmp[2][jj], tmp[3][jj],
                    tmp[4][jj], tmp[5][jj]);
            const float b = HAS_BIAS ? bias[k] : 0.f;
            float* tp = tile + ((size_t)unit_l * g.R) * g.LDW + (POOL ? 3 : 6) * tx;
            float prev0 = 0.f, prev1 = 0.f, prev2 = 0.f;
#pragma unroll
            for (int a = 0; a < 6; ++a)
            {
                float y[6];
                at6(tmp[a][0], tmp[a][1], tmp[a][2], tmp[a][3], tmp[a][4], tmp[a][5], tmp[a][6], tmp[a][7], y[0], y[1], y[2], y[3], y[4], y[5]);
#pragma unroll
                for (int bb = 0; bb < 6; ++bb)
                {
                    float v = y[bb] + b;
                    if (RELU) v = fmaxf(v, 0.f);
                    y[bb] = v;
                }
                if (POOL)
                {
                    const float h0 = fmaxf(y[0], y[1]), h1 = fmaxf(y[2], y[3]), h2 = fmaxf(y[4], y[5]);
                    if ((a & 1) == 0)
                    {
                        prev0 = h0;
                        prev1 = h1;
                        prev2 = h2;
                    }
                    else
                    {
                        float* t = tp + (size_t)(a >> 1) * g.LDW;
                        t[0] = fmaxf(prev0, h0);
                        t[1] = fmaxf(prev1, h1);
                        t[2] = fmaxf(prev2, h2);
                    }
                    continue;
                }
                float2* t2 = reinterpret_cast<float2*>(tp + (size_t)a * g.LDW);
                t2[0] = make_float2(y[0], y[1]);
                t2[1] = make_float2(y[2], y[3]);
                t2[2] = make_float2(y[4], y[5]);
            }
        }
        __syncthreads();
        // ---- the next item's tiles: in flight while this item's rows are copied out
        if (item + xcd_blocks < i_hi) fetch(item + xcd_blocks);
        __builtin_amdgcn_sched_barrier(0); // hipcc would sink the loads to their uses
        const int R = POOL ? 3 : 6, OWo = POOL ? q.OW >> 1 : q.OW;
        if ((OWo & 3) == 0)
        {
            const int w4 = OWo >> 2;
            const int total = g.UB * R * w4;
            for (int idx = tid; idx < total; idx += 256)
            {
                const int row = idx / w4, x4 = idx - row * w4;
                const int ul = row / R, a = row - ul * R;
                if (a < urows[ul])
                {
                    const float4 v = *reinterpret_cast<const float4*>(tile + ((size_t)ul * R + a) * g.LDW + 4 * x4);
                    stg4_act<2>(out + ubase[ul] + (long long)a * OWo + 4 * x4, v);
                }
            }
        }
        else
        {
            const int total = g.UB * R * OWo;
            for (int idx = tid; idx < total; idx += 256)
            {
                const int row = idx / OWo, x = idx - row * OWo;
                const int ul = row / R, a = row - ul * R;
                if (a < urows[ul]) out[ubase[ul] + (long long)a * OWo + x] = tile[((size_t)ul * R + a) * g.LDW + x];
            }
        }
        __syncthreads(); // the rows are out (and ubase / urows read): the next item may overwrite the LDS
    }
}

// ---------------------------------------------------------------------------------------------------
// K4 -> K2 chained: Y = A^T m A, + bias, ReLU [, 2x2 max pooling] of layer L, then V' = B^T d B of the 3x3 / stride-1 / pad-1 layer
// that consumes it, WITHOUT the activation tensor in between: a block owns whole (image, channel) planes -- phase 1 transforms the
// plane's M tiles and writes the activation into LDS (zero border = the consumer's padding, cells beyond the image zero), phase 2
// reads the consumer's 8x8 windows from LDS and stores its V.  Same butterflies and the same fp32 values as K4 followed by K2, so
// the chained V is bit-identical; what disappears is one write and one read of every activation between two Winograd layers
// (VGG-16 b32: 1.47 GB of the 6.8 GB the transforms move per step) and one launch per layer.
struct WinoChain
{
    int K, N;         // channels of the plane set (layer L's output = the consumer's input channels), images
    int OH, OW;       // layer L's output image
    int TX, T, Pp;    // layer L's tiling (M columns p = n*T + ty*TX + tx)
    int AH, AW;       // activation the consumer sees: OH x OW, or OH/2 x OW/2 behind the fused pooling
    int TX2, T2, Pp2; // the consumer's tiling / V' pitch
    int planes;       // K * N, image index fastest: the planes of a block are consecutive images of one channel, whose tiles are
                      // consecutive columns of M and V' (a 14 x 14 plane alone is 9 columns = 36 bytes of a row)
    int ppb;          // planes per block
    int LDW, LDH;     // LDS plane: (AH + 2 rows) x LDW floats, 2 border columns left, >= 2 right
    WinoLayout Lm, Lv2; // layer L's M (rows = K) and the consumer's V' (rows = K): wino_layout.h
};

// Persistent blocks, software-pipelined over units of ppb planes (round 4).  A one-shot block is [64 loads per lane] -> A^T m A -> barrier ->
// B^T d B -> [64 stores per lane]: its loads are in flight during the first third of its life only, and the 2 - 3 blocks of a CU drift through
// the same stages (0.51 of the HBM rate over VGG-16's boundaries).  Here a block walks its units, and the NEXT unit's 64 M values per lane are
// requested right after the barrier, before the current unit's windows are read out of LDS, transformed and stored: loads and stores of one
// block overlap (64 more registers -> 3 waves per SIMD, which 3 four-wave or 2 six-wave blocks per CU fit exactly).  VGG-16 b32, same box,
// interleaved: 1.003 -> 0.90 ms over the 12 boundaries, 9423 -> 9707 img/s.  Requesting the next tiles earlier still -- as soon as the column
// pass has consumed m -- needs 10 more registers than 3 waves have, and the spill reloads (scratch is vmcnt-ordered behind the prefetch) bring
// it back to the one-shot time.
//   MULTI: a plane with more tiles than the block has lanes (224 x 224: 1444; ppb = 1) -- the lane's tiles are tid, tid + blockDim, ...; only the
//   first one of the next unit is prefetched.
template <bool HAS_BIAS, bool RELU, bool POOL, bool MULTI>
__global__ __launch_bounds__(512, 3) void wino_chain_kernel(float* __restrict__ Vn, const float* __restrict__ M, const float* __restrict__ bias,
                                                            const WinoChain g, const int units)
{
    extern __shared__ __attribute__((aligned(16))) float smem[]; // [ppb][LDH][LDW]
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int plane_floats = g.LDH * g.LDW;
    // Phase 1 writes rows 1 .. AH x columns 2 .. 2 + CW - 1 of every plane (CW = the columns layer L's tiles cover; cells beyond the image are
    // written as zeros) and nothing else: the consumer's padding and the slack its last tiles read are zeroed once per block
    for (int i = tid; i < g.ppb * plane_floats; i += nthreads) smem[i] = 0.f;
    const int n1 = MULTI ? (g.T + nthreads - 1) / nthreads : 1;
    int pl = MULTI ? 0 : tid / g.T, t = tid - pl * g.T; // phase 1: layer L's tile (ty, tx) of plane pl of the unit
    bool lane_on = MULTI ? t < g.T : pl < g.ppb;
    int ty = t / g.TX, tx = t - ty * g.TX;
    const int pl2 = tid / g.T2, t2 = tid - pl2 * g.T2; // phase 2: the consumer's tile
    const bool lane_on2 = pl2 < g.ppb;
    const int ty2 = t2 / g.TX2, tx2 = t2 - ty2 * g.TX2;
    // unit order: the XCD blockIdx % 8 owns a contiguous eighth of the units, and its blocks take neighbouring units at the same time -- their
    // column runs of M and V' share cache lines at both ends, which merge into whole-line traffic only inside one L2 (tools/chain_bench.py:
    // -2 ... -9 % per boundary on VGG-16)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, xcd_blocks = (gridDim.x + 7 - xcd) >> 3;
    const int u_lo = (int)((long long)units * xcd / 8), u_hi = (int)((long long)units * (xcd + 1) / 8);
    const size_t xi_stride = g.Lm.xis, xi_stride2 = g.Lv2.xis;
    float* const lp1 = smem + pl * plane_floats;

    float m[8][8];
    auto fetch = [&](int unit, int tile) {
        // clamped and unconditional (a load under a branch is waited for on the spot)
        const int plane = min(unit * g.ppb + (lane_on ? pl : 0), g.planes - 1);
        const int k = plane / g.N, n = plane - k * g.N;
        const float* mp = M + (size_t)k * g.Lm.bp + g.Lm.col(n * g.T + min(tile, g.T - 1));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) m[i][jj] = ld_scratch(mp + (size_t)(i * 8 + jj) * xi_stride);
    };
    int unit = u_lo + j;
    if (unit < u_hi) fetch(unit, t);
    __syncthreads();
    for (; unit < u_hi; unit += xcd_blocks)
    {
        const int plane = unit * g.ppb + pl;
        const int k = min(plane, g.planes - 1) / g.N;
        // ---- phase 1: layer L's tiles -> activation plane(s) in LDS
        for (int it = 0; it < n1; ++it)
        {
            if (MULTI)
            {
                t = tid + it * nthreads;
                lane_on = t < g.T;
                ty = t / g.TX;
                tx = t - ty * g.TX;
            }
            if (lane_on && plane < g.planes)
            {
                float tmp[6][8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    at6(m[0][jj], m[1][jj], m[2][jj], m[3][jj], m[4][jj], m[5][jj], m[6][jj], m[7][jj], tmp[0][jj], tmp[1][jj], tmp[2][jj], tmp[3][jj],
                        tmp[4][jj], tmp[5][jj]);
                const float b = HAS_BIAS ? bias[k] : 0.f;
                float prev0 = 0.f, prev1 = 0.f, prev2 = 0.f;
#pragma unroll
                for (int a = 0; a < 6; ++a)
                {
                    float y[6];
                    at6(tmp[a][0], tmp[a][1], tmp[a][2], tmp[a][3], tmp[a][4], tmp[a][5], tmp[a][6], tmp[a][7], y[0], y[1], y[2], y[3], y[4], y[5]);
#pragma unroll
                    for (int bb = 0; bb < 6; ++bb)
                    {
                        float v = y[bb] + b;
                        if (RELU) v = fmaxf(v, 0.f);
                        y[bb] = v;
                    }
                    if (POOL)
                    {
                        // OH, OW even: a 2x2 cell never straddles the image edge; a whole cell is inside or outside
                        const float h0 = fmaxf(y[0], y[1]), h1 = fmaxf(y[2], y[3]), h2 = fmaxf(y[4], y[5]);
                        if ((a & 1) == 0)
                        {
                            prev0 = h0;
                            prev1 = h1;
                            prev2 = h2;
                        }
                        else
                        {
                            const int ay = 3 * ty + (a >> 1);
                            if (ay < g.AH)
                            {
                                float* row = lp1 + (size_t)(ay + 1) * g.LDW + 2 + 3 * tx;
                                row[0] = (3 * tx < g.AW) ? fmaxf(prev0, h0) : 0.f;
                                row[1] = (3 * tx + 1 < g.AW) ? fmaxf(prev1, h1) : 0.f;
                                row[2] = (3 * tx + 2 < g.AW) ? fmaxf(prev2, h2) : 0.f;
                            }
                        }
                        continue;
                    }
                    const int ay = 6 * ty + a;
                    if (ay < g.AH)
                    {
                        float* row = lp1 + (size_t)(ay + 1) * g.LDW + 2 + 6 * tx; // even offset: 8-byte aligned pairs
#pragma unroll
                        for (int bb = 0; bb < 6; bb += 2)
                        {
                            const float v0 = (6 * tx + bb < g.AW) ? y[bb] : 0.f, v1 = (6 * tx + bb + 1 < g.AW) ? y[bb + 1] : 0.f;
                            *reinterpret_cast<float2*>(row + bb) = make_float2(v0, v1);
                        }
                    }
                }
            }
            if (MULTI && it + 1 < n1) fetch(unit, tid + (it + 1) * nthreads);
        }
        __syncthreads();
        // ---- the next unit's tiles: in flight through phase 2
        if (unit + xcd_blocks < u_hi) fetch(unit + xcd_blocks, MULTI ? tid : t);
        __builtin_amdgcn_sched_barrier(0); // hipcc would sink the loads to their uses
        // ---- phase 2: the consumer's tiles: window rows 6ty-1 .. 6ty+6, columns 6tx-1 .. 6tx+6 of the activation = LDS rows 6ty .. 6ty+7,
        // columns 6tx+1 .. 6tx+8 (one border row on top, two border columns on the left)
        const int plane2 = unit * g.ppb + pl2;
        if (lane_on2 && plane2 < g.planes)
        {
            const int k2 = plane2 / g.N, n2 = plane2 - k2 * g.N;
            const float* lp = smem + pl2 * plane_floats + (size_t)(6 * ty2) * g.LDW + 6 * tx2 + 1;
            float d[8][8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) d[i][jj] = lp[(size_t)i * g.LDW + jj];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) bt8(d[0][jj], d[1][jj], d[2][jj], d[3][jj], d[4][jj], d[5][jj], d[6][jj], d[7][jj]);
#pragma unroll
            for (int i = 0; i < 8; ++i) bt8(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5], d[i][6], d[i][7]);
            float* vp = Vn + (size_t)k2 * g.Lv2.bp + g.Lv2.col(n2 * g.T2 + t2);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) st_scratch(vp + (size_t)(i * 8 + jj) * xi_stride2, d[i][jj]);
        }
        __syncthreads(); // the windows are read: the next phase 1 may overwrite the planes
    }
}

// ---------------------------------------------------------------------------------------------------
// K2 staged (round 4): the input transform of a 3x3 / stride-1 / pad-1 layer on planes that fit the LDS, in the chained kernel's form -- a
// block walks units of ppb (image, channel) planes; a plane comes in as whole 16-byte vectors (NV per lane and unit, the next unit's requested
// before this one's windows are read) and is written into the zero-bordered LDS plane, phase 2 is the chained kernel's.  wino_input_transform_
// kernel reads its 8 x 8 window straight from the image: 64 dword loads per lane at a 24-byte lane stride, 12 cache lines per load, most
// lanes of ResNet-50's 56 / 28 / 14-px planes on the bounds-checked path.  Same butterflies on the same values: bit-identical V.
template <int NV>
__global__ __launch_bounds__(512, 3) void wino_input_staged_kernel(float* __restrict__ Vn, const float* __restrict__ in, const WinoChain g,
                                                                   const int units)
{
    extern __shared__ __attribute__((aligned(16))) float smem[]; // [ppb][LDH][LDW]
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int plane_floats = g.LDH * g.LDW;
    for (int i = tid; i < g.ppb * plane_floats; i += nthreads) smem[i] = 0.f; // borders stay zero: the copies below write the interior only
    const int hw = g.AH * g.AW, hw4 = hw >> 2, unit4 = g.ppb * hw4; // hw % 4 == 0 (launcher)
    const int pl2 = tid / g.T2, t2 = tid - pl2 * g.T2;
    const bool lane_on2 = pl2 < g.ppb;
    const int ty2 = t2 / g.TX2, tx2 = t2 - ty2 * g.TX2;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, xcd_blocks = (gridDim.x + 7 - xcd) >> 3; // unit order: as wino_chain_kernel
    const int u_lo = (int)((long long)units * xcd / 8), u_hi = (int)((long long)units * (xcd + 1) / 8);
    const size_t xi_stride2 = g.Lv2.xis;

    // the lane's NV vectors of a unit: vector e = tid + v * nthreads of the unit's ppb * hw4 (clamped: the loads stay unconditional)
    int e_pl[NV], e_q[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v)
    {
        const int e = min(tid + v * nthreads, unit4 - 1);
        e_pl[v] = e / hw4;
        e_q[v] = e - e_pl[v] * hw4;
    }
    float4 r[NV];
    auto fetch = [&](int unit) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
        {
            const int plane = min(unit * g.ppb + e_pl[v], g.planes - 1);
            const int k = plane / g.N, n = plane - k * g.N;
            r[v] = *reinterpret_cast<const float4*>(in + ((size_t)n * g.K + k) * hw + 4 * e_q[v]);
        }
    };
    int unit = u_lo + j;
    if (unit < u_hi) fetch(unit);
    __syncthreads();
    for (; unit < u_hi; unit += xcd_blocks)
    {
        // ---- phase 1: the planes' pixels -> LDS rows 1 .. AH, columns 2 .. AW + 1
#pragma unroll
        for (int v = 0; v < NV; ++v)
        {
            if (tid + v * nthreads < unit4)
            {
                int y = (4 * e_q[v]) / g.AW, x = 4 * e_q[v] - y * g.AW;
                float* lp = smem + e_pl[v] * plane_floats;
                const float val[4] = {r[v].x, r[v].y, r[v].z, r[v].w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
                {
                    lp[(y + 1) * g.LDW + 2 + x] = val[c];
                    if (++x == g.AW)
                    {
                        x = 0;
                        ++y;
                    }
                }
            }
        }
        __syncthreads();
        if (unit + xcd_blocks < u_hi) fetch(unit + xcd_blocks);
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 2: as wino_chain_kernel
        const int plane2 = unit * g.ppb + pl2;
        if (lane_on2 && plane2 < g.planes)
        {
            const int k2 = plane2 / g.N, n2 = plane2 - k2 * g.N;
            const float* lp = smem + pl2 * plane_floats + (size_t)(6 * ty2) * g.LDW + 6 * tx2 + 1;
            float d[8][8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) d[i][jj] = lp[(size_t)i * g.LDW + jj];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) bt8(d[0][jj], d[1][jj], d[2][jj], d[3][jj], d[4][jj], d[5][jj], d[6][jj], d[7][jj]);
#pragma unroll
            for (int i = 0; i < 8; ++i) bt8(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5], d[i][6], d[i][7]);
            float* vp = Vn + (size_t)k2 * g.Lv2.bp + g.Lv2.col(n2 * g.T2 + t2);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) st_scratch(vp + (size_t)(i * 8 + jj) * xi_stride2, d[i][jj]);
        }
        __syncthreads(); // the windows are read: the next copies may overwrite the planes
    }
}

// ---------------------------------------------------------------------------------------------------
// F(4x4, 3x3) for planes of 7 or 8 output pixels per side (round 4; ResNet-50's res5 3x3 layers).  The reference sends such layers to
// IM2COL (avx/booster.cpp:289: h, w <= 8); this library's tuned rule runs them as Winograd, and on a 7 x 7 plane F(6x6,3x3) needs 2 x 2 tiles
// of 6 x 6 outputs -- 144 computed for 49 used -- with 64 frequency points each.  2 x 2 tiles of 4 x 4 outputs cover 8 x 8 with 36 frequency
// points: 0.5625 x the tile-GEMM work and 0.5625 x the V / M bytes for the same layer.  Same pipeline (filter transform -> input transform ->
// 36 GEMMs through the same kernels -> output transform), same layouts with 36 in place of 64; Lavin's matrices
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// (smaller constants than F(6,3): the results are closer to the fp64 convolution than the F(6,3) route's).
__device__ __forceinline__ void bt6(float& r0, float& r1, float& r2, float& r3, float& r4, float& r5)
{
    const float o0 = (4.f * r0 - 5.f * r2) + r4;
    const float o5 = (4.f * r1 - 5.f * r3) + r5;
    const float a = r4 - 4.f * r2, b = r3 - 4.f * r1; // rows 1, 2: a +- b
    const float c = r4 - r2, e = 2.f * (r3 - r1);     // rows 3, 4: c +- e
    r0 = o0;
    r1 = a + b;
    r2 = a - b;
    r3 = c + e;
    r4 = c - e;
    r5 = o5;
}

__device__ __forceinline__ void at4(float m0, float m1, float m2, float m3, float m4, float m5, float& s0, float& s1, float& s2, float& s3)
{
    const float a12 = m1 + m2, d12 = m1 - m2, a34 = m3 + m4, d34 = m3 - m4;
    s0 = (m0 + a12) + a34;
    s1 = d12 + 2.f * d34;
    s2 = a12 + 4.f * a34;
    s3 = (d12 + 8.f * d34) + m5;
}

__global__ __launch_bounds__(256) void wino43_filter_transform_kernel(float* __restrict__ U, const float* __restrict__ w, int C, int K, int Cp, int Kp)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (k >= K) return;
    const float* g = w + ((size_t)k * C + c) * 9;
    float gg[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) gg[i][j] = g[i * 3 + j];
    const float G[6][3] = {{0.25f, 0.0f, 0.0f},           {-1.0f / 6, -1.0f / 6, -1.0f / 6}, {-1.0f / 6, 1.0f / 6, -1.0f / 6},
                           {1.0f / 24, 1.0f / 12, 1.0f / 6}, {1.0f / 24, -1.0f / 12, 1.0f / 6},   {0.0f, 0.0f, 1.0f}};
    float mid[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) mid[i][j] = G[i][0] * gg[0][j] + G[i][1] * gg[1][j] + G[i][2] * gg[2][j];
    const size_t xi_stride = (size_t)Cp * Kp;
    float* up = U + (size_t)c * Kp + k;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) up[(size_t)(i * 6 + j) * xi_stride] = mid[i][0] * G[j][0] + mid[i][1] * G[j][1] + mid[i][2] * G[j][2];
}

// V = B^T d B on 6x6 tiles at stride 4, one tile per lane (lanes along the column index p, as in K2)
__global__ __launch_bounds__(256) void wino43_input_transform_kernel(float* __restrict__ V, const float* __restrict__ in, const WinoXformParams q)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)q.C * q.P) return;
    const int c = (int)(idx / q.P);
    const int p = (int)(idx - (long long)c * q.P);
    const int n = p / q.T, t = p - n * q.T;
    const int ty = t / q.TX, tx = t - ty * q.TX;
    const int y0 = ty * 4 - q.PT, x0 = tx * 4 - q.PL;
    const float* ip = in + ((size_t)n * q.C + c) * q.H * q.W;
    float d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
    {
        const int y = y0 + i;
        const bool yok = (unsigned)y < (unsigned)q.H;
#pragma unroll
        for (int j = 0; j < 6; ++j)
        {
            const int x = x0 + j;
            const bool ok = yok && ((unsigned)x < (unsigned)q.W);
            d[i][j] = ok ? ip[(size_t)y * q.W + x] : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
#pragma unroll
    for (int i = 0; i < 6; ++i) bt6(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
    const size_t xi_stride = q.Lv.xis;
    float* vp = V + (size_t)c * q.Lv.bp + q.Lv.col(p);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) st_scratch(vp + (size_t)(i * 6 + j) * xi_stride, d[i][j]);
}

// Y = A^T m A, + bias, ReLU, clipped 4x4 store.  Planes of at most 8 x 8: a (image, channel) plane is the 2 x 2 tiles of four consecutive lanes.
template <bool HAS_BIAS, bool RELU>
__global__ __launch_bounds__(256) void wino43_output_transform_kernel(float* __restrict__ out, const float* __restrict__ M, const float* __restrict__ bias,
                                                                     const WinoXformParams q)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (p >= q.P) return;
    const int n = p / q.T, t = p - n * q.T;
    const int ty = t / q.TX, tx = t - ty * q.TX;
    const size_t xi_stride = q.Lm.xis;
    const float* mp = M + (size_t)k * q.Lm.bp + q.Lm.col(p);
    float m[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) m[i][j] = ld_scratch(mp + (size_t)(i * 6 + j) * xi_stride);
    float tmp[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) at4(m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j], tmp[0][j], tmp[1][j], tmp[2][j], tmp[3][j]);
    const float b = HAS_BIAS ? bias[k] : 0.f;
    const int oy0 = ty * 4, ox0 = tx * 4;
    float* op = out + (((size_t)n * q.K + k) * q.OH + oy0) * q.OW + ox0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
    {
        float y[4];
        at4(tmp[a][0], tmp[a][1], tmp[a][2], tmp[a][3], tmp[a][4], tmp[a][5], y[0], y[1], y[2], y[3]);
        if (oy0 + a < q.OH)
        {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
                if (ox0 + bb < q.OW)
                {
                    float v = y[bb] + b;
                    if (RELU) v = fmaxf(v, 0.f);
                    op[(size_t)a * q.OW + bb] = v;
                }
        }
    }
}

// K3: the tile GEMM is gemm_core.h driven by WinoGemmPolicy (wino_gemm_policy.h)
using WinoShapeBig = GemmShape<128, 64, 16, 2, 2>;     // K > 64: measured best on C >= 256 (73 % vs 70 %) and on small P
using WinoShapeSmallM = GemmShape<64, 128, 16, 1, 4>;
constexpr int kWinoColTile = 128; // column padding of V / M
constexpr int kWinoKTile = 16;    // reduction padding of U

static bool wino_small_m(int K) { return K <= 64; }
#ifndef FHIP_ROW_SPLIT_ROUNDS
#define FHIP_ROW_SPLIT_ROUNDS 8
#endif
#ifndef FHIP_K4_PERSIST
#define FHIP_K4_PERSIST 3
#endif
constexpr size_t kK4PersistBlocksPerCu = FHIP_K4_PERSIST; // persistent staged output transform: resident blocks per CU (0: always the one-shot kernel)
constexpr int kWinoRowSplitMaxRounds = FHIP_ROW_SPLIT_ROUNDS; // wino_gemm_row_split: launches of at most this many tiles per CU (0: never)

// F(4x4,3x3) instead of F(6x6,3x3): planes of at most 8 output pixels per side on which 4 x 4 output tiles need fewer (tile, frequency point)
// pairs than 6 x 6 ones -- 7 and 8 pixels per side (2 x 2 x 36 = 144 against 2 x 2 x 64 = 256); 5- and 6-pixel planes are ONE F(6,3) tile.
// A pure function of the geometry (GetBufferSize, Init, Forward and the stage-level entry points agree); the chained / pooled / first-layer
// forms are F(6,3)-only and refuse such layers.
static bool wino_f43(const fhip_conv_param& p)
{
    if (p.kernel_h != 3 || p.kernel_w != 3 || p.stride_h > 1 || p.stride_w > 1 || p.group > 1) return false;
    const int oh = p.input_h + p.pad_top + p.pad_bottom - 2, ow = p.input_w + p.pad_left + p.pad_right - 2; // = AssignOutputDim's
    if (oh < 1 || ow < 1 || oh > 8 || ow > 8 || (oh < 7 && ow < 7)) return false; // planes up to 6 x 6 are ONE F(6,3) tile and keep their chains
    return ((oh + 3) / 4) * ((ow + 3) / 4) * 36 < ((oh + 5) / 6) * ((ow + 5) / 6) * 64;
}

// Column block BP of a layer's V and M (wino_layout.h): a pure function of the geometry, like everything else GetBufferSize, the stage-level
// entry points and Forward must agree on.  0 = whole rows.  FHIP_WINO_BP (build-time, tools/layout_ab.sh) overrides the rule for A/B runs.
// Measured (round 4, VGG-16 b32 and ResNet-50 b64 in the net, tools/layout_ab.sh + tools/layout_trace.sh: every layer on BP = 0 / 256 / 512 /
// 1024 / 2048 / 4096): only the HBM-bound tile GEMM of the 64 -> 64-channel layers gains -- VGG-16's conv1_2 321.7 -> 300.8 us at BP = 512,
// 291.0 at 1024 (tools/gemm_bench.hip GEMM_LAYOUT=1 in isolation: 396 -> 378 / 328 / 319 us at BP = 128 / 256 / 512) -- while the MFMA-bound
// GEMMs do not move (+-1 %), conv2_1's (64 -> 128) gets 1 - 3 % slower and the chained transforms 0 - 7 % slower at BP <= 512.  So: blocks of
// 1024 columns for layers with at most 64 output channels (the 64 x 128 GEMM tile), whole rows everywhere else.
#ifdef FHIP_WINO_BP
static int wino_col_block(const fhip_conv_param&, int columns) { return columns > FHIP_WINO_BP ? FHIP_WINO_BP : 0; }
#else
static int wino_col_block(const fhip_conv_param& p, int columns) { return (wino_small_m(p.output_channels) && columns > 1024) ? 1024 : 0; }
#endif

int winograd_plan(const fhip_conv_param& p, int batch, fhip_winograd_plan* plan)
{
    if (p.kernel_h != 3 || p.kernel_w != 3 || p.stride_h > 1 || p.stride_w > 1 || p.group > 1)
        return fail(FHIP_E_UNSUPPORTED, "Winograd F(6x6,3x3) needs a 3x3 stride-1 group-1 convolution");
    if (batch < 1) return fail(FHIP_E_BADARG, "batch < 1");
    const int hp = p.input_h + p.pad_top + p.pad_bottom, wp = p.input_w + p.pad_left + p.pad_right;
    if (hp < 3 || wp < 3) return fail(FHIP_E_BADARG, "input smaller than the kernel");
    // nRowBlocks / nColBlocks exactly as WINOGRADF63_Forward (avx/booster.cpp:209-211)
    plan->tiles_x = (wp + 3) / 6;
    plan->tiles_y = (hp + 3) / 6;
    plan->frequency_points = 64;
    plan->tile_outputs = 6;
    if (wino_f43(p))
    {
        // 7- and 8-pixel planes: 2 x 2 tiles of F(4x4,3x3) (36 frequency points) instead of 2 x 2 tiles of F(6x6,3x3) (64)
        plan->tiles_x = (wp + 1) / 4;
        plan->tiles_y = (hp + 1) / 4;
        plan->frequency_points = 36;
        plan->tile_outputs = 4;
    }
    const int nxi = plan->frequency_points;
    plan->tiles_per_image = plan->tiles_x * plan->tiles_y;
    const long long P = (long long)plan->tiles_per_image * batch;
    if (P > 0x7fffff00LL) return fail(FHIP_E_BADARG, "too many Winograd tiles for 32-bit column indices");
    plan->columns = (int)P;
    plan->column_block = wino_col_block(p, (int)P);
    plan->columns_padded = round_up((int)P, std::max(kWinoColTile, plan->column_block));
    if (plan->column_block <= 0 || plan->column_block >= plan->columns_padded) plan->column_block = plan->columns_padded; // whole rows
    plan->in_channels_padded = round_up(p.input_channels, kWinoKTile);
    plan->out_channels_padded = round_up(p.output_channels, wino_small_m(p.output_channels) ? 64 : 128);
    plan->v_offset_bytes = 0;
    plan->v_bytes = round_up_sz((size_t)nxi * p.input_channels * plan->columns_padded * sizeof(float), 256);
    plan->m_offset_bytes = plan->v_bytes;
    plan->m_bytes = round_up_sz((size_t)nxi * p.output_channels * plan->columns_padded * sizeof(float), 256);
    plan->u_bytes = (size_t)nxi * plan->in_channels_padded * plan->out_channels_padded * sizeof(float);
    return FHIP_OK;
}

static WinoXformParams xform_params(const fhip_conv_param& p, int batch, const fhip_winograd_plan& pl)
{
    WinoXformParams q;
    q.C = p.input_channels;
    q.K = p.output_channels;
    q.H = p.input_h;
    q.W = p.input_w;
    q.OH = p.output_h;
    q.OW = p.output_w;
    q.PL = p.pad_left;
    q.PT = p.pad_top;
    q.TX = pl.tiles_x;
    q.T = pl.tiles_per_image;
    q.P = pl.columns;
    q.Pp = pl.columns_padded;
    q.N = batch;
    q.Lv = wino_layout(q.C, q.Pp, pl.column_block, pl.frequency_points);
    q.Lm = wino_layout(q.K, q.Pp, pl.column_block, pl.frequency_points);
    return q;
}

int winograd_transform_kernel(const fhip_conv_param& p, float* u, const float* kernel, hipStream_t s)
{
    fhip_winograd_plan pl;
    int rc = winograd_plan(p, 1, &pl);
    if (rc) return rc;
    StageTimer tm(FHIP_STAGE_INIT, s);
    FHIP_CHECK_HIP(hipMemsetAsync(u, 0, pl.u_bytes, s));
    dim3 grid(ceil_div(p.output_channels, 256), p.input_channels);
    if (pl.frequency_points == 36)
        hipLaunchKernelGGL(wino43_filter_transform_kernel, grid, dim3(256), 0, s, u, kernel, p.input_channels, p.output_channels, pl.in_channels_padded,
                           pl.out_channels_padded);
    else
        hipLaunchKernelGGL(wino_filter_transform_kernel, grid, dim3(256), 0, s, u, kernel, p.input_channels, p.output_channels,
                           pl.in_channels_padded, pl.out_channels_padded);
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

// The staged input transform where it applies (pad 1 all round, whole 16-byte vectors per plane, a plane within 48 KB of LDS, at most 7 vectors
// per lane and unit); false = not launched.
static bool winograd_input_staged(const fhip_conv_param& p, int batch, const fhip_winograd_plan& pl, float* v, const float* input, hipStream_t s)
{
#ifdef FHIP_K2_DIRECT
    return false;
#endif
    if (p.pad_left != 1 || p.pad_top != 1 || p.pad_right != 1 || p.pad_bottom != 1) return false;
    const int hw = p.input_h * p.input_w;
    if (hw & 3) return false;
    if (reinterpret_cast<uintptr_t>(input) & 15) return false; // whole planes as aligned 16-byte vectors: a 4-byte-aligned tensor takes the direct kernel
    WinoChain g;
    g.K = p.input_channels;
    g.N = batch;
    g.OH = g.AH = p.input_h;
    g.OW = g.AW = p.input_w;
    g.TX = g.TX2 = pl.tiles_x;
    g.T = g.T2 = pl.tiles_per_image;
    g.Pp = g.Pp2 = pl.columns_padded;
    g.Lm = g.Lv2 = wino_layout(g.K, g.Pp2, pl.column_block);
    const long long planes = (long long)batch * g.K;
    if (planes > 0x7fffffffLL) return false;
    g.planes = (int)planes;
    g.LDH = 6 * pl.tiles_y + 2;
    g.LDW = 6 * pl.tiles_x + 4;
    const size_t plane_bytes = (size_t)g.LDH * g.LDW * sizeof(float);
    if (plane_bytes > 48 * 1024 || g.T2 > 512) return false;
    int ppb = std::max(1, 256 / g.T2);
    ppb = (int)std::min<size_t>(ppb, (48 * 1024) / plane_bytes);
    const unsigned threads = std::max(256u, (unsigned)(g.T2 + 63) / 64 * 64);
    // at most 7 vectors per lane and unit: the 8th costs the registers 3 waves per SIMD have (spill reloads are vmcnt-ordered behind the prefetch)
    while (ppb > 1 && ceil_div(ppb * (hw >> 2), (int)threads) > 7) --ppb;
    g.ppb = (int)std::min<long long>(ppb, planes);
    const int nv = ceil_div(g.ppb * (hw >> 2), (int)threads);
    if (nv > 7) return false;
    const size_t lds = plane_bytes * g.ppb;
    const long long units = (planes + g.ppb - 1) / g.ppb;
    const int bpc = std::max(1, std::min((int)(device_lds_bytes() / lds), 12 / (int)(threads / 64)));
    const unsigned grid = ((unsigned)std::min<long long>(units, (long long)device_compute_units() * bpc) + 7u) & ~7u;
#define FHIP_K2S(NV_) case NV_: hipLaunchKernelGGL((wino_input_staged_kernel<NV_>), dim3(grid), dim3(threads), lds, s, v, input, g, (int)units); break
    switch (nv)
    {
        FHIP_K2S(1);
        FHIP_K2S(2);
        FHIP_K2S(3);
        FHIP_K2S(4);
        FHIP_K2S(5);
        FHIP_K2S(6);
        default: FHIP_K2S(7);
    }
#undef FHIP_K2S
    return true;
}

int winograd_input_transform(const fhip_conv_param& p, int batch, float* v, const float* input, hipStream_t s)
{
    fhip_winograd_plan pl;
    int rc = winograd_plan(p, batch, &pl);
    if (rc) return rc;
    const WinoXformParams q = xform_params(p, batch, pl);
    StageTimer tm(FHIP_STAGE_WINO_INPUT, s);
    const long long work = (long long)q.C * q.P;
    if ((work + 255) / 256 > 0x7fffffffLL) return fail(FHIP_E_BADARG, "input transform grid too large");
    dim3 grid((unsigned)((work + 255) / 256));
    if (pl.frequency_points == 36)
        hipLaunchKernelGGL(wino43_input_transform_kernel, grid, dim3(256), 0, s, v, input, q);
    else if (!winograd_input_staged(p, batch, pl, v, input, s))
        hipLaunchKernelGGL(wino_input_transform_kernel, grid, dim3(256), 0, s, v, input, q);
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

// The first layer (3x3 / stride 1 / pad 1, 2 .. 4 input channels -- with 1 channel and group 1 ConvParam makes it a depthwise layer,
// booster.h:113-125 --, any number of output channels) inside the input transform of `next`,
// the 3x3 / stride-1 / pad-1 Winograd layer that is its only consumer (wino_first.h).
bool winograd_can_fuse_first(const fhip_conv_param& first, const fhip_conv_param& next, int batch)
{
    auto k3p1 = [](const fhip_conv_param& c) {
        return c.kernel_h == 3 && c.kernel_w == 3 && c.stride_h == 1 && c.stride_w == 1 && c.group == 1 && c.pad_left == 1 && c.pad_right == 1 &&
               c.pad_top == 1 && c.pad_bottom == 1;
    };
    if (!k3p1(first) || !k3p1(next) || batch < 1 || wino_f43(next)) return false;
    if (first.input_channels < 2 || first.input_channels > 4 || next.input_channels != first.output_channels) return false;
    if (first.activation != FHIP_ACT_NONE && first.activation != FHIP_ACT_RELU) return false;
    if (next.input_h != first.input_h || next.input_w != first.input_w || (first.input_w & 1)) return false; // float2 image reads: even rows
    const unsigned long long in_bytes = 4ull * batch * first.input_channels * first.input_h * first.input_w;
    return in_bytes < kWinoFirstOob && first.output_channels <= 65535;
}

int winograd_input_from_first(const fhip_conv_param& first, const fhip_conv_param& next, int batch, float* v, const float* input,
                              const float* first_kernel, const float* first_bias, hipStream_t s)
{
    if (!winograd_can_fuse_first(first, next, batch))
        return fail(FHIP_E_UNSUPPORTED, "this first layer cannot be computed inside the next layer's input transform (fhip_conv_can_fuse_first_winograd)");
    if (first.bias_term && !first_bias) return fail(FHIP_E_BADARG, "bias_term set but bias_arr is NULL");
    fhip_winograd_plan pl;
    const int rc = winograd_plan(next, batch, &pl);
    if (rc) return rc;
    WinoFirstParams q;
    q.in = input;
    q.w = first_kernel;
    q.bias = first.bias_term ? first_bias : nullptr;
    q.V = v;
    q.K = first.output_channels;
    q.H = first.input_h;
    q.W = first.input_w;
    q.TX = pl.tiles_x;
    q.T = pl.tiles_per_image;
    q.P = pl.columns;
    q.Pp = pl.columns_padded;
    q.Lv = wino_layout(q.K, q.Pp, pl.column_block);
    q.in_bytes = (unsigned)(4ull * batch * first.input_channels * first.input_h * first.input_w);
    q.relu = first.activation == FHIP_ACT_RELU;
    StageTimer tm(FHIP_STAGE_WINO_INPUT, s);
    // staged form: 64 (63 with shared columns) consecutive tiles of an image span at most floor((TX + 62) / TX) + 1 tile rows
    q.N = batch;
    const unsigned long long v_bytes = 64ull * q.K * q.Pp * sizeof(float);
    // the row-end tile's columns 6, 7 are padding, and V is within reach of 32-bit buffer offsets (wino_first.h)
#ifdef FHIP_FIRST_NOSHARE
    const bool share = false;
#else
    const bool share = 6 * q.TX - 1 >= q.W && v_bytes <= 0x80000000ull;
#endif
    q.v_bytes = share ? (unsigned)v_bytes : 0u;
    const int tpb = share ? kFirstTiles - 1 : kFirstTiles;
    q.bpi = ceil_div(q.T, tpb);
    q.LDW = 6 * q.TX + 4;
    q.rows = 6 * std::min(pl.tiles_y, (q.TX + kFirstTiles - 2) / q.TX + 1) + 4;
    const size_t lds = (size_t)first.input_channels * q.rows * q.LDW * sizeof(float);
    // the staged form gives every image blocks of its own: below 3/4 full lanes (small planes: 14 x 14 has 9 tiles) the direct form,
    // whose lanes run across images, is the better one
    const bool lanes_full = 4 * q.T >= 3 * q.bpi * tpb;
    if (lanes_full && lds <= 64 * 1024 && q.LDW <= 256 && (long long)q.bpi * batch <= 0x7fffffffLL) // 128 float2 columns: one per thread pair
    {
        const dim3 grid((unsigned)(q.bpi * batch), (unsigned)ceil_div(q.K, kFirstCpb));
#define FHIP_FIRST(C_)                                                                                                \
    do                                                                                                                \
    {                                                                                                                 \
        if (share) hipLaunchKernelGGL((wino_input_from_first_staged_kernel<C_, true>), grid, dim3(256), lds, s, q);  \
        else hipLaunchKernelGGL((wino_input_from_first_staged_kernel<C_, false>), grid, dim3(256), lds, s, q);       \
    } while (0)
        switch (first.input_channels)
        {
            case 2: FHIP_FIRST(2); break;
            case 3: FHIP_FIRST(3); break;
            default: FHIP_FIRST(4); break;
        }
#undef FHIP_FIRST
        FHIP_CHECK_HIP(hipGetLastError());
        return FHIP_OK;
    }
    const dim3 grid((unsigned)ceil_div(q.P, 256), (unsigned)q.K); // image rows too wide for the LDS: read them through the L1
    switch (first.input_channels)
    {
        case 2: hipLaunchKernelGGL(wino_input_from_first_kernel<2>, grid, dim3(256), 0, s, q); break;
        case 3: hipLaunchKernelGGL(wino_input_from_first_kernel<3>, grid, dim3(256), 0, s, q); break;
        default: hipLaunchKernelGGL(wino_input_from_first_kernel<4>, grid, dim3(256), 0, s, q); break;
    }
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

template <int NT>
static void wino_tile_gemm_launch(WinoGemmPolicy::Params& g, int columns, hipStream_t s)
{
    const bool v_nt = FHIP_V_NT_ONCE && (size_t)g.batches * g.C * g.Pp * sizeof(float) >= (size_t)FHIP_V_NT_BYTES;
    if (wino_small_m(g.K))
    {
        g.m_tiles = g.Kp / WinoShapeSmallM::BM;
        g.n_tiles = ceil_div(columns, WinoShapeSmallM::BN); // Pp is a multiple of every BN: no pure-padding tiles
        if (v_nt)
            hipLaunchKernelGGL((gemm_mfma_kernel<WinoShapeSmallM, WinoGemmPolicyT<NT | 1>>), dim3(g.batches * g.m_tiles * g.n_tiles),
                               dim3(WinoShapeSmallM::THREADS), 0, s, g);
        else
            hipLaunchKernelGGL((gemm_mfma_kernel<WinoShapeSmallM, WinoGemmPolicyT<NT>>), dim3(g.batches * g.m_tiles * g.n_tiles),
                               dim3(WinoShapeSmallM::THREADS), 0, s, g);
        return;
    }
    g.m_tiles = g.Kp / WinoShapeBig::BM;
    g.n_tiles = ceil_div(columns, WinoShapeBig::BN);
    if (g.k_tiles >= 8 && wino_gemm_prefers_96(columns))
    {
        // column counts that 64-column tiles pad by a third of a tile or more (VGG-16 conv5 b32: 288 = 3 x 96)
        g.n_tiles = ceil_div(columns, 96);
        hipLaunchKernelGGL(wino_gemm_glds96_kernel<NT>, dim3(g.batches * g.m_tiles * g.n_tiles), dim3(256), 0, s, g);
    }
    else if (g.k_tiles >= 8) // C = 64 (4 k-tiles, HBM bound): the register-staged loop is 6 % faster (83.9 vs 78.7 TF)
    {
        const int tiles = g.batches * g.m_tiles * g.n_tiles;
        wino_gemm_row_split(tiles, device_compute_units(), kWinoRowSplitMaxRounds, g.tail_first, g.tail_parts);
        if (g.tail_parts > 1)
            hipLaunchKernelGGL((wino_gemm_glds_kernel<2, 16, 6, NT, true>), dim3(wino_gemm_row_split_grid(tiles, g.tail_first, g.tail_parts)), dim3(256), 0, s, g);
        else
            hipLaunchKernelGGL((wino_gemm_glds_kernel<2, 16, 6, NT>), dim3(tiles), dim3(256), 0, s, g);
    }
    else if (v_nt && g.m_tiles == 1) // only while one row tile covers all of K: with more, every V element is read by m_tiles blocks and must stay cached
        hipLaunchKernelGGL((gemm_mfma_kernel<WinoShapeBig, WinoGemmPolicyT<NT | 1>>), dim3(g.batches * g.m_tiles * g.n_tiles),
                           dim3(WinoShapeBig::THREADS), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_mfma_kernel<WinoShapeBig, WinoGemmPolicyT<NT>>), dim3(g.batches * g.m_tiles * g.n_tiles),
                           dim3(WinoShapeBig::THREADS), 0, s, g);
}

int winograd_tile_gemm(const fhip_conv_param& p, int batch, float* m, const float* u, const float* v, hipStream_t s)
{
    fhip_winograd_plan pl;
    int rc = winograd_plan(p, batch, &pl);
    if (rc) return rc;
    WinoGemmPolicy::Params g;
    g.batches = pl.frequency_points;
    g.U = u;
    g.V = v;
    g.M = m;
    g.C = p.input_channels;
    g.K = p.output_channels;
    g.Cp = pl.in_channels_padded;
    g.Kp = pl.out_channels_padded;
    g.Pp = pl.columns_padded;
    g.Lv = wino_layout(g.C, g.Pp, pl.column_block, pl.frequency_points);
    g.Lm = wino_layout(g.K, g.Pp, pl.column_block, pl.frequency_points);
    g.k_tiles = g.Cp / kWinoKTile;
    StageTimer tm(FHIP_STAGE_WINO_GEMM, s);
    // M is written once here and read once by the next launch.  A large M cannot stay in any cache until then: its stores take the
    // streaming policy FHIP_M_NT_BIG; a small one keeps FHIP_M_NT_SMALL (wino_gemm_policy.h)
    const bool big_m = (size_t)pl.frequency_points * g.K * g.Pp * sizeof(float) >= (size_t)FHIP_M_NT_BYTES;
    if (big_m)
        wino_tile_gemm_launch<FHIP_M_NT_BIG>(g, pl.columns, s);
    else
        wino_tile_gemm_launch<FHIP_M_NT_SMALL>(g, pl.columns, s);
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

template <bool POOL>
static void launch_staged_persist(dim3 grid, size_t lds, hipStream_t s, bool has_bias, bool relu, float* output, const float* m, const float* bias,
                                  const WinoXformParams& q, const WinoStaged& g, int xblocks, int items)
{
    if (has_bias && relu)
        hipLaunchKernelGGL((wino_output_transform_persist_kernel<true, true, POOL>), grid, dim3(256), lds, s, output, m, bias, q, g, xblocks, items);
    else if (has_bias)
        hipLaunchKernelGGL((wino_output_transform_persist_kernel<true, false, POOL>), grid, dim3(256), lds, s, output, m, bias, q, g, xblocks, items);
    else if (relu)
        hipLaunchKernelGGL((wino_output_transform_persist_kernel<false, true, POOL>), grid, dim3(256), lds, s, output, m, bias, q, g, xblocks, items);
    else
        hipLaunchKernelGGL((wino_output_transform_persist_kernel<false, false, POOL>), grid, dim3(256), lds, s, output, m, bias, q, g, xblocks, items);
}

template <bool POOL>
static void launch_staged(dim3 grid, size_t lds, hipStream_t s, bool has_bias, bool relu, float* output, const float* m, const float* bias,
                          const WinoXformParams& q, const WinoStaged& g)
{
    if (has_bias && relu)
        hipLaunchKernelGGL((wino_output_transform_staged_kernel<true, true, POOL>), grid, dim3(256), lds, s, output, m, bias, q, g);
    else if (has_bias)
        hipLaunchKernelGGL((wino_output_transform_staged_kernel<true, false, POOL>), grid, dim3(256), lds, s, output, m, bias, q, g);
    else if (relu)
        hipLaunchKernelGGL((wino_output_transform_staged_kernel<false, true, POOL>), grid, dim3(256), lds, s, output, m, bias, q, g);
    else
        hipLaunchKernelGGL((wino_output_transform_staged_kernel<false, false, POOL>), grid, dim3(256), lds, s, output, m, bias, q, g);
}

bool winograd_can_pool(const fhip_conv_param& p)
{
    if (wino_f43(p)) return false; // the pooled output transform is F(6,3)-only
    return (p.output_h % 2) == 0 && (p.output_w % 2) == 0 && ceil_div(p.output_w, 6) <= 256;
}

// pool != 0: `output` is the 2x2 / stride-2 max-pooled tensor [N][K][OH/2][OW/2] (winograd_can_pool must hold)
int winograd_output_transform(const fhip_conv_param& p, int batch, float* output, const float* m, const float* bias,
                              hipStream_t s, int pool)
{
    fhip_winograd_plan pl;
    int rc = winograd_plan(p, batch, &pl);
    if (rc) return rc;
    const WinoXformParams q = xform_params(p, batch, pl);
    const bool has_bias = p.bias_term != 0, relu = p.activation == FHIP_ACT_RELU;
    if (has_bias && !bias) return fail(FHIP_E_BADARG, "bias_term set but bias_arr is NULL");
    if (pool && !winograd_can_pool(p)) return fail(FHIP_E_UNSUPPORTED, "fused 2x2 max pooling needs even output dims");
    StageTimer tm(FHIP_STAGE_WINO_OUTPUT, s);
    if (pl.frequency_points == 36)
    {
        if (pool) return fail(FHIP_E_UNSUPPORTED, "fused max pooling is not available on F(4x4,3x3) planes");
        dim3 grid(ceil_div(q.P, 256), q.K);
        if (has_bias && relu)
            hipLaunchKernelGGL((wino43_output_transform_kernel<true, true>), grid, dim3(256), 0, s, output, m, bias, q);
        else if (has_bias)
            hipLaunchKernelGGL((wino43_output_transform_kernel<true, false>), grid, dim3(256), 0, s, output, m, bias, q);
        else if (relu)
            hipLaunchKernelGGL((wino43_output_transform_kernel<false, true>), grid, dim3(256), 0, s, output, m, bias, q);
        else
            hipLaunchKernelGGL((wino43_output_transform_kernel<false, false>), grid, dim3(256), 0, s, output, m, bias, q);
        FHIP_CHECK_HIP(hipGetLastError());
        return FHIP_OK;
    }
    if (q.TX <= 256)
    {
        WinoStaged g;
        g.TY = pl.tiles_y;
        g.units = batch * pl.tiles_y;
        g.UB = min(256 / q.TX, g.units);
        g.R = pool ? 3 : 6;
        g.LDW = round_up(g.R * q.TX, 4);
        const size_t lds = (size_t)g.UB * g.R * g.LDW * sizeof(float) + (size_t)g.UB * (sizeof(long long) + sizeof(int));
        dim3 sgrid(ceil_div(g.units, g.UB), q.K);
        // more work items than blocks that are resident at once: persistent blocks with the next item's loads under this item's stores
        const long long items = (long long)sgrid.x * sgrid.y;
        const int resident = device_compute_units() * (int)std::min<size_t>(kK4PersistBlocksPerCu, device_lds_bytes() / lds);
        if (kK4PersistBlocksPerCu > 0 && items > resident && items <= 0x7fffffffLL && lds <= 64 * 1024)
        {
            const dim3 pgrid((unsigned)(resident + 7) & ~7u);
            if (pool)
                launch_staged_persist<true>(pgrid, lds, s, has_bias, relu, output, m, bias, q, g, (int)sgrid.x, (int)items);
            else
                launch_staged_persist<false>(pgrid, lds, s, has_bias, relu, output, m, bias, q, g, (int)sgrid.x, (int)items);
        }
        else if (pool)
            launch_staged<true>(sgrid, lds, s, has_bias, relu, output, m, bias, q, g);
        else
            launch_staged<false>(sgrid, lds, s, has_bias, relu, output, m, bias, q, g);
        FHIP_CHECK_HIP(hipGetLastError());
        return FHIP_OK;
    }
    if (pool) return fail(FHIP_E_UNSUPPORTED, "pooled output transform needs the staged kernel");
    dim3 grid(ceil_div(q.P, 256), q.K);
    if (has_bias && relu)
        hipLaunchKernelGGL((wino_output_transform_kernel<true, true>), grid, dim3(256), 0, s, output, m, bias, q);
    else if (has_bias)
        hipLaunchKernelGGL((wino_output_transform_kernel<true, false>), grid, dim3(256), 0, s, output, m, bias, q);
    else if (relu)
        hipLaunchKernelGGL((wino_output_transform_kernel<false, true>), grid, dim3(256), 0, s, output, m, bias, q);
    else
        hipLaunchKernelGGL((wino_output_transform_kernel<false, false>), grid, dim3(256), 0, s, output, m, bias, q);
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

// Chained output -> input transform.  `next` is the 3x3 / stride-1 / pad-1 Winograd layer that consumes layer p's output (after a
// 2x2 / stride-2 max pooling when pool != 0); `vn` is ITS V buffer ([64][next.C][Pp2], winograd_plan(next, batch)).
bool winograd_can_chain(const fhip_conv_param& p, const fhip_conv_param& next, int pool)
{
    if (p.kernel_h != 3 || p.kernel_w != 3 || p.stride_h > 1 || p.stride_w > 1 || p.group > 1) return false;
    if (next.kernel_h != 3 || next.kernel_w != 3 || next.stride_h > 1 || next.stride_w > 1 || next.group > 1) return false;
    if (next.pad_left != 1 || next.pad_right != 1 || next.pad_top != 1 || next.pad_bottom != 1) return false;
    if (next.input_channels != p.output_channels) return false;
    if (wino_f43(p) || wino_f43(next)) return false; // the chained transform is F(6,3) -> F(6,3) only
    if (pool && ((p.output_h & 1) || (p.output_w & 1))) return false;
    const int ah = pool ? p.output_h / 2 : p.output_h, aw = pool ? p.output_w / 2 : p.output_w;
    if (next.input_h != ah || next.input_w != aw) return false;
    // one plane of the activation (+ borders) must fit the LDS of a block: (ah + 2) x (6 * ceil((aw + 2 + 3) / 6) + 4) floats <= 64 KB
    const int tx2 = (aw + 2 + 3) / 6, ty2 = (ah + 2 + 3) / 6;
    return (size_t)(6 * ty2 + 2) * (6 * tx2 + 4) * sizeof(float) <= 64 * 1024;
}

int winograd_output_to_next_input(const fhip_conv_param& p, const fhip_conv_param& next, int batch, float* vn, const float* m, const float* bias,
                                  hipStream_t s, int pool)
{
    if (!winograd_can_chain(p, next, pool)) return fail(FHIP_E_UNSUPPORTED, "these two layers cannot be chained (fhip_conv_can_chain_winograd)");
    fhip_winograd_plan pl, pn;
    int rc = winograd_plan(p, batch, &pl);
    if (rc) return rc;
    if ((rc = winograd_plan(next, batch, &pn))) return rc;
    const bool has_bias = p.bias_term != 0, relu = p.activation == FHIP_ACT_RELU;
    if (has_bias && !bias) return fail(FHIP_E_BADARG, "bias_term set but bias_arr is NULL");
    WinoChain g;
    g.K = p.output_channels;
    g.N = batch;
    g.OH = p.output_h;
    g.OW = p.output_w;
    g.TX = pl.tiles_x;
    g.T = pl.tiles_per_image;
    g.Pp = pl.columns_padded;
    g.AH = next.input_h;
    g.AW = next.input_w;
    g.TX2 = pn.tiles_x;
    g.T2 = pn.tiles_per_image;
    g.Pp2 = pn.columns_padded;
    g.Lm = wino_layout(g.K, g.Pp, pl.column_block);
    g.Lv2 = wino_layout(g.K, g.Pp2, pn.column_block);
    const long long planes = (long long)batch * g.K;
    if (planes > 0x7fffffffLL) return fail(FHIP_E_BADARG, "N*K too large");
    g.planes = (int)planes;
    g.LDH = 6 * pn.tiles_y + 2;
    g.LDW = 6 * pn.tiles_x + 4;
    const size_t plane_bytes = (size_t)g.LDH * g.LDW * sizeof(float);
    // one tile per lane: as many planes per block as 256 lanes and 48 KB of LDS (three blocks per CU) take; a plane of more than 256
    // tiles (112 x 112: 361) gets a block of its own with a lane per tile
    const int work = std::max(g.T, g.T2);
    int ppb = std::max(1, 256 / work);
    ppb = (int)std::min<size_t>(ppb, std::max<size_t>(1, (48 * 1024) / plane_bytes));
    g.ppb = (int)std::min<long long>(ppb, planes);
    // phase 2 always has a lane per tile (can_chain's 64 KB plane holds at most 455 consumer tiles); phase 1 too, except behind the fused pooling
    // on planes of more than 512 tiles (224 x 224), whose block is sized for phase 2 and loops in phase 1
    const bool multi = work > 512;
    const unsigned threads = std::max(256u, (unsigned)((multi ? g.T2 : work) + 63) / 64 * 64);
    if (threads > 512u) return fail(FHIP_E_UNSUPPORTED, "chained transform: the consumer's plane has more than 512 tiles");
    const size_t lds = plane_bytes * g.ppb;
    const long long units = (planes + g.ppb - 1) / g.ppb;
    // persistent: what is resident at 3 waves per SIMD (12 waves per CU) and 160 KB of LDS, a multiple of 8 so that every XCD runs the same count
    const int bpc = std::max(1, std::min((int)(device_lds_bytes() / lds), 12 / (int)(threads / 64)));
    const unsigned grid = ((unsigned)std::min<long long>(units, (long long)device_compute_units() * bpc) + 7u) & ~7u;
    StageTimer tm(FHIP_STAGE_WINO_CHAIN, s);
#define FHIP_CHAIN(B_, R_, P_)                                                                                                                       \
    do                                                                                                                                               \
    {                                                                                                                                                \
        if (multi) hipLaunchKernelGGL((wino_chain_kernel<B_, R_, P_, true>), dim3(grid), dim3(threads), lds, s, vn, m, bias, g, (int)units);        \
        else hipLaunchKernelGGL((wino_chain_kernel<B_, R_, P_, false>), dim3(grid), dim3(threads), lds, s, vn, m, bias, g, (int)units);             \
    } while (0)
    if (pool)
    {
        if (has_bias && relu) FHIP_CHAIN(true, true, true);
        else if (has_bias) FHIP_CHAIN(true, false, true);
        else if (relu) FHIP_CHAIN(false, true, true);
        else FHIP_CHAIN(false, false, true);
    }
    else
    {
        if (has_bias && relu) FHIP_CHAIN(true, true, false);
        else if (has_bias) FHIP_CHAIN(true, false, false);
        else if (relu) FHIP_CHAIN(false, true, false);
        else FHIP_CHAIN(false, false, false);
    }
#undef FHIP_CHAIN
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

} // namespace fhip
