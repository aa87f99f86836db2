// tools/b2b_probe.hip -- VERDICT r05 next #3 lever (a): ResNet-50's res2 hand-over  expand 1x1 (64 -> 256, + residual, ReLU)  ->  next block's reduce
// 1x1 (256 -> 64, ReLU)  at 56 x 56, batch 64, as ONE kernel against the product's two launches (fhip_conv_forward_residual + fhip_conv_forward).
// Two launches move X 51 + R 205 + T 205 (write) + T 205 (read back) + Y 51 = 717 MB; fused, T is written once and never read back: 512 MB.
//
// Structure measured here (not the LDS-held 256 x 64 tile the verdict sketched -- that is one block per CU): the intermediate never leaves the
// REGISTERS.  A wave owns 64 of the 256 middle channels.  GEMM1 (K = 64) leaves its 64 x 32 tile of T in the MFMA C/D layout: register r of lane l
// holds T[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].  v_mfma_f32_32x32x2_f32 takes its B operand as B[k = l >> 5][j = l & 31] -- the same column
// per lane -- so accumulator register r IS a B fragment for the k pair {row(r), row(r) + 4}; GEMM2 (Y = W2 T) walks the wave's own 64 middle
// channels in that order with W2's fragments pre-permuted to match.  No LDS round trip for T, no barrier between the GEMMs; both filter matrices
// live in registers for the life of a persistent block (64 + 64 VGPRs); the four waves' partial Y tiles meet in LDS (fixed order).  The residual
// operand is loaded straight into the accumulators (acc = R + b1, then += W1 X), one tile ahead.
// Not part of the product.   usage: b2b_probe [reps]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "feather_hip/feather_hip.h"
#include "feather_hip/feather_net.h"

#define CK(x)                                                                            \
    do                                                                                   \
    {                                                                                    \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess)                                                             \
        {                                                                                \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)
#define CF(x)                                                                                   \
    do                                                                                          \
    {                                                                                           \
        int rc = (x);                                                                           \
        if (rc)                                                                                 \
        {                                                                                       \
            printf("fhip error %d (%s) at %s:%d\n", rc, fhip_last_error(), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct B2bParams
{
    const float* X;   // [N][C1][HW]
    const float* R;   // [N][KM][HW] residual operand of the expand layer
    const float* W1;  // [KM][C1]
    const float* b1;  // [KM]
    const float* W2;  // [K2][KM]
    const float* b2;  // [K2]
    float* T;         // [N][KM][HW]  = relu(W1 X + b1 + R)
    float* Y;         // [N][K2][HW]  = relu(W2 T + b2)
    int HW, tiles, tiles_per_image;
};

constexpr int C1 = 64, KM = 256, K2 = 64, BN = 32;

__device__ __forceinline__ int row_of(int r) { return (r & 3) + 8 * (r >> 2); }

// ABL: 1 no residual loads, 2 no T stores, 4 no Y stores / reduction, 8 no X loads (timing-only builds)
template <int ABL>
__global__ __launch_bounds__(256, 2) void b2b_kernel(const B2bParams p)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * C1 * BN + 4 * K2 * BN + KM + K2];
    float* const Xs = lds;                     // 2 x [64 k][32 n]
    float* const red = lds + 2 * C1 * BN;      // [4 waves][64 rows][32 cols]
    float* const b1s = red + 4 * K2 * BN;      // [256]
    float* const b2s = b1s + KM;               // [64]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, half = lane >> 5;

    // ---- the two filter matrices as MFMA A fragments, once per block
    float a1[2][32], a2[2][32];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kp = 0; kp < 32; ++kp) a1[i][kp] = p.W1[(size_t)(64 * w + 32 * i + l31) * C1 + 2 * kp + half];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < 32; ++j) a2[i2][j] = p.W2[(size_t)(32 * i2 + l31) * KM + 64 * w + 32 * (j >> 4) + row_of(j & 15) + 4 * half];
    b1s[tid] = p.b1[tid];
    if (tid < K2) b2s[tid] = p.b2[tid];

    const int G = gridDim.x;
    int tile = blockIdx.x;
    if (tile >= p.tiles) return;
    // loader mapping of the X tile: 512 float4 = [64 k][8 float4]; thread t takes (k = t / 8 (+32), c4 = t % 8)
    const int xk = tid >> 3, xc4 = tid & 7;
    auto x_ptr = [&](int t, int kk) {
        const int img = t / p.tiles_per_image, pix0 = (t - img * p.tiles_per_image) * BN;
        return reinterpret_cast<const f32x4*>(p.X + ((size_t)img * C1 + kk) * p.HW + pix0 + 4 * xc4);
    };
    f32x4 xv0 = *x_ptr(tile, xk), xv1 = *x_ptr(tile, xk + 32);
    *reinterpret_cast<f32x4*>(&Xs[xk * BN + 4 * xc4]) = xv0;
    *reinterpret_cast<f32x4*>(&Xs[(xk + 32) * BN + 4 * xc4]) = xv1;

    f32x16 acc1[2];
    auto res_load = [&](int t) {
        const int img = t / p.tiles_per_image, pix0 = (t - img * p.tiles_per_image) * BN;
        const float* rp = p.R + ((size_t)img * KM + 64 * w + 4 * half) * p.HW + pix0 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][r] = (ABL & 1) ? 0.f : rp[(size_t)(32 * i + row_of(r)) * p.HW];
    };
    res_load(tile);
    __syncthreads();

    int cur = 0;
    for (; tile < p.tiles; tile += G)
    {
        const int nxt = tile + G;
        const bool more = nxt < p.tiles;
        const int img = tile / p.tiles_per_image, pix0 = (tile - img * p.tiles_per_image) * BN;
        // next tile's X: requested now, written to the other LDS buffer behind GEMM1
        if (more && !(ABL & 8))
        {
            xv0 = *x_ptr(nxt, xk);
            xv1 = *x_ptr(nxt, xk + 32);
        }
        // acc1 = R + b1 (the residual was requested one tile ago)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][r] += b1s[64 * w + 32 * i + row_of(r) + 4 * half];
        // ---- GEMM1: T[64w .. 64w+63][32 columns] += W1 X
        const float* xs = Xs + cur * (C1 * BN) + half * BN + l31;
#pragma unroll
        for (int kp = 0; kp < 32; ++kp)
        {
            const float b = xs[2 * kp * BN];
            acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0][kp], b, acc1[0], 0, 0, 0);
            acc1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1][kp], b, acc1[1], 0, 0, 0);
        }
        if (more)
        {
            *reinterpret_cast<f32x4*>(&Xs[(cur ^ 1) * (C1 * BN) + xk * BN + 4 * xc4]) = xv0;
            *reinterpret_cast<f32x4*>(&Xs[(cur ^ 1) * (C1 * BN) + (xk + 32) * BN + 4 * xc4]) = xv1;
        }
        // ReLU, T out (dword per register: 2 x 128-byte row pieces per instruction)
        float* tp = p.T + ((size_t)img * KM + 64 * w + 4 * half) * p.HW + pix0 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
            {
                acc1[i][r] = fmaxf(acc1[i][r], 0.f);
                if (!(ABL & 2)) tp[(size_t)(32 * i + row_of(r)) * p.HW] = acc1[i][r];
            }
        // ---- GEMM2: this wave's share of Y = W2[:, its 64 middle channels] T: the accumulator registers are the B fragments
        f32x16 acc2[2];
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i2][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j)
        {
            const float b = acc1[j >> 4][j & 15];
            acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[0][j], b, acc2[0], 0, 0, 0);
            acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[1][j], b, acc2[1], 0, 0, 0);
        }
        // next tile's residual operand -> the accumulators GEMM2 has just finished reading
        if (more) res_load(nxt);
        __builtin_amdgcn_sched_barrier(0); // keep the requests here (hipcc sinks loads to their uses)
        if (!(ABL & 4))
        {
            // the four waves' partial Y tiles meet in LDS, summed in wave order
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(w * K2 + 32 * i2 + row_of(r) + 4 * half) * BN + l31] = acc2[i2][r];
            __syncthreads();
            float* yp = p.Y + (size_t)img * K2 * p.HW + pix0;
#pragma unroll
            for (int q = 0; q < 2; ++q)
            {
                const int idx = tid + 256 * q, row = idx >> 3, c4 = idx & 7;
                f32x4 s = *reinterpret_cast<const f32x4*>(&red[(0 * K2 + row) * BN + 4 * c4]);
#pragma unroll
                for (int ww = 1; ww < 4; ++ww) s += *reinterpret_cast<const f32x4*>(&red[(ww * K2 + row) * BN + 4 * c4]);
                const float bb = b2s[row];
                s.x = fmaxf(s.x + bb, 0.f);
                s.y = fmaxf(s.y + bb, 0.f);
                s.z = fmaxf(s.z + bb, 0.f);
                s.w = fmaxf(s.w + bb, 0.f);
                *reinterpret_cast<f32x4*>(yp + (size_t)row * p.HW + 4 * c4) = s;
            }
        }
        __syncthreads(); // red is free again; the other X buffer is visible
        cur ^= 1;
    }
}


// ---- ping-pong form: ONE 512-thread block per CU = two groups of four waves (one of each group per SIMD).  Between two block barriers one group
// runs the 128 MFMAs of its tile while the other does everything that is not matrix work for ITS tiles -- T stores, the partial-Y reduction and
// Y stores of the tile it has just computed, then the residual operand, X and bias of its next tile -- so the memory phase of one group sits under
// the matrix phase of the other on every SIMD.  One barrier per tile.
template <int ABL, bool WIDE = false>
__global__ __launch_bounds__(512, 2) void b2b_pp_kernel(const B2bParams p)
{
    constexpr int SCR = 32 * 36; // WIDE: a wave-private 32 x 32 transpose scratch (row pitch 36), so that T leaves and R arrives as 16 bytes per lane
    __shared__ __attribute__((aligned(16))) float lds[2 * C1 * BN + 2 * 4 * K2 * BN + KM + K2 + (WIDE ? 8 * SCR : 0)];
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 8, t8 = tid & 255, w = t8 >> 6, l31 = lane & 31, half = lane >> 5;
    float* const Xs = lds + g * (C1 * BN);                         // this group's [64 k][32 n]
    float* const red = lds + 2 * C1 * BN + g * (4 * K2 * BN);      // this group's [4 waves][64 rows][32 cols]
    float* const b1s = lds + 2 * C1 * BN + 2 * 4 * K2 * BN;        // [256]
    float* const b2s = b1s + KM;                                   // [64]
    float* const scr = b2s + K2 + (tid >> 6) * SCR;                // (WIDE) this wave's transpose scratch
    const int e_row = lane >> 3, e_c4 = (lane & 7) * 4;

    float a1[2][32], a2[2][32];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kp = 0; kp < 32; ++kp) a1[i][kp] = p.W1[(size_t)(64 * w + 32 * i + l31) * C1 + 2 * kp + half];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < 32; ++j) a2[i2][j] = p.W2[(size_t)(32 * i2 + l31) * KM + 64 * w + 32 * (j >> 4) + row_of(j & 15) + 4 * half];
    if (tid < KM) b1s[tid] = p.b1[tid];
    if (tid < K2) b2s[tid] = p.b2[tid];
    __syncthreads();

    const int G = gridDim.x;
#ifdef B2B_ADJ
    // adjacent tiles to the two groups of a block: tile(q) = 2 (blockIdx + (q >> 1) G) + (q & 1)   (tiles is even)
    const int npairs = 2 * blockIdx.x < p.tiles ? (p.tiles / 2 - blockIdx.x + G - 1) / G : 0;
    const int n = 2 * npairs;
#define TILE_OF(q) (2 * ((int)blockIdx.x + ((q) >> 1) * G) + ((q) & 1))
#else
    const int n = blockIdx.x < p.tiles ? (p.tiles - blockIdx.x + G - 1) / G : 0; // tiles of this block: blockIdx + q G, q < n; group q & 1
#define TILE_OF(q) ((int)blockIdx.x + (q) * G)
#endif
    const int xk = t8 >> 3, xc4 = t8 & 7;
    f32x16 acc1[2], acc2[2];

    auto prologue = [&](int q) { // residual -> accumulators, X -> LDS, + bias
        const int t = TILE_OF(q), img = t / p.tiles_per_image, pix0 = (t - img * p.tiles_per_image) * BN;
        const float* rp = p.R + ((size_t)img * KM + 64 * w + 4 * half) * p.HW + pix0 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][r] = (ABL & 1) ? 0.f : rp[(size_t)(32 * i + row_of(r)) * p.HW];
        if (!(ABL & 8))
        {
            const float* xp = p.X + ((size_t)img * C1 + xk) * p.HW + pix0 + 4 * xc4;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(xp), v1 = *reinterpret_cast<const f32x4*>(xp + (size_t)32 * p.HW);
            *reinterpret_cast<f32x4*>(&Xs[xk * BN + 4 * xc4]) = v0;
            *reinterpret_cast<f32x4*>(&Xs[(xk + 32) * BN + 4 * xc4]) = v1;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][r] += b1s[64 * w + 32 * i + row_of(r) + 4 * half];
    };
    auto epilogue = [&](int q) { // T out of the accumulators, the four partial Y tiles summed in wave order, Y out
        const int t = TILE_OF(q), img = t / p.tiles_per_image, pix0 = (t - img * p.tiles_per_image) * BN;
        if (!(ABL & 2))
        {
            float* tp = p.T + ((size_t)img * KM + 64 * w + 4 * half) * p.HW + pix0 + l31;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) tp[(size_t)(32 * i + row_of(r)) * p.HW] = acc1[i][r];
        }
        if (!(ABL & 4))
        {
            float* yp = p.Y + (size_t)img * K2 * p.HW + pix0;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
            {
                const int idx = t8 + 256 * qq, row = idx >> 3, c4 = idx & 7;
                f32x4 sum = *reinterpret_cast<const f32x4*>(&red[(0 * K2 + row) * BN + 4 * c4]);
#pragma unroll
                for (int ww = 1; ww < 4; ++ww) sum += *reinterpret_cast<const f32x4*>(&red[(ww * K2 + row) * BN + 4 * c4]);
                const float bb = b2s[row];
                sum.x = fmaxf(sum.x + bb, 0.f);
                sum.y = fmaxf(sum.y + bb, 0.f);
                sum.z = fmaxf(sum.z + bb, 0.f);
                sum.w = fmaxf(sum.w + bb, 0.f);
                *reinterpret_cast<f32x4*>(yp + (size_t)row * p.HW + 4 * c4) = sum;
            }
        }
    };
    auto compute = [&]() {
        const float* xs = Xs + half * BN + l31;
#pragma unroll
        for (int kp = 0; kp < 32; ++kp)
        {
            const float b = xs[2 * kp * BN];
            acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0][kp], b, acc1[0], 0, 0, 0);
            acc1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1][kp], b, acc1[1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][r] = fmaxf(acc1[i][r], 0.f);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i2][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j)
        {
            const float b = acc1[j >> 4][j & 15];
            acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[0][j], b, acc2[0], 0, 0, 0);
            acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[1][j], b, acc2[1], 0, 0, 0);
        }
        if (!(ABL & 4))
        {
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(w * K2 + 32 * i2 + row_of(r) + 4 * half) * BN + l31] = acc2[i2][r];
        }
    };

    // WIDE memory phase: next tile's requests first (they have the latency), then the finished tile's T / Y, then the arrivals
    auto memory_wide = [&](int prev, int next) {
        f32x4 rv[8], xv0, xv1;
        int imgn = 0, pixn = 0;
        if (next >= 0)
        {
            const int t = TILE_OF(next);
            imgn = t / p.tiles_per_image;
            pixn = (t - imgn * p.tiles_per_image) * BN;
            const float* rp = p.R + ((size_t)imgn * KM + 64 * w + e_row) * p.HW + pixn + e_c4;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) rv[i * 4 + qd] = (ABL & 1) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(rp + (size_t)(32 * i + 8 * qd) * p.HW);
            const float* xp = p.X + ((size_t)imgn * C1 + xk) * p.HW + pixn + 4 * xc4;
            xv0 = (ABL & 8) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(xp);
            xv1 = (ABL & 8) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(xp + (size_t)32 * p.HW);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (prev >= 0)
        {
            const int t = TILE_OF(prev), img = t / p.tiles_per_image, pix0 = (t - img * p.tiles_per_image) * BN;
            if (!(ABL & 2))
            {
                float* tp = p.T + ((size_t)img * KM + 64 * w + e_row) * p.HW + pix0 + e_c4;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                {
#pragma unroll
                    for (int r = 0; r < 16; ++r) scr[(row_of(r) + 4 * half) * 36 + l31] = acc1[i][r];
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
                        *reinterpret_cast<f32x4*>(tp + (size_t)(32 * i + 8 * qd) * p.HW) = *reinterpret_cast<const f32x4*>(&scr[(8 * qd + e_row) * 36 + e_c4]);
                }
            }
            if (!(ABL & 4))
            {
                float* yp = p.Y + (size_t)img * K2 * p.HW + pix0;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
                {
                    const int idx = t8 + 256 * qq, row = idx >> 3, c4 = idx & 7;
                    f32x4 sum = *reinterpret_cast<const f32x4*>(&red[(0 * K2 + row) * BN + 4 * c4]);
#pragma unroll
                    for (int ww = 1; ww < 4; ++ww) sum += *reinterpret_cast<const f32x4*>(&red[(ww * K2 + row) * BN + 4 * c4]);
                    const float bb = b2s[row];
                    sum.x = fmaxf(sum.x + bb, 0.f);
                    sum.y = fmaxf(sum.y + bb, 0.f);
                    sum.z = fmaxf(sum.z + bb, 0.f);
                    sum.w = fmaxf(sum.w + bb, 0.f);
                    *reinterpret_cast<f32x4*>(yp + (size_t)row * p.HW + 4 * c4) = sum;
                }
            }
        }
        if (next >= 0)
        {
#pragma unroll
            for (int i = 0; i < 2; ++i)
            {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<f32x4*>(&scr[(8 * qd + e_row) * 36 + e_c4]) = rv[i * 4 + qd];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[i][r] = scr[(row_of(r) + 4 * half) * 36 + l31] + b1s[64 * w + 32 * i + row_of(r) + 4 * half];
            }
            *reinterpret_cast<f32x4*>(&Xs[xk * BN + 4 * xc4]) = xv0;
            *reinterpret_cast<f32x4*>(&Xs[(xk + 32) * BN + 4 * xc4]) = xv1;
        }
    };

    if (g == 0 && n > 0) prologue(0);
    __syncthreads();
    for (int s = 0; s <= n; ++s)
    {
        if ((s & 1) == g)
        {
            if (s < n) compute();
        }
        else if (WIDE)
            memory_wide(s >= 1 ? s - 1 : -1, s + 1 < n ? s + 1 : -1);
        else
        {
            if (s >= 1) epilogue(s - 1);
            if (s + 1 < n) prologue(s + 1);
        }
        // the hand-over between the groups is LDS only (X tile, partial Y): wait for the LDS queue, not for the global stores in flight
        // (__syncthreads() = workgroup fence + barrier makes hipcc drain vmcnt too: every phase would end on the T / Y store acknowledgements)
        if (WIDE) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
    }
}

static hipEvent_t g_a, g_b;
template <class F>
static double time_us(F&& f, int reps)
{
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(g_a, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(g_b, 0));
    CK(hipEventSynchronize(g_b));
    float ms;
    CK(hipEventElapsedTime(&ms, g_a, g_b));
    return ms / reps * 1e3;
}

static void fill_random(float* d, size_t n, unsigned seed, float scale)
{
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (auto& x : h)
    {
        s = s * 1664525u + 1013904223u;
        x = ((s >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f) * scale;
    }
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
}

static fhip_conv_param conv1x1(int c, int k, int h)
{
    fhip_conv_param p;
    memset(&p, 0, sizeof p);
    p.input_channels = c;
    p.output_channels = k;
    p.input_h = p.input_w = h;
    p.kernel_h = p.kernel_w = 1;
    p.stride_h = p.stride_w = 1;
    p.group = 1;
    p.bias_term = 1;
    p.activation = FHIP_ACT_RELU;
    CF(fhip_conv_assign_output_dim(&p));
    return p;
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 30;
    const int N = 64, H = 56, HW = H * H;
    CK(hipEventCreate(&g_a));
    CK(hipEventCreate(&g_b));
    int cus = 0;
    {
        char nm[64];
        int lds;
        CF(fhip_device_info(nm, 64, &cus, &lds));
    }
    const size_t nx = (size_t)N * C1 * HW, nt = (size_t)N * KM * HW, ny = (size_t)N * K2 * HW;
    float *X, *R, *T, *Y, *Tref, *Yref, *W1, *b1, *W2, *b2;
    CK(hipMalloc(&X, nx * 4));
    CK(hipMalloc(&R, nt * 4));
    CK(hipMalloc(&T, nt * 4));
    CK(hipMalloc(&Tref, nt * 4));
    CK(hipMalloc(&Y, ny * 4));
    CK(hipMalloc(&Yref, ny * 4));
    CK(hipMalloc(&W1, KM * C1 * 4));
    CK(hipMalloc(&W2, K2 * KM * 4));
    CK(hipMalloc(&b1, KM * 4));
    CK(hipMalloc(&b2, K2 * 4));
    fill_random(X, nx, 1, 1.f);
    fill_random(R, nt, 2, 1.f);
    fill_random(W1, KM * C1, 3, 0.125f);
    fill_random(W2, K2 * KM, 4, 0.0625f);
    fill_random(b1, KM, 5, 0.1f);
    fill_random(b2, K2, 6, 0.1f);

    // ---- the product's two launches
    fhip_conv_param pe = conv1x1(C1, KM, H), pr = conv1x1(KM, K2, H);
    size_t be, ke, br, kr;
    CF(fhip_conv_get_buffer_size(&pe, FHIP_IM2COL, N, &be, &ke));
    CF(fhip_conv_get_buffer_size(&pr, FHIP_IM2COL, N, &br, &kr));
    float *pke, *pkr, *buf;
    CK(hipMalloc(&pke, ke));
    CK(hipMalloc(&pkr, kr));
    CK(hipMalloc(&buf, std::max<size_t>(std::max(be, br), 256)));
    CF(fhip_conv_init(&pe, FHIP_IM2COL, pke, W1, 0));
    CF(fhip_conv_init(&pr, FHIP_IM2COL, pkr, W2, 0));
    auto two = [&] {
        CF(fhip_conv_forward_residual(&pe, FHIP_IM2COL, N, Tref, X, pke, buf, b1, R, 0));
        CF(fhip_conv_forward(&pr, FHIP_IM2COL, N, Yref, Tref, pkr, buf, b2, 0));
    };
    auto expand = [&] { CF(fhip_conv_forward_residual(&pe, FHIP_IM2COL, N, Tref, X, pke, buf, b1, R, 0)); };
    auto reduce = [&] { CF(fhip_conv_forward(&pr, FHIP_IM2COL, N, Yref, Tref, pkr, buf, b2, 0)); };
    two();
    CK(hipDeviceSynchronize());

    B2bParams q{X, R, W1, b1, W2, b2, T, Y, HW, N * HW / BN, HW / BN};
    const int grid = std::min(q.tiles, 2 * cus);
    auto fused = [&] { hipLaunchKernelGGL(b2b_kernel<0>, dim3(grid), dim3(256), 0, 0, q); };
    CK(hipMemset(T, 0xff, nt * 4));
    CK(hipMemset(Y, 0xff, ny * 4));
    fused();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    {
        std::vector<float> a(nt), b(nt);
        CK(hipMemcpy(a.data(), T, nt * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), Tref, nt * 4, hipMemcpyDeviceToHost));
        double e = 0, m = 0;
        for (size_t i = 0; i < nt; ++i)
        {
            e = std::max(e, (double)std::fabs(a[i] - b[i]));
            m = std::max(m, (double)std::fabs(b[i]));
        }
        printf("T: max |fused - product| / max |product| = %.3g\n", e / m);
        a.resize(ny);
        b.resize(ny);
        CK(hipMemcpy(a.data(), Y, ny * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), Yref, ny * 4, hipMemcpyDeviceToHost));
        e = m = 0;
        for (size_t i = 0; i < ny; ++i)
        {
            e = std::max(e, (double)std::fabs(a[i] - b[i]));
            m = std::max(m, (double)std::fabs(b[i]));
        }
        printf("Y: max |fused - product| / max |product| = %.3g\n", e / m);
    }
    const double gf = 2.0 * 2.0 * KM * C1 * (double)N * HW / 1e9, mb_fused = (nx + 2 * nt + ny) * 4 / 1e6, mb_two = (nx + 3 * nt + ny) * 4 / 1e6;
    for (int round = 0; round < 3; ++round)
    {
        const double t2 = time_us(two, reps), te = time_us(expand, reps), tr = time_us(reduce, reps), tf = time_us(fused, reps);
        printf("round %d: product expand %.1f us + reduce %.1f us; back to back %.1f us (%.2f TB/s);  fused %.1f us (%.2f TB/s, %.1f TF)\n", round, te, tr, t2,
               mb_two / t2, tf, mb_fused / tf, gf / tf * 1e3);
    }
    printf("fused ablations (us): no residual loads %.1f, no T stores %.1f, no Y path %.1f, no X loads %.1f, none of them %.1f\n",
           time_us([&] { hipLaunchKernelGGL(b2b_kernel<1>, dim3(grid), dim3(256), 0, 0, q); }, reps),
           time_us([&] { hipLaunchKernelGGL(b2b_kernel<2>, dim3(grid), dim3(256), 0, 0, q); }, reps),
           time_us([&] { hipLaunchKernelGGL(b2b_kernel<4>, dim3(grid), dim3(256), 0, 0, q); }, reps),
           time_us([&] { hipLaunchKernelGGL(b2b_kernel<8>, dim3(grid), dim3(256), 0, 0, q); }, reps),
           time_us([&] { hipLaunchKernelGGL(b2b_kernel<15>, dim3(grid), dim3(256), 0, 0, q); }, reps));
    for (int g : {cus, 2 * cus, 3 * cus, 4 * cus})
    {
        const int gg = std::min(q.tiles, g);
        printf("grid %d: %.1f us\n", gg, time_us([&] { hipLaunchKernelGGL(b2b_kernel<0>, dim3(gg), dim3(256), 0, 0, q); }, reps));
    }
    // ---- ping-pong form, one 512-thread block per CU
    {
        const int gp = std::min(q.tiles, cus);
        CK(hipMemset(T, 0xff, nt * 4));
        CK(hipMemset(Y, 0xff, ny * 4));
        hipLaunchKernelGGL(b2b_pp_kernel<0>, dim3(gp), dim3(512), 0, 0, q);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        std::vector<float> a(nt), b(nt);
        CK(hipMemcpy(a.data(), T, nt * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), Tref, nt * 4, hipMemcpyDeviceToHost));
        double e = 0, m = 0;
        for (size_t i = 0; i < nt; ++i) e = std::max(e, (double)std::fabs(a[i] - b[i])), m = std::max(m, (double)std::fabs(b[i]));
        printf("ping-pong T: %.3g", e / m);
        a.resize(ny);
        b.resize(ny);
        CK(hipMemcpy(a.data(), Y, ny * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), Yref, ny * 4, hipMemcpyDeviceToHost));
        e = m = 0;
        for (size_t i = 0; i < ny; ++i) e = std::max(e, (double)std::fabs(a[i] - b[i])), m = std::max(m, (double)std::fabs(b[i]));
        printf("  Y: %.3g (normalised max error vs the product)\n", e / m);
        for (int round = 0; round < 3; ++round)
        {
            const double t2 = time_us(two, reps), tf = time_us([&] { hipLaunchKernelGGL(b2b_pp_kernel<0>, dim3(gp), dim3(512), 0, 0, q); }, reps);
            printf("round %d: product back to back %.1f us;  ping-pong fused %.1f us (%.2f TB/s, %.1f TF)\n", round, t2, tf, mb_fused / tf, gf / tf * 1e3);
        }
        CK(hipMemset(T, 0xff, nt * 4));
        CK(hipMemset(Y, 0xff, ny * 4));
        hipLaunchKernelGGL((b2b_pp_kernel<0, true>), dim3(gp), dim3(512), 0, 0, q);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        a.resize(nt);
        b.resize(nt);
        CK(hipMemcpy(a.data(), T, nt * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), Tref, nt * 4, hipMemcpyDeviceToHost));
        e = m = 0;
        for (size_t i = 0; i < nt; ++i) e = std::max(e, (double)std::fabs(a[i] - b[i])), m = std::max(m, (double)std::fabs(b[i]));
        printf("ping-pong WIDE T: %.3g", e / m);
        a.resize(ny);
        b.resize(ny);
        CK(hipMemcpy(a.data(), Y, ny * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), Yref, ny * 4, hipMemcpyDeviceToHost));
        e = m = 0;
        for (size_t i = 0; i < ny; ++i) e = std::max(e, (double)std::fabs(a[i] - b[i])), m = std::max(m, (double)std::fabs(b[i]));
        printf("  Y: %.3g\n", e / m);
        for (int round = 0; round < 3; ++round)
        {
            const double t2 = time_us(two, reps), tf = time_us([&] { hipLaunchKernelGGL((b2b_pp_kernel<0, true>), dim3(gp), dim3(512), 0, 0, q); }, reps);
            printf("round %d: product back to back %.1f us;  ping-pong WIDE fused %.1f us (%.2f TB/s, %.1f TF)\n", round, t2, tf, mb_fused / tf, gf / tf * 1e3);
        }
        printf("ping-pong WIDE ablations (us): no residual loads %.1f, no T stores %.1f, no X loads %.1f, no R/T/X %.1f\n",
               time_us([&] { hipLaunchKernelGGL((b2b_pp_kernel<1, true>), dim3(gp), dim3(512), 0, 0, q); }, reps),
               time_us([&] { hipLaunchKernelGGL((b2b_pp_kernel<2, true>), dim3(gp), dim3(512), 0, 0, q); }, reps),
               time_us([&] { hipLaunchKernelGGL((b2b_pp_kernel<8, true>), dim3(gp), dim3(512), 0, 0, q); }, reps),
               time_us([&] { hipLaunchKernelGGL((b2b_pp_kernel<11, true>), dim3(gp), dim3(512), 0, 0, q); }, reps));
        printf("ping-pong ablations (us): no residual loads %.1f, no T stores %.1f, no Y path %.1f, no X loads %.1f, no memory at all %.1f\n",
               time_us([&] { hipLaunchKernelGGL(b2b_pp_kernel<1>, dim3(gp), dim3(512), 0, 0, q); }, reps),
               time_us([&] { hipLaunchKernelGGL(b2b_pp_kernel<2>, dim3(gp), dim3(512), 0, 0, q); }, reps),
               time_us([&] { hipLaunchKernelGGL(b2b_pp_kernel<4>, dim3(gp), dim3(512), 0, 0, q); }, reps),
               time_us([&] { hipLaunchKernelGGL(b2b_pp_kernel<8>, dim3(gp), dim3(512), 0, 0, q); }, reps),
               time_us([&] { hipLaunchKernelGGL(b2b_pp_kernel<11>, dim3(gp), dim3(512), 0, 0, q); }, reps));
    }
    return 0;
}
