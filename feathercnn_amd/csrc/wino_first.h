// wino_first.h -- a net's FIRST convolution (2 .. 4 input channels, 3x3 / stride 1 / pad 1: VGG-16's conv1_1) computed inside the input
// transform of the Winograd layer that consumes it (round 3).  Included by winograd_f63.hip (bt8, the V layout).
//
// VGG-16 b32: conv1_1 writes 64 x 224 x 224 x 32 floats = 411 MB which conv1_2's input transform (K2) reads back -- 0.135 + 0.236 ms of
// the 3.56 ms step for 5.5 GFLOP of arithmetic.  The image itself is 19 MB.  Here K2's lane -- one 8x8 window of one channel k of the
// activation -- computes its 64 activation values itself from the 10 x 10 x Cin image patch under the window (27 wave-uniform weights in
// scalar registers; 1728 FMAs, issued as packed pairs) instead of loading them: (8/6)^2 = 1.78 x the first layer's arithmetic on the vector
// ALUs, and the activation tensor never exists.  Same per-value formula as the first layer on its own (bias + sum over (c, u, v), ReLU),
// summed in a fixed (c, row, u, v) order; cells of the window outside the image are the CONSUMER's zero padding, not first-layer outputs.
//
// Image reads are raw buffer loads: a row or column of the patch that lies outside the image is given an offset beyond the buffer's
// num_records, for which the hardware returns 0 -- the first layer's zero padding costs no select and no branch.
#pragma once

#include "common.h"
#include "wino_layout.h"

namespace fhip
{

typedef float f32x2 __attribute__((ext_vector_type(2)));


struct WinoFirstParams
{
    const float* in;   // [N][CIN][H][W]: the first layer's input
    const float* w;    // [K][CIN][3][3]: its filters as loaded (BatchNorm folded), K = the consumer's input channels
    const float* bias; // [K] or nullptr
    float* V;          // the consumer's V[64][K][Pp]
    int K, H, W;       // first layer: H x W in and out
    int TX, T, P, Pp;  // the consumer's tiling
    unsigned in_bytes; // < 2^30 (the out-of-range encoding below)
    unsigned v_bytes;  // all of V, for the shared-column form's buffer stores (<= 2^31, else that form is not used)
    int relu;
    int N, bpi;        // staged form: images, blocks per image (64 tiles each)
    int LDW, rows;     // staged form: LDS row pitch 6 TX + 4 and the most patch rows a block stages
    WinoLayout Lv;     // where the consumer's V (rows = K) lives
};

constexpr unsigned kWinoFirstOob = 0x40000000u; // + any in-range offset (and + itself) is still >= num_records

__device__ __forceinline__ f32x2 first_load2(__amdgpu_buffer_rsrc_t rsrc, unsigned off)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, 0, 0);
    f32x2 r;
    r.x = __uint_as_float(v.x);
    r.y = __uint_as_float(v.y);
    return r;
}

template <int CIN>
__global__ __launch_bounds__(256, 4) void wino_input_from_first_kernel(const WinoFirstParams q)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y; // uniform: the weights below are scalar loads
    if (p >= q.P) return;
    const int n = p / q.T, t = p - n * q.T;
    const int ty = t / q.TX, tx = t - ty * q.TX;
    const int y0 = 6 * ty - 1, x0 = 6 * tx - 1; // the window's origin in the activation (consumer pad 1)

    float w[CIN][3][3];
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) w[c][u][v] = q.w[((size_t)k * CIN + c) * 9 + u * 3 + v];
    const float b = q.bias ? q.bias[k] : 0.f;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.in), 0, (int)q.in_bytes, 0x00020000);
    // patch columns x0 - 1 + 2 m, m = 0 .. 4 (x0 - 1 = 6 tx - 2 is even and W is even: a float2 is inside or outside as a whole)
    unsigned coloff[5];
#pragma unroll
    for (int m = 0; m < 5; ++m)
    {
        const int xx = x0 - 1 + 2 * m;
        coloff[m] = ((unsigned)xx < (unsigned)q.W) ? (unsigned)xx * 4u : kWinoFirstOob;
    }

    f32x2 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x2){b, b};

    const unsigned row_bytes = (unsigned)q.W * 4u;
#pragma unroll
    for (int c = 0; c < CIN; ++c)
    {
        const unsigned plane = (unsigned)((n * CIN + c) * q.H) * row_bytes;
#pragma unroll
        for (int r = 0; r < 10; ++r)
        {
            const int yy = y0 - 1 + r;
            const unsigned rowoff = ((unsigned)yy < (unsigned)q.H) ? plane + (unsigned)yy * row_bytes : kWinoFirstOob;
            float row[10];
#pragma unroll
            for (int m = 0; m < 5; ++m)
            {
                const f32x2 d = first_load2(rsrc, rowoff + coloff[m]);
                row[2 * m] = d.x;
                row[2 * m + 1] = d.y;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
            {
                const int i = r - u; // activation row of the window this patch row feeds through filter row u
                if (i < 0 || i > 7) continue;
#pragma unroll
                for (int v = 0; v < 3; ++v)
                {
                    const f32x2 wv = (f32x2){w[c][u][v], w[c][u][v]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_elementwise_fma(wv, (f32x2){row[2 * j + v], row[2 * j + v + 1]}, acc[i][j]);
                }
            }
        }
    }

    // activation + the consumer's zero padding
    const float lo = q.relu ? 0.f : -__builtin_huge_valf();
    bool colok[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) colok[j] = (unsigned)(x0 + j) < (unsigned)q.W;
    float d[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
        const bool rowok = (unsigned)(y0 + i) < (unsigned)q.H;
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            const float a = (j & 1) ? acc[i][j >> 1].y : acc[i][j >> 1].x;
            d[i][j] = (rowok && colok[j]) ? fmaxf(a, lo) : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) bt8(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], d[6][j], d[7][j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) bt8(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5], d[i][6], d[i][7]);

    // uniform base (scalar registers, advanced on the scalar unit) + one 32-bit lane offset: no 64-bit vector adds per store
    // uniform per-xi base (scalar registers) + one lane offset: no 64-bit vector adds per store
    const size_t xi_stride = q.Lv.xis;
    float* const vb = q.V + (size_t)k * q.Lv.bp;
    const size_t lane_off = q.Lv.col(p);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) st_scratch(vb + (size_t)(i * 8 + j) * xi_stride + lane_off, d[i][j]);
}


// ---- staged form (the one normally used) -----------------------------------------------------------------------------------------
// The direct form above re-reads the image through the L1 once per output channel: 150 float2 loads per lane at a 24-byte lane stride
// touch 12 cache lines each, 10.6 GB of line traffic for VGG-16 b32 -- the kernel ran at the L1's rate (0.39 ms), not the ALUs'.  Here a
// block owns 64 consecutive tiles of ONE image (lane = tile, so V stores stay 256-byte runs) and kFirstCpb output channels (a wave takes
// every fourth of them: its channel, hence its 27 weights, is wave-uniform): the image rows under those tiles -- at most three tile rows
// at 224 px: 22 rows x 232 floats x Cin = 61 KB -- are staged in LDS once, zero-filled outside the image (both paddings), and every
// (lane, channel) reads its 10 x 10 x Cin patch from there (ds_read2_b64, all 50 of an image channel requested before its first FMA).
// Same arithmetic, same order.  Measured, VGG-16 b32 (tools/first_bench.py, one box): conv1_1 173 us + input transform 257 us on their own,
// 267 us fused; in the net conv1_1 + conv1_2 0.881 -> 0.805 ms.  The kernel is vector-ALU bound: 2.5 k VALU instructions per (tile,
// channel) -- 1728 FMAs + the two butterflies -- against a 2.4 k floor, the SIMDs issue VALU 64 % of the launch (PMC: SQ_ACTIVE_INST_VALU),
// v_pk_fma_f32 pairs ran no faster than scalar v_fmac, and neither 8 waves per block nor hoisting / sinking the LDS reads moves it.
#ifndef FHIP_FIRST_ORDER
#define FHIP_FIRST_ORDER 0 // block order of wino_input_from_first_staged_kernel: 0 channel group slowest, 1 tile block slowest (measurement switch)
#endif
constexpr int kFirstCpb = 16;   // output channels per block, 4 per wave (tools/first_bench.hip: 16 / 32 / 64 -> 322 / 332 / 358 us)
constexpr int kFirstStage = 12; // image loads a thread keeps in flight while staging
constexpr int kFirstWaves = 4;  // waves per block (8 waves at 128 registers spill: 392 us)
constexpr bool kFirstHoist = true;
constexpr bool kFirstXcd = true;
constexpr int kFirstTiles = 64; // tiles per block

//
// SHARE (round 4): horizontally adjacent windows overlap in two columns -- a lane's columns 6, 7 are its right neighbour's columns 0, 1, the same
// first-layer values behind the same row clamps.  A lane computes columns 0 .. 5 only (1296 FMAs instead of 1728, 40 LDS reads per image channel
// instead of 50) and takes the other two from lane + 1 (16 ds_bpermute per channel).  The row-end tile has no right neighbour and needs none
// when its columns 6, 7 lie outside the image (6 TX - 1 >= W: the launcher's condition -- they are the consumer's zero padding); the wave-end
// lane has none either, so a block owns 63 tiles and lane 63 computes the 64th for its neighbour's sake only (no stores; ceil(T / 63) blocks per
// image: 23 either way at 224 px).  Same values, same order of summation.
template <int CIN, bool SHARE, int CPB = kFirstCpb, bool HOIST = kFirstHoist, int WAVES = kFirstWaves, bool XCD = kFirstXcd>
__global__ __launch_bounds__(64 * WAVES, WAVES / 2) void wino_input_from_first_staged_kernel(const WinoFirstParams q)
{
    constexpr int TPB = SHARE ? kFirstTiles - 1 : kFirstTiles; // tiles a block stores
    constexpr int NC = SHARE ? 6 : 8;                          // window columns a lane computes
    extern __shared__ __attribute__((aligned(16))) float smem[]; // [CIN][rows][LDW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // consecutive tile blocks on the same XCD: their 256-byte V runs share cache lines at both ends (p = n T + t is 16-byte aligned at
    // best), which only merge into whole-line writes inside one L2
    const int lin = XCD ? xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y) : blockIdx.y * gridDim.x + blockIdx.x;
#if FHIP_FIRST_ORDER == 1
    // tile blocks slowest, channel groups fastest: the gridDim.y blocks that read the SAME image rows are neighbours in an XCD's walk, so the rows come
    // over the fabric once and from that XCD's L2 gridDim.y - 1 times (round 6; with the channel group slowest an XCD walks one channel group over
    // ALL images: rocprofv3 counted 883 MB per launch for 776 algorithmic -- VGG-16 b32: the 19 MB image fetched gridDim.y = 4 times over plus halos)
    const int bx = lin / gridDim.y, by = lin - bx * gridDim.y;
#else
    const int bx = lin % gridDim.x, by = lin / gridDim.x;
#endif
    const int n = bx / q.bpi, b = bx - n * q.bpi;
    const int t0 = b * TPB, nt = min(TPB, q.T - t0);
    const int ty0 = t0 / q.TX, ty1 = (t0 + nt - 1) / q.TX;
    const int rows = 6 * (ty1 - ty0 + 1) + 4; // image rows 6 ty0 - 2 ... 6 ty1 + 7
    const int ry0 = 6 * ty0 - 2;
    // staging: LDS column xx holds image column xx - 2; a thread owns one float2 column of every second row vector (c, r); kFirstStage
    // loads are issued before the first of them is stored (a block lives ~30 us: sequential round trips would show)
    constexpr int RS = WAVES / 2; // row vectors staged side by side (128 threads each)
    const int half = q.LDW >> 1, cp = tid & 127, rsub = tid >> 7;
    const int nrow = CIN * rows, x = 2 * cp - 2;
    const bool xok = cp < half && (unsigned)x < (unsigned)q.W; // W even, x even: the pair is inside or outside as a whole
    const float* img = q.in + (size_t)n * CIN * q.H * q.W;
    for (int r0 = 0; r0 < nrow; r0 += RS * kFirstStage)
    {
        f32x2 v[kFirstStage];
#pragma unroll
        for (int u = 0; u < kFirstStage; ++u)
        {
            const int rv = r0 + RS * u + rsub;
            const int c = (rv >= rows) + (rv >= 2 * rows) + (rv >= 3 * rows), y = ry0 + rv - c * rows;
            v[u] = (f32x2){0.f, 0.f};
            if (rv < nrow && xok && (unsigned)y < (unsigned)q.H) v[u] = *reinterpret_cast<const f32x2*>(img + ((size_t)c * q.H + y) * q.W + x);
        }
#pragma unroll
        for (int u = 0; u < kFirstStage; ++u)
        {
            const int rv = r0 + RS * u + rsub;
            const int c = (rv >= rows) + (rv >= 2 * rows) + (rv >= 3 * rows);
            if (rv < nrow && cp < half) *reinterpret_cast<f32x2*>(smem + ((size_t)c * q.rows + (rv - c * rows)) * q.LDW + 2 * cp) = v[u];
        }
    }
    __syncthreads();
    // SHARE: lane 63 of a full block is the helper -- the tile after the block's last one if that is its right neighbour, else (next tile row,
    // or none: the last tile's columns 6, 7 are padding) a copy of the last one whose values nobody takes
    const bool helper = SHARE && lane == TPB && nt == TPB;
    if (lane >= nt && !helper) return;
    int t = t0 + lane;
    if (helper && (t >= q.T || t % q.TX == 0)) t -= 1;
    const int ty = t / q.TX, tx = t - ty * q.TX;
    const int y0 = 6 * ty - 1, x0 = 6 * tx - 1;
    const float* patch = smem + (size_t)(6 * (ty - ty0)) * q.LDW + 6 * tx; // image (6 ty - 2, 6 tx - 2)
    const size_t xi_stride = q.Lv.xis;
    // the lane's column: its block, relative to the block of the wave's first column (a wave's 64 columns touch at most two blocks), so the
    // per-store address stays a wave-uniform base + one 32-bit lane offset
    const int p_lane = n * q.T + t, p_first = __builtin_amdgcn_readfirstlane(n * q.T + t0);
    const size_t col_first = q.Lv.col(p_first);
    const unsigned lane_off = (unsigned)(q.Lv.col(p_lane) - col_first);
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(q.V, 0, (int)q.v_bytes, 0x00020000);
    const unsigned voff = helper ? 0x80000000u : lane_off * 4u;
    const float lo = q.relu ? 0.f : -__builtin_huge_valf();
    // activation + the consumer's zero padding in two instructions per value: clamp(a, lo_i, hi_i) with [lo_i, hi_i] = [lo, inf) on rows of the
    // window inside the image and [0, 0] outside, then a select on the column (64 precomputed (row, column) lane masks would not fit the
    // scalar registers: hipcc spilled them and paid two v_readlane per value)
    bool colok[8];
    float row_lo[8], row_hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
    {
        colok[j] = (unsigned)(x0 + j) < (unsigned)q.W;
        const bool rowok = (unsigned)(y0 + j) < (unsigned)q.H;
        row_lo[j] = rowok ? lo : 0.f;
        row_hi[j] = rowok ? __builtin_huge_valf() : 0.f;
    }
    const int kbase = by * CPB;
    for (int kk = wave; kk < CPB; kk += WAVES)
    {
        const int k = __builtin_amdgcn_readfirstlane(kbase + kk);
        if (k >= q.K) break;
        float w[CIN][3][3];
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v) w[c][u][v] = q.w[((size_t)k * CIN + c) * 9 + u * 3 + v];
        const float bias = q.bias ? q.bias[k] : 0.f;
        float acc[8][NC];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NC; ++j) acc[i][j] = bias;
#pragma unroll
        for (int c = 0; c < CIN; ++c)
        {
            // the whole 10 x 10 patch of this image channel is requested before its first FMA (100 registers; 2 waves per SIMD leave 256):
            // one LDS round trip per image channel instead of one per patch row
            const float* pc = patch + (size_t)c * q.rows * q.LDW;
            if (SHARE) __builtin_amdgcn_sched_barrier(0); // ... and would hoist the next channel's reads over this one's FMAs (spills)
            f32x2 pr[10][NC / 2 + 1];
#pragma unroll
            for (int r = 0; r < 10; ++r)
#pragma unroll
                for (int m = 0; m < NC / 2 + 1; ++m) pr[r][m] = *reinterpret_cast<const f32x2*>(pc + (size_t)r * q.LDW + 2 * m);
            if (HOIST) __builtin_amdgcn_sched_barrier(0); // hipcc otherwise sinks the reads to their uses: 76 waits per channel instead of a counted few
#pragma unroll
            for (int r = 0; r < 10; ++r)
            {
                float row[NC + 2];
#pragma unroll
                for (int m = 0; m < NC / 2 + 1; ++m)
                {
                    row[2 * m] = pr[r][m].x;
                    row[2 * m + 1] = pr[r][m].y;
                }
#pragma unroll
                for (int u = 0; u < 3; ++u)
                {
                    const int i = r - u;
                    if (i < 0 || i > 7) continue;
#pragma unroll
                    for (int v = 0; v < 3; ++v)
#pragma unroll
                        for (int j = 0; j < NC; ++j) acc[i][j] = fmaf(w[c][u][v], row[j + v], acc[i][j]);
                }
            }
        }
        float d[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NC; ++j)
            {
                d[i][j] = colok[j] ? __builtin_amdgcn_fmed3f(acc[i][j], row_lo[i], row_hi[i]) : 0.f;
            }
        if (SHARE)
        {
#pragma unroll
            for (int i = 0; i < 8; ++i)
            {
                const float n0 = __shfl_down(d[i][0], 1), n1 = __shfl_down(d[i][1], 1);
                d[i][6] = colok[6] ? n0 : 0.f;
                d[i][7] = colok[7] ? n1 : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) bt8(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], d[6][j], d[7][j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) bt8(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5], d[i][6], d[i][7]);
        // wave-uniform base (scalar registers) + the lane's tile index: no 64-bit vector address arithmetic per store
        if (SHARE)
        {
            // buffer stores (V is at most 2 GiB here): the helper's offset lies beyond num_records, so the hardware drops its stores -- a branch
            // around the 64 stores costs hipcc 57 more registers than the kernel has
            const unsigned sbase = (unsigned)(((size_t)k * q.Lv.bp + col_first) * sizeof(float));
            const unsigned sstep = (unsigned)(xi_stride * sizeof(float));
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d[i][j]), vrsrc, (int)voff, (int)(sbase + (unsigned)(i * 8 + j) * sstep), (FHIP_XFORM_NT & 2) ? 2 : 0);
        }
        else
        {
            float* vb = q.V + (size_t)k * q.Lv.bp + col_first;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) st_scratch(vb + (size_t)(i * 8 + j) * xi_stride + lane_off, d[i][j]);
        }
    }
}

} // namespace fhip
