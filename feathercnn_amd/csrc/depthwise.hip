// depthwise.hip -- depthwise convolution (group == input_channels) for gfx950.
// Replaces DEPTHWISE_GetBufferSize/Init/Forward (reference src/booster/avx/booster.cpp:121-160):
// pad_input (avx/generic_kernels.cpp:31-48) + dwConv_template<bias,relu> (avx/depthwise.cpp:161-207, incl. the
// global-kernel case :30-54).  Same arithmetic -- per-channel 2-D correlation, taps accumulated in (m, n) order
// in fp32, + bias[c], ReLU, floor output dims -- but no padded copy of the input: padding is a bounds check.
//
// This is an HBM-bound kernel (9 FMA per 8 bytes moved): the whole job is to move each input and output element once.
// Five forms live here, chosen by shape (measurements in DESIGN.md 3.3):
//   * depthwise3x3_flat_kernel    -- 3x3, stride 1/2, pad 1, 7 x 7 / 14 x 14 / 28 x 28 planes: a chunk of whole planes staged in LDS with
//     coalesced 16-byte loads, four consecutive outputs of the flat output stream per lane (7 x 7: one image row per lane) (round 3).
//   * depthwise3x3_band_kernel    -- 3x3, stride 1, pad 1, 112- and 56-pixel planes: a band of output rows of one plane per block, its
//     input rows staged in LDS the same way (round 3).
//   * depthwise3x3_direct_kernel  -- 3x3, stride 1/2, pad_left 1 (the MobileNet shapes).  NO LDS: every lane produces
//     a VX-wide x R-high output patch straight from global memory with aligned vector loads, takes its two halo taps
//     per row from the neighbouring lanes (cross-lane moves, not loads), and stores R vectors.
//   * depthwise_lds_scalar_kernel -- small planes of any other shape: a block copies a chunk of whole planes global->LDS with
//     coalesced 16-byte loads and computes one output per lane from LDS.  (The whole-plane LDS staging BASELINE.json names for the
//     3x3 case was built and measured at 24 % of HBM peak against 52 % for the direct form: tools/experiments)
//   * depthwise_generic_kernel -- any kernel size / stride / plane size, one output per lane (incl. the reference's
//     global-kernel special case).
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace fhip
{

struct DwParams
{
    const float* in;
    const float* w;   // dense [C][kh*kw]
    const float* w12; // 3x3 only: [C][12] (9 taps + 3 zeros), 16-byte aligned rows
    const float* bias;
    float* out;
    int C, H, W, OH, OW, KH, KW, SH, SW, PL, PT;
    int planes;           // N*C
    int planes_per_chunk; // fast path
    int has_bias, relu;
};

constexpr int kDwChunkFloats = 3136;   // plane data per block of the small-plane 3x3 kernel
constexpr int kDwChunkMaxPlane = 256;  // stride-2 planes up to 16 x 16 go there -- the one case it beats the direct kernel (tools/dw_bench.hip)
constexpr int kDwLdsFloats = 14336; // 56 KiB of plane data per block (+ weights) -> 2 blocks per CU (depthwise_lds_scalar_kernel)

// Direct 3x3 form: no LDS, every lane produces a VX-wide x R-high output patch straight from global memory.
// Per input row it issues the aligned VX-wide centre vector(s) plus the two halo scalars (which hit lines its
// neighbours fetch anyway); all (R*S+2) rows' loads are issued before the first FMA, the patch leaves as R
// VX-wide stores.  No barrier, <= 64 VGPRs, so 8 waves per SIMD hide the HBM latency the way a plain copy
// kernel does; the vertical halo rows are re-read through L1/L2, not HBM.
template <int VX>
struct DwVec;
template <>
struct DwVec<4>
{
    typedef float4 type;
    static __device__ void unpack(const float4& v, float* o)
    {
        o[0] = v.x;
        o[1] = v.y;
        o[2] = v.z;
        o[3] = v.w;
    }
    static __device__ float4 pack(const float* o) { return make_float4(o[0], o[1], o[2], o[3]); }
};
template <>
struct DwVec<2>
{
    typedef float2 type;
    static __device__ void unpack(const float2& v, float* o)
    {
        o[0] = v.x;
        o[1] = v.y;
    }
    static __device__ float2 pack(const float* o) { return make_float2(o[0], o[1]); }
};
template <>
struct DwVec<1>
{
    typedef float type;
    static __device__ void unpack(const float& v, float* o) { o[0] = v; }
    static __device__ float pack(const float* o) { return o[0]; }
};

template <int S, int VX, int R>
__global__ __launch_bounds__(256) void depthwise3x3_direct_kernel(const DwParams q, int yblocks, int xvecs, long long total)
{
    typedef typename DwVec<VX>::type vec_t;
    constexpr int ROWS = (R - 1) * S + 3; // input rows of the patch
    constexpr int SPAN = (VX - 1) * S + 3; // input columns of the patch: xb-1 .. xb+SPAN-2
    constexpr int NV = (SPAN - 2 + VX - 1) / VX; // aligned centre vectors covering xb .. xb+SPAN-3
    const int HW = q.H * q.W, OHW = q.OH * q.OW;
    const int lane = threadIdx.x & 63;
    // every lane of a wave runs the same number of iterations (the shuffles below need their neighbours): the
    // host rounds `total` work items up to whole waves and the tail lanes clamp to the last item without storing
    for (long long idx0 = (long long)blockIdx.x * 256 + threadIdx.x; idx0 - lane < total; idx0 += (long long)gridDim.x * 256)
    {
        const bool live = idx0 < total;
        const long long idx = live ? idx0 : total - 1;
        const int xq = (int)(idx % xvecs);
        const long long t = idx / xvecs;
        const int yb = (int)(t % yblocks);
        const int plane = (int)(t / yblocks);
        const int c = plane % q.C;
        const int ox0 = xq * VX, oy0 = yb * R;
        const int xb = ox0 * S; // centre starts here; the left halo tap is xb-1 (pad_left == 1)
        const float* ip = q.in + (size_t)plane * HW;

        float x[ROWS][NV * VX + 2]; // x[r][j] = input(row r, column xb - 1 + j)
        bool rok[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
        {
            const int y = oy0 * S - q.PT + r;
            rok[r] = (unsigned)y < (unsigned)q.H;
            const float* row = ip + (size_t)(rok[r] ? y : 0) * q.W;
#pragma unroll
            for (int v = 0; v < NV; ++v)
            {
                const int xc = xb + v * VX; // aligned; clamp whole vectors that start past the row
                const vec_t cv = *reinterpret_cast<const vec_t*>(row + min(xc, q.W - VX));
                DwVec<VX>::unpack(cv, &x[r][1 + v * VX]);
                if (xc >= q.W)
                {
#pragma unroll
                    for (int e = 0; e < VX; ++e) x[r][1 + v * VX + e] = 0.f;
                }
            }
            // halo taps: the left one is the neighbouring lane's last centre element, the right one its first (lanes
            // run along x inside a row block) -- a cross-lane move instead of two more VMEM instructions per row
            // (the CU's address pipe, not HBM, was the limit: 22 -> 10 VMEM instructions per 16 outputs).  Only the
            // wave's first / last lane has no neighbour and loads for real.
            // Valid only when the neighbours' centres are contiguous with ours (NV == S; not for VX = 1 at stride 2).
            float lft, rgt;
            if (NV == S)
            {
                lft = __shfl_up(x[r][NV * VX], 1);
                rgt = __shfl_down(x[r][1], 1);
                // A neighbouring lane is a neighbour in the IMAGE only inside one row block.  The first vector of a row never uses its
                // left tap (xb == 0: padding, pad_left == 1); the last vector's right tap is padding too when OW * S == W, but a REAL
                // pixel when pad_right < pad_left (OW * S < W) -- then the next lane belongs to another row and the tap is loaded.
                if (lane == 0) lft = row[max(xb - 1, 0)];
                if (lane == 63 || (xq == xvecs - 1 && xb + NV * VX < q.W)) rgt = row[min(xb + NV * VX, q.W - 1)];
            }
            else
            {
                lft = row[max(xb - 1, 0)];
                rgt = row[min(xb + NV * VX, q.W - 1)];
            }
            x[r][0] = lft;
            x[r][NV * VX + 1] = rgt;
        }
        // taps from the 12-float-per-channel copy Init appended to the packed weights: 3 VMEM instructions, not 9
        float w[12];
        {
            const float4* w4 = reinterpret_cast<const float4*>(q.w12 + (size_t)c * 12);
            const float4 a = w4[0], bq = w4[1], cq = w4[2];
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
            w[4] = bq.x; w[5] = bq.y; w[6] = bq.z; w[7] = bq.w;
            w[8] = cq.x; w[9] = cq.y; w[10] = cq.z; w[11] = cq.w;
        }
        const float b = q.has_bias ? q.bias[c] : 0.f;
        const bool lok = xb > 0;
        const bool rrok = xb + NV * VX < q.W;
        float* op = q.out + (size_t)plane * OHW + (size_t)oy0 * q.OW + ox0;
#pragma unroll
        for (int j = 0; j < R; ++j)
        {
            float acc[VX];
#pragma unroll
            for (int e = 0; e < VX; ++e) acc[e] = 0.f;
#pragma unroll
            for (int m = 0; m < 3; ++m)
            {
                const int r = j * S + m;
                if (!rok[r]) continue;
#pragma unroll
                for (int e = 0; e < VX; ++e)
#pragma unroll
                    for (int n = 0; n < 3; ++n)
                    {
                        const int jx = e * S + n; // column index into x[r][]
                        float v = x[r][jx];
                        if (jx == 0) v = lok ? v : 0.f;
                        if (jx == NV * VX + 1) v = rrok ? v : 0.f;
                        acc[e] += v * w[m * 3 + n];
                    }
            }
            if (live && oy0 + j < q.OH)
            {
                float o[VX];
#pragma unroll
                for (int e = 0; e < VX; ++e) o[e] = apply_act(acc[e] + b, q.relu);
                *reinterpret_cast<vec_t*>(op + (size_t)j * q.OW) = DwVec<VX>::pack(o);
            }
        }
    }
}

// 3x3 on SMALL planes (28x28 and below), stride 1 / 2, any pads.  The direct form above maps a lane to a VX x R patch: with 7 (28 px)
// or 3 (14 px, 8-byte vectors) lanes per image row its 16-byte loads touch 10-20 partially used cache lines per instruction and it
// ran at 34-53 % of the HBM peak there.  Here a block stages a small chunk of WHOLE planes -- consecutive planes are one contiguous
// run of memory, so the copy is fully coalesced 16-byte loads whatever W is -- computes every output from LDS and writes four
// consecutive outputs per lane as one 16-byte store (the output planes of a chunk are contiguous too).  The chunk is ~12 KB, not
// the 56 KB of the whole-plane staging measured in round 1 (24 %): ten blocks share a CU, so the load, compute and store phases
// of different blocks overlap.
template <int S>
__global__ __launch_bounds__(256) void depthwise3x3_chunk_kernel(const DwParams q, int chunk_planes, int tile_floats)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const tile = smem;             // [chunk_planes][H][W]
    float* const wl = smem + tile_floats; // [chunk_planes][12]: 9 taps, bias, 2 unused
    const int HW = q.H * q.W, OHW = q.OH * q.OW;
    const int tid = threadIdx.x;
    const int plane0 = blockIdx.x * chunk_planes;
    const int np = min(chunk_planes, q.planes - plane0);
    {
        // plane0 * HW is a multiple of 4 floats (the host picks chunk_planes % 4 == 0 unless HW % 4 == 0)
        const float* srcf = q.in + (size_t)plane0 * HW;
        const float4* src = reinterpret_cast<const float4*>(srcf);
        float4* dst = reinterpret_cast<float4*>(tile);
        const int nf = np * HW, n4 = nf >> 2;
        float4 v[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) v[b] = src[min(tid + b * 256, max(n4 - 1, 0))]; // every load in flight before the first LDS write
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (tid + b * 256 < n4) dst[tid + b * 256] = v[b];
        for (int i = tid + 1024; i < n4; i += 256) dst[i] = src[i];
        for (int i = (n4 << 2) + tid; i < nf; i += 256) tile[i] = srcf[i];
        for (int i = tid; i < np * 12; i += 256)
        {
            const int pl = i / 12, e = i - pl * 12;
            const int c = (plane0 + pl) % q.C;
            wl[i] = e < 9 ? q.w[c * 9 + e] : ((e == 9 && q.has_bias) ? q.bias[c] : 0.f);
        }
    }
    __syncthreads();
    const int nout = np * OHW;
    float* const obase = q.out + (size_t)plane0 * OHW;
    const bool vec = (OHW & 3) == 0; // then 4 consecutive outputs share a plane and the store is 16-byte aligned
    for (int o0 = tid * 4; o0 < nout; o0 += 1024)
    {
        float res[4];
        int pl = o0 / OHW;
        int r = o0 - pl * OHW;
        int oy = r / q.OW, ox = r - oy * q.OW;
#pragma unroll
        for (int e = 0; e < 4; ++e)
        {
            float acc = 0.f;
            if (o0 + e < nout)
            {
                const float* ip = tile + pl * HW;
                const float* wp = wl + pl * 12;
                const int iy0 = oy * S - q.PT, ix0 = ox * S - q.PL;
#pragma unroll
                for (int m = 0; m < 3; ++m)
                {
                    const int iy = iy0 + m;
                    if ((unsigned)iy >= (unsigned)q.H) continue;
                    const float* row = ip + iy * q.W;
#pragma unroll
                    for (int n = 0; n < 3; ++n)
                    {
                        const int ix = ix0 + n;
                        const float xv = row[min(max(ix, 0), q.W - 1)];
                        acc += (((unsigned)ix < (unsigned)q.W) ? xv : 0.f) * wp[m * 3 + n];
                    }
                }
                acc = apply_act(acc + wp[9], q.relu);
            }
            res[e] = acc;
            if (++ox == q.OW)
            {
                ox = 0;
                if (++oy == q.OH)
                {
                    oy = 0;
                    ++pl;
                }
            }
        }
        if (vec)
            stg4_act<4>(obase + o0, make_float4(res[0], res[1], res[2], res[3]));
        else
        {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (o0 + e < nout) obase[o0 + e] = res[e];
        }
    }
}

#ifndef FHIP_DW_FLAT_MINW
#define FHIP_DW_FLAT_MINW 1
#endif
// 3x3, pad 1 on every side, stride 1 / 2 on small SQUARE planes whose size is a compile-time constant (HH = 7, 14, 28: the last
// three stages of MobileNet-V1, 70 % of its depthwise bytes).  Round 3.  A tensor of such planes is ONE contiguous stream (a 14 x 14
// plane is 49 float4), so the block copies a chunk of whole planes to LDS with fully coalesced 16-byte loads -- every request of
// the chunk in flight before the first LDS write -- and every lane then produces FOUR CONSECUTIVE outputs of the flat output stream:
//   * stride 1, HW % 4 == 0: the input window of outputs f .. f+3 is three 6-float runs of the flat LDS image (f-W-1.., f-1..,
//     f+W-1..): 18 LDS dwords for 36 FMAs; row / column validity is a mask on the value, not on the address, so a group that wraps
//     from one image row into the next (14-pixel rows are 3.5 float4 long) needs no special case;
//   * otherwise (stride 2, or 7 x 7 planes whose 49 floats make groups straddle planes) each output decodes its own (plane, y, x)
//     with constant divisions and reads its 9 taps;
//   * the four results leave as one 16-byte store of the contiguous output stream.
// Against the direct kernel (lane = VX x R patch: 7 or 3 lanes per image row, 8-byte loads, three runtime integer divisions per
// item) and the generic chunk kernel above (per-output decode with runtime divisions, 18 LDS reads per output): tools/dw_bench.hip.
// the row-per-lane path of the flat kernel: small odd planes at stride 1 (a second LDS image holds the results)
constexpr bool dw_flat_row_path(int h, int stride) { return (h & 1) && h <= 9 && stride == 1; }

template <int HH, int S, int UNR>
__global__ __launch_bounds__(256, FHIP_DW_FLAT_MINW) void depthwise3x3_flat_kernel(const DwParams q, int chunk_planes, int chunks)
{
    constexpr int WW = HH, HW = HH * WW, OH = (HH - 1) / S + 1, OW = OH, OHW = OH * OW;
    constexpr int PAD = (WW + 1 + 3) / 4 * 4; // reads of masked taps reach WW + 1 floats before / WW + 4 after the chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tile_floats = chunk_planes * HW; // a multiple of 4 floats (host: chunk_planes % 4 == 0 unless HW % 4 == 0)
    float* const tile = smem + PAD;
    float* const wl = tile + tile_floats + PAD + 4; // [chunk_planes][12]: 9 taps, 3 unused
    float* const bl = wl + chunk_planes * 12;       // [chunk_planes] bias
    const int tid = threadIdx.x;

    // A block walks chunks blockIdx.x, + gridDim.x, ...; the requests of the NEXT chunk (planes, taps, bias) are issued before this
    // chunk's arithmetic, so a resident block always has loads in flight (a block that loads, computes and stores one chunk in
    // sequence leaves the memory pipe idle for two thirds of its life).
    f32x4 v[UNR], wv; // (native vectors: an array of HIP float4 structs carried around the chunk loop is not promoted to registers)
    float bv;
    // (a macro, not a lambda: a by-reference capture of v[] sends the array to scratch memory)
#define FHIP_DW_REQUEST(chunk_)                                                                                   \
    {                                                                                                             \
        const int rq_plane0 = (chunk_)*chunk_planes;                                                              \
        const int rq_np = min(chunk_planes, q.planes - rq_plane0);                                                \
        const f32x4* rq_src = reinterpret_cast<const f32x4*>(q.in + (size_t)rq_plane0 * HW); /* 16-B aligned */  \
        const int rq_n4 = (rq_np * HW) >> 2;                                                                      \
        _Pragma("unroll") for (int b = 0; b < UNR; ++b) v[b] = rq_src[min(tid + b * 256, rq_n4 - 1)];            \
        /* taps: plane pl is channel (plane0 + pl) % C; w12 rows are 3 float4 each */                             \
        const int rq_pl = min(tid / 3, rq_np - 1), rq_part = tid - (tid / 3) * 3;                                 \
        wv = reinterpret_cast<const f32x4*>(q.w12 + (size_t)((rq_plane0 + rq_pl) % q.C) * 12)[rq_part];          \
        bv = q.has_bias ? q.bias[(rq_plane0 + min(tid, rq_np - 1)) % q.C] : 0.f;                                  \
    }
    int chunk = blockIdx.x;
    if (chunk < chunks) FHIP_DW_REQUEST(chunk)
    for (; chunk < chunks; chunk += gridDim.x)
    {
        const int plane0 = chunk * chunk_planes;
        const int np = min(chunk_planes, q.planes - plane0);
        const int nf = np * HW, n4 = nf >> 2;
        {
            f32x4* dst = reinterpret_cast<f32x4*>(tile);
#pragma unroll
            for (int b = 0; b < UNR; ++b)
                if (tid + b * 256 < n4) dst[tid + b * 256] = v[b];
            for (int i = (n4 << 2) + tid; i < nf; i += 256) tile[i] = q.in[(size_t)plane0 * HW + i]; // last chunk of a ragged tensor only
            if (tid < 3 * np) reinterpret_cast<f32x4*>(wl)[tid] = wv;
            if (tid < np) bl[tid] = bv;
        }
        __syncthreads();
        if (chunk + (int)gridDim.x < chunks) FHIP_DW_REQUEST(chunk + (int)gridDim.x)
        const int nout = np * OHW;
        float* const obase = q.out + (size_t)plane0 * OHW; // 16-byte aligned
        if constexpr (dw_flat_row_path(HH, S))
        {
            // small odd planes at stride 1 -- 7 x 7 (49 floats; round 3), 5 x 5 and 9 x 9 (round 6) --: groups of four outputs straddle rows AND planes.
            // A lane owns one whole image row: 3 x HH LDS dwords, column validity known at compile time, the HH results go to a second LDS image
            // and leave as 16-byte stores.
            float* const otile = bl + chunk_planes; // [chunk_planes][HW]
            for (int it = tid; it < np * HH; it += 256)
            {
                const int pl = it / HH, y = it - pl * HH;
                const float4* w4 = reinterpret_cast<const float4*>(wl + pl * 12);
                const float4 wa = w4[0], wb = w4[1], wc = w4[2];
                const float w[9] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w, wc.x};
                const float bias = bl[pl];
                float acc[HH];
#pragma unroll
                for (int x = 0; x < HH; ++x) acc[x] = 0.f;
#pragma unroll
                for (int m = 0; m < 3; ++m)
                {
                    const int yy = y + m - 1;
                    const bool rowok = (unsigned)yy < (unsigned)HH;
                    const float* row = tile + pl * HW + min(max(yy, 0), HH - 1) * HH;
                    float r[HH];
#pragma unroll
                    for (int x = 0; x < HH; ++x) r[x] = rowok ? row[x] : 0.f;
#pragma unroll
                    for (int x = 0; x < HH; ++x)
#pragma unroll
                        for (int n = 0; n < 3; ++n)
                            if (x + n - 1 >= 0 && x + n - 1 < HH) acc[x] += r[x + n - 1] * w[m * 3 + n];
                }
#pragma unroll
                for (int x = 0; x < HH; ++x) otile[it * HH + x] = apply_act(acc[x] + bias, q.relu);
            }
            __syncthreads();
            const int n4o = nout >> 2;
            for (int i = tid; i < n4o; i += 256) reinterpret_cast<f32x4*>(obase)[i] = reinterpret_cast<const f32x4*>(otile)[i];
            for (int i = (n4o << 2) + tid; i < nout; i += 256) obase[i] = otile[i];
        }
        else
        for (int o0 = tid * 4; o0 < nout; o0 += 1024)
        {
            float res[4];
            if constexpr (S == 1 && (HW % 4) == 0)
            {
                // the four outputs share a plane; output flat index == input flat index
                const int pl = o0 / HW, r = o0 - pl * HW;
                const int y0 = r / WW, x0 = r - y0 * WW;
                const float4* w4 = reinterpret_cast<const float4*>(wl + pl * 12);
                const float4 wa = w4[0], wb = w4[1], wc = w4[2];
                const float w[9] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w, wc.x};
                const float bias = bl[pl];
                float t[3][6];
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int j = 0; j < 6; ++j) t[m][j] = tile[o0 + (m - 1) * WW - 1 + j];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                {
                    const bool wrap = x0 + e >= WW;
                    const int x = wrap ? x0 + e - WW : x0 + e, y = wrap ? y0 + 1 : y0;
                    float acc = 0.f;
#pragma unroll
                    for (int m = 0; m < 3; ++m)
                    {
                        const bool rowok = (unsigned)(y + m - 1) < (unsigned)HH;
#pragma unroll
                        for (int n = 0; n < 3; ++n)
                        {
                            const bool ok = rowok && (unsigned)(x + n - 1) < (unsigned)WW;
                            acc += (ok ? t[m][e + n] : 0.f) * w[m * 3 + n];
                        }
                    }
                    res[e] = apply_act(acc + bias, q.relu);
                }
            }
            else
            {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                {
                    const int o = min(o0 + e, nout - 1);
                    const int pl = o / OHW, r = o - pl * OHW;
                    const int oy = r / OW, ox = r - oy * OW;
                    const float* wp = wl + pl * 12;
                    const float* ip = tile + pl * HW + (oy * S - 1) * WW + ox * S - 1;
                    float acc = 0.f;
#pragma unroll
                    for (int m = 0; m < 3; ++m)
                    {
                        const bool rowok = (unsigned)(oy * S + m - 1) < (unsigned)HH;
#pragma unroll
                        for (int n = 0; n < 3; ++n)
                        {
                            const bool ok = rowok && (unsigned)(ox * S + n - 1) < (unsigned)WW;
                            acc += (ok ? ip[m * WW + n] : 0.f) * wp[m * 3 + n];
                        }
                    }
                    res[e] = apply_act(acc + bl[pl], q.relu);
                }
            }
            if (o0 + 3 < nout)
                stg4_act<4>(obase + o0, make_float4(res[0], res[1], res[2], res[3]));
            else
            {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (o0 + e < nout) obase[o0 + e] = res[e];
            }
        }
        if (chunk + (int)gridDim.x < chunks) __syncthreads(); // every lane is done with this chunk's LDS image
    }
#undef FHIP_DW_REQUEST
}

// The same idea for LARGE planes whose rows are whole float4 (W % 4 == 0: 112, 56): a block owns a BAND of RB output rows of one plane.
// The RB + 2 input rows it needs are one contiguous run of the tensor (coalesced 16-byte loads, the two halo rows are re-read by the
// neighbouring bands through L2), four consecutive outputs of a row per lane (a group never wraps: rows are whole float4), three 6-float
// LDS runs for 36 FMAs, one 16-byte store.  Stride 1, pad 1 on every side.  WW and RB are compile-time: no division by a variable.
template <int WW, int RB, int UNR>
__global__ __launch_bounds__(256) void depthwise3x3_band_kernel(const DwParams q, int bands_per_plane)
{
    constexpr int ROWS = RB + 2, W4 = WW / 4;
    __shared__ __attribute__((aligned(16))) float smem[4 + ROWS * WW + 4];
    float* const tile = smem + 4; // row r of the image = input row y0 - 1 + r
    const int tid = threadIdx.x;
    const int plane = blockIdx.x / bands_per_plane, band = blockIdx.x - plane * bands_per_plane;
    const int y0 = band * RB, c = plane % q.C;
    const int ya = max(y0 - 1, 0), yb = min(y0 + RB + 1, q.H); // input rows [ya, yb) exist
    const f32x4* src = reinterpret_cast<const f32x4*>(q.in + (size_t)plane * q.H * WW + (size_t)ya * WW);
    const int n4 = (yb - ya) * W4;
    f32x4 v[UNR];
#pragma unroll
    for (int b = 0; b < UNR; ++b) v[b] = src[min(tid + b * 256, n4 - 1)];
    // taps and bias: wave-uniform scalar loads
    const float4* w4 = reinterpret_cast<const float4*>(q.w12 + (size_t)c * 12);
    const float4 wa = w4[0], wb = w4[1], wc = w4[2];
    const float w[9] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w, wc.x};
    const float bias = q.has_bias ? q.bias[c] : 0.f;
    f32x4* dst = reinterpret_cast<f32x4*>(tile + (ya - (y0 - 1)) * WW);
#pragma unroll
    for (int b = 0; b < UNR; ++b)
        if (tid + b * 256 < n4) dst[tid + b * 256] = v[b];
    __syncthreads();
    const int rows_out = min(RB, q.OH - y0);
    float* const obase = q.out + (size_t)plane * q.OH * WW + (size_t)y0 * WW;
    for (int g = tid; g < rows_out * W4; g += 256)
    {
        const int ly = g / W4, x0 = (g - ly * W4) * 4;
        float t[3][6];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int j = 0; j < 6; ++j) t[m][j] = tile[(ly + m) * WW + x0 - 1 + j];
        float res[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
        {
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < 3; ++m)
            {
                const bool rowok = (unsigned)(y0 + ly + m - 1) < (unsigned)q.H;
#pragma unroll
                for (int n = 0; n < 3; ++n)
                {
                    const bool ok = rowok && (unsigned)(x0 + e + n - 1) < (unsigned)WW;
                    acc += (ok ? t[m][e + n] : 0.f) * w[m * 3 + n];
                }
            }
            res[e] = apply_act(acc + bias, q.relu);
        }
        stg4_act<4>(obase + (size_t)ly * WW + x0, make_float4(res[0], res[1], res[2], res[3]));
    }
}

// Plane sizes the band kernel is instantiated for: X(W, RB, UNR) -- 112 / 56 pixels (round 3: MobileNet-V1's conv2 / conv4 at 224-pixel inputs) and, round 6,
// the stride-1 planes above the flat kernel's range that the nets produce at 160 ... 288-pixel inputs (40 / 48 / 64 / 72 / 80 / 96 / 128 / 144).  RB output rows
// per band: (RB + 2) * W / 4 float4 requests fit UNR per lane, RB divides the plane (or the last band is ragged).
#define FHIP_DW_BAND_SIZES(X) X(144, 12, 2) X(128, 16, 3) X(112, 16, 2) X(96, 16, 2) X(80, 20, 2) X(72, 24, 2) X(64, 32, 3) X(56, 28, 2) X(48, 24, 2) X(40, 40, 2)

// does the band kernel take this geometry?  3x3, stride 1, pad 1 on every side, square planes of a size in FHIP_DW_BAND_SIZES
static inline bool dw_band_applicable(const DwParams& q, int pad_right, int pad_bottom)
{
    bool size_ok = false;
#define FHIP_X(W_, RB_, U_) size_ok = size_ok || q.H == W_;
    FHIP_DW_BAND_SIZES(FHIP_X)
#undef FHIP_X
    return q.KH == 3 && q.KW == 3 && q.SH == 1 && q.SW == 1 && q.PL == 1 && q.PT == 1 && pad_right == 1 && pad_bottom == 1 && q.H == q.W && size_ok;
}

static inline void dw_band_launch(const DwParams& q, hipStream_t s)
{
#define FHIP_X(W_, RB_, U_)                                                                                                                   \
    case W_:                                                                                                                                  \
    {                                                                                                                                         \
        static_assert((RB_ + 2) * (W_ / 4) <= 256 * U_ && W_ % 4 == 0, "a band's rows are U_ float4 requests per lane");                      \
        constexpr int bands = (W_ + RB_ - 1) / RB_;                                                                                           \
        hipLaunchKernelGGL((depthwise3x3_band_kernel<W_, RB_, U_>), dim3((unsigned)q.planes * (unsigned)bands), dim3(256), 0, s, q, bands); \
        break;                                                                                                                                \
    }
    switch (q.H)
    {
        FHIP_DW_BAND_SIZES(FHIP_X)
        default: break;
    }
#undef FHIP_X
}

// LDS bytes of the flat kernel for a chunk of `cp` planes of HH x HH
static inline size_t dw_flat_lds_bytes(int hh, int cp, int stride = 0)
{
    const int pad = (hh + 1 + 3) / 4 * 4;
    // row-per-lane path (5 / 7 / 9 pixels, stride 1): a second image for the results (cp * 13 floats of taps + bias is a multiple of 4 floats when cp % 4 == 0)
    return (size_t)(pad + cp * hh * hh + pad + 4 + cp * 13 + (dw_flat_row_path(hh, stride) ? cp * hh * hh : 0)) * sizeof(float);
}

// Plane sizes the flat kernel is instantiated for (round 6: any square plane of at most 36 pixels that the benchmark nets produce at 160 ... 288-pixel
// inputs -- MobileNet-V1: 20 / 10 / 5, 24 / 12 / 6, 28 / 14 / 7, 32 / 16 / 8, 36 / 18 / 9 -- instead of the three literals 7 / 14 / 28: at 160 pixels
// the nine depthwise layers behind conv5 took 570 us on the direct / chunk kernels against 370 us at equal pixels per step on 224-pixel inputs,
// tools/res_layers.py).  X(H) lists them once: the launcher's switch and the applicability test are generated from it.
#define FHIP_DW_FLAT_SIZES(X) X(5) X(6) X(7) X(8) X(9) X(10) X(12) X(14) X(16) X(18) X(20) X(24) X(28) X(32) X(36)
// planes per chunk: the measured optima for 28 / 14 / 7 pixels (tools/dw_bench.hip), ~3 100 (stride 1) / ~3 900 (stride 2) floats of planes for the
// other sizes (what the measured ones have); a multiple of 4 planes where H * H is not a multiple of 4 floats; at most 84 (a block's taps are
// requested by lanes 0 .. 3 * cp - 1, one float4 each)
constexpr int dw_flat_cp(int h, int stride)
{
    if (h == 28) return stride == 1 ? 4 : 5;
    if (h == 14) return stride == 1 ? 15 : 20;
    if (h == 7) return 36; // 252 row items
    int cp = (stride == 1 ? 3136 : 3920) / (h * h);
    if ((h * h) % 4) cp = cp / 4 * 4;
    return cp < 1 ? 1 : (cp > 84 ? 84 : cp);
}
constexpr int dw_flat_unr(int h, int stride) { return (dw_flat_cp(h, stride) * h * h / 4 + 255) / 256; }

// does the flat kernel take this geometry?  3x3, pad 1 on every side, stride 1 / 2, square planes of a size in FHIP_DW_FLAT_SIZES
static inline bool dw_flat_applicable(const DwParams& q, int pad_right, int pad_bottom)
{
    bool size_ok = false;
#define FHIP_X(H_) size_ok = size_ok || q.H == H_;
    FHIP_DW_FLAT_SIZES(FHIP_X)
#undef FHIP_X
    return q.KH == 3 && q.KW == 3 && q.SH == q.SW && (q.SH == 1 || q.SH == 2) && q.PL == 1 && q.PT == 1 && pad_right == 1 && pad_bottom == 1 &&
           q.H == q.W && size_ok;
}

// launch the flat kernel on `grid` persistent blocks with chunks of dw_flat_cp planes
static inline void dw_flat_launch(const DwParams& q, int grid, hipStream_t s)
{
#define FHIP_X(H_)                                                                                                                              \
    case H_:                                                                                                                                    \
    {                                                                                                                                           \
        constexpr int cp1 = dw_flat_cp(H_, 1), cp2 = dw_flat_cp(H_, 2);                                                                         \
        static_assert(cp1 * H_ * H_ <= 6 * 1024 && cp2 * H_ * H_ <= 6 * 1024 && dw_flat_unr(H_, 1) <= 6 && dw_flat_unr(H_, 2) <= 6, "chunk size"); \
        static_assert(((cp1 * H_ * H_) % 4) == 0 && ((cp2 * H_ * H_) % 4) == 0, "a chunk is whole float4s");                                    \
        const int cp = q.SH == 1 ? cp1 : cp2, chunks = ceil_div(q.planes, cp);                                                                  \
        const size_t lds = dw_flat_lds_bytes(H_, cp, q.SH);                                                                                     \
        if (q.SH == 1)                                                                                                                          \
            hipLaunchKernelGGL((depthwise3x3_flat_kernel<H_, 1, dw_flat_unr(H_, 1)>), dim3(std::min(grid, chunks)), dim3(256), lds, s, q, cp, chunks); \
        else                                                                                                                                    \
            hipLaunchKernelGGL((depthwise3x3_flat_kernel<H_, 2, dw_flat_unr(H_, 2)>), dim3(std::min(grid, chunks)), dim3(256), lds, s, q, cp, chunks); \
        break;                                                                                                                                  \
    }
    switch (q.H)
    {
        FHIP_DW_FLAT_SIZES(FHIP_X)
        default: break;
    }
#undef FHIP_X
}

// Small planes of any shape (7x7, 14x14 stride 2, 5x5 kernels ...): same chunk-of-whole-planes staging, then ONE
// output per lane with lanes along the flattened output index, so the LDS reads of a wave are consecutive
// addresses (stride SW) and the global stores are consecutive dwords.  planes_per_chunk is a multiple of 4,
// which keeps every chunk 16-byte aligned whatever H*W is.
__global__ __launch_bounds__(256) void depthwise_lds_scalar_kernel(const DwParams q)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HW = q.H * q.W, OHW = q.OH * q.OW, KK = q.KH * q.KW;
    float* tile = smem;                                                     // [planes_per_chunk][H][W]
    float* wl = smem + (((size_t)q.planes_per_chunk * HW + 3) & ~(size_t)3); // [planes_per_chunk][KK + 1]
    const int tid = threadIdx.x;
    const int chunks = (q.planes + q.planes_per_chunk - 1) / q.planes_per_chunk;
    for (int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x)
    {
        const int plane0 = chunk * q.planes_per_chunk;
        const int np = min(q.planes_per_chunk, q.planes - plane0);
        const int nf = np * HW;
        {
            const float* srcf = q.in + (size_t)plane0 * HW; // 16-byte aligned: plane0 % 4 == 0
            const float4* src = reinterpret_cast<const float4*>(srcf);
            float4* dst = reinterpret_cast<float4*>(tile);
            const int n4 = nf >> 2;
            for (int i0 = tid; i0 < n4; i0 += 256 * 8)
            {
                float4 v[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) v[b] = src[min(i0 + b * 256, n4 - 1)];
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (i0 + b * 256 < n4) dst[i0 + b * 256] = v[b];
            }
            for (int i = (n4 << 2) + tid; i < nf; i += 256) tile[i] = srcf[i];
            for (int i = tid; i < np * (KK + 1); i += 256)
            {
                const int pl = i / (KK + 1), e = i - pl * (KK + 1);
                const int c = (plane0 + pl) % q.C;
                wl[i] = e < KK ? q.w[(size_t)c * KK + e] : (q.has_bias ? q.bias[c] : 0.f);
            }
        }
        __syncthreads();
        const int nout = np * OHW;
        float* obase = q.out + (size_t)plane0 * OHW;
        for (int idx = tid; idx < nout; idx += 256)
        {
            const int pl = idx / OHW, r = idx - pl * OHW;
            const int oy = r / q.OW, ox = r - oy * q.OW;
            const float* ip = tile + (size_t)pl * HW;
            const float* wp = wl + pl * (KK + 1);
            float s = 0.f;
            for (int m = 0; m < q.KH; ++m)
            {
                const int y = oy * q.SH - q.PT + m;
                if ((unsigned)y >= (unsigned)q.H) continue;
                for (int n = 0; n < q.KW; ++n)
                {
                    const int x = ox * q.SW - q.PL + n;
                    if ((unsigned)x < (unsigned)q.W) s += ip[y * q.W + x] * wp[m * q.KW + n];
                }
            }
            obase[idx] = apply_act(s + wp[KK], q.relu);
        }
        __syncthreads();
    }
}

// Generic path: one output per lane, lanes along the flattened (plane, oy, ox) index.
__global__ __launch_bounds__(256) void depthwise_generic_kernel(const DwParams q, long long total)
{
    const int OHW = q.OH * q.OW, HW = q.H * q.W;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x)
    {
        const int plane = (int)(idx / OHW), r = (int)(idx - (long long)plane * OHW);
        const int oy = r / q.OW, ox = r - oy * q.OW;
        const int c = plane % q.C;
        const float* ip = q.in + (size_t)plane * HW;
        const float* wp = q.w + (size_t)c * q.KH * q.KW;
        float s = 0.f;
        for (int m = 0; m < q.KH; ++m)
        {
            const int y = oy * q.SH - q.PT + m;
            if ((unsigned)y >= (unsigned)q.H) continue;
            for (int n = 0; n < q.KW; ++n)
            {
                const int x = ox * q.SW - q.PL + n;
                if ((unsigned)x >= (unsigned)q.W) continue;
                s += ip[y * q.W + x] * wp[m * q.KW + n];
            }
        }
        if (q.has_bias) s += q.bias[c];
        q.out[idx] = apply_act(s, q.relu);
    }
}

// packed depthwise weights: the dense copy, then (3x3 kernels) a 12-floats-per-channel copy for 16-byte tap loads
size_t depthwise_packed_floats(const fhip_conv_param& p, size_t* w12_offset)
{
    const size_t dense = round_up_sz((size_t)p.group * p.kernel_h * p.kernel_w, 4);
    if (w12_offset) *w12_offset = dense;
    return dense + ((p.kernel_h == 3 && p.kernel_w == 3) ? (size_t)p.group * 12 : 0);
}

__global__ __launch_bounds__(256) void depthwise_pack12_kernel(float* __restrict__ w12, const float* __restrict__ w, int C)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * 12) return;
    const int c = i / 12, e = i - c * 12;
    w12[i] = e < 9 ? w[c * 9 + e] : 0.f;
}

int depthwise_init(const fhip_conv_param& p, float* packed, const float* kernel, hipStream_t s)
{
    StageTimer tm(FHIP_STAGE_INIT, s);
    const size_t n = (size_t)p.group * p.kernel_h * p.kernel_w;
    FHIP_CHECK_HIP(hipMemcpyAsync(packed, kernel, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    size_t off = 0;
    if (depthwise_packed_floats(p, &off) > off)
    {
        hipLaunchKernelGGL(depthwise_pack12_kernel, dim3(ceil_div(p.group * 12, 256)), dim3(256), 0, s, packed + off, kernel, p.group);
        FHIP_CHECK_HIP(hipGetLastError());
    }
    return FHIP_OK;
}

int depthwise_forward(const fhip_conv_param& p, int batch, float* out, const float* in, const float* kernel, const float* bias,
                      hipStream_t s)
{
    if (p.group != p.input_channels) return fail(FHIP_E_UNSUPPORTED, "depthwise needs group == input_channels");
    if (batch < 1) return fail(FHIP_E_BADARG, "batch < 1");
    if (p.bias_term && !bias) return fail(FHIP_E_BADARG, "bias_term set but bias_arr is NULL");
    DwParams q;
    q.in = in;
    q.w = kernel;
    {
        size_t off = 0;
        depthwise_packed_floats(p, &off);
        q.w12 = kernel + off;
    }
    q.bias = bias;
    q.out = out;
    q.C = p.input_channels;
    q.H = p.input_h;
    q.W = p.input_w;
    q.OH = p.output_h;
    q.OW = p.output_w;
    q.KH = p.kernel_h;
    q.KW = p.kernel_w;
    q.SH = p.stride_h > 0 ? p.stride_h : 1;
    q.SW = p.stride_w > 0 ? p.stride_w : 1;
    q.PL = p.pad_left;
    q.PT = p.pad_top;
    const long long planes = (long long)batch * q.C;
    if (planes > 0x7fffffffLL) return fail(FHIP_E_BADARG, "N*C too large");
    q.planes = (int)planes;
    q.has_bias = p.bias_term != 0;
    q.relu = p.activation == FHIP_ACT_RELU;
    q.planes_per_chunk = 0;
    if (q.OH < 1 || q.OW < 1) return fail(FHIP_E_BADARG, "empty output");

    const int HW = q.H * q.W;
    StageTimer tm(FHIP_STAGE_DEPTHWISE, s);
    const bool k3 = q.KH == 3 && q.KW == 3 && q.SH == q.SW && (q.SH == 1 || q.SH == 2) && q.PL == 1;
    // small stride-2 planes (MobileNet's 14 x 14 -> 7 x 7, where a lane of the direct kernel has only 2 x 2 outputs to amortise its halo
    // loads over): the chunk-of-planes kernel; everything else: the direct kernel
    const bool small_plane = k3 && q.SH == 2 && HW <= kDwChunkMaxPlane;
    // 14 x 14 and 28 x 28 planes, pad 1 all round (MobileNet-V1's conv6 ... conv13, 63 % of the depthwise bytes that stay depthwise
    // launches in the fused net): the flat kernel.  Chunk sizes measured on MI355X, tensors NOT resident in the memory-side cache
    // (tools/dw_bench.hip, b256): 28 px s1 73 us vs 107 us for the direct kernel, s2 50 vs 67; 14 px s1 40 vs 70, s2 26 vs 34 (chunk
    // kernel).  One chunk per block: persistent blocks with the next chunk prefetched were slower at every grid size.  7 x 7 stride 1
    // (conv14) takes the flat kernel's row-per-lane path: 23 us vs 49 us for the direct kernel (18 vs 37 cache-resident).
    // 112- and 56-pixel planes at stride 1 (conv2, and conv4 when it is not fused into its 1x1 layer): the band kernel, 144 vs 165 us
    // and 147 vs 181 us.
    // (odd planes at stride 2 -- 7 -> 4 pixels -- stay on the chunk kernel below, as measured for 7 x 7)
    if (dw_flat_applicable(q, p.pad_right, p.pad_bottom) && ((q.H & 1) == 0 || q.SH == 1))
        dw_flat_launch(q, 0x7fffffff, s);
    else if (dw_band_applicable(q, p.pad_right, p.pad_bottom) && planes * 12 <= 0x7fffffffLL)
        dw_band_launch(q, s);
    else if (small_plane)
    {
        // ~3136 floats (12.25 KB) of planes per block: 4 x 28^2, 16 x 14^2, 64 x 7^2; a multiple of 4 planes keeps every chunk
        // 16-byte aligned whatever HW is
        int cp = std::max(1, kDwChunkFloats / HW);
        if (HW % 4) cp = std::max(4, cp / 4 * 4);
        cp = (int)std::min<long long>(cp, planes);
        const int tile_floats = round_up(cp * HW, 4);
        const size_t lds = (size_t)(tile_floats + cp * 12) * sizeof(float);
        const long long chunks = (planes + cp - 1) / cp;
        if (chunks > 0x7fffffffLL) return fail(FHIP_E_BADARG, "N*C too large");
        if (q.SH == 1)
            hipLaunchKernelGGL(depthwise3x3_chunk_kernel<1>, dim3((unsigned)chunks), dim3(256), lds, s, q, cp, tile_floats);
        else
            hipLaunchKernelGGL(depthwise3x3_chunk_kernel<2>, dim3((unsigned)chunks), dim3(256), lds, s, q, cp, tile_floats);
    }
    else if (k3)
    {
        const int vx = ((q.W % 4) == 0 && (q.OW % 4) == 0) ? 4 : (((q.W % 2) == 0 && (q.OW % 2) == 0) ? 2 : 1);
        // output rows per lane: 4 at stride 1, 2 at stride 2 (measured against 1, 2 and 7: DESIGN.md 3.3)
        const int R = q.SH == 1 ? 4 : 2;
        const int yblocks = ceil_div(q.OH, R), xvecs = q.OW / vx;
        const long long total = planes * yblocks * xvecs;
        // grid-stride over at most 8192 blocks (measured optimum of 2048 ... 32768 and of "one item per lane")
        const int grid = (int)min(8192LL, (total + 255) / 256);
#define FHIP_DW_LAUNCH(S_, VX_, R_) \
    hipLaunchKernelGGL((depthwise3x3_direct_kernel<S_, VX_, R_>), dim3(grid), dim3(256), 0, s, q, yblocks, xvecs, total)
        if (q.SH == 1)
        {
            if (vx == 4) FHIP_DW_LAUNCH(1, 4, 4);
            else if (vx == 2) FHIP_DW_LAUNCH(1, 2, 4);
            else FHIP_DW_LAUNCH(1, 1, 4);
        }
        else
        {
            if (vx == 4) FHIP_DW_LAUNCH(2, 4, 2);
            else if (vx == 2) FHIP_DW_LAUNCH(2, 2, 2);
            else FHIP_DW_LAUNCH(2, 1, 2);
        }
#undef FHIP_DW_LAUNCH
    }
    else if (4 * (HW + q.KH * q.KW + 1) + 4 <= kDwLdsFloats)
    {
        // at least 4 whole planes fit: LDS-staged one-output-per-lane path, chunks of a multiple of 4 planes
        const int per_plane = HW + q.KH * q.KW + 1;
        q.planes_per_chunk = max(4, min(round_up(q.planes, 4), (kDwLdsFloats - 4) / per_plane / 4 * 4));
        const int chunks = ceil_div(q.planes, q.planes_per_chunk);
        const size_t lds = ((((size_t)q.planes_per_chunk * HW + 3) & ~(size_t)3) + (size_t)q.planes_per_chunk * (q.KH * q.KW + 1)) * sizeof(float);
        hipLaunchKernelGGL(depthwise_lds_scalar_kernel, dim3(min(chunks, 256 * 8)), dim3(256), lds, s, q);
    }
    else
    {
        const long long total = planes * q.OH * q.OW;
        const int grid = (int)min((long long)256 * 16, (total + 255) / 256);
        hipLaunchKernelGGL(depthwise_generic_kernel, dim3(grid), dim3(256), 0, s, q, total);
    }
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

} // namespace fhip
