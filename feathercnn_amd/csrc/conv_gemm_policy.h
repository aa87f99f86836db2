// conv_gemm_policy.h -- operand loaders / accumulator store of the implicit-GEMM convolution (IM2COL route) for the
// shared fp32-MFMA main loop in gemm_core.h: the B-operand loader gathers booster::im2col's column matrix
// (reference src/booster/avx/generic_kernels.cpp:50-85) straight from the NCHW input, the store applies the bias / ReLU /
// residual epilogue of packed_sgemm_activation<bias,relu> (avx/sgemm.cpp:377-433).
#pragma once

#include <type_traits>

#include "gemm_core.h"

namespace fhip
{

constexpr int kConvKTile = 16;
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4))); // a float4 that is only 4-byte aligned (MODE 5)

struct ConvGemmParams
{
    int batches, m_tiles, n_tiles, k_tiles;
    const float* Wt;
    const float* in;
    float* out;
    const float* bias;
    int C, K, H, W, OH, OW, SH, SW, PL, PT, KH, KW;
    int Kd;   // C*KH*KW
    int Kp;   // padded output channels
    int Kdp;  // padded reduction length (rows of a weight panel)
    int bm;   // rows per weight panel = the kernel's BM: Wt is [Kp / bm panels][Kdp][bm]
    int Ntot; // N*OH*OW
    int OHW, HW, KHW;
    int has_bias, relu;
    // split-K (under-filled grids): the GEMM "batch" index is the K split; split s reduces k-tiles
    // [s*k_tiles/S, (s+1)*k_tiles/S) of the k_tiles in total (pieces may differ by one tile) and writes raw partial sums to
    // partial[s][K][Ntot]; a reduce kernel finishes.  (An in-kernel fix-up -- the block that draws a tile's last ticket sums the
    // pieces -- was built and measured 3x SLOWER: the pieces of a tile run on different XCDs, whose L2s are only made coherent by
    // agent-scope fences, and every such fence writes back the whole L2.)
    int split_k;
    float* partial;
    // tail split (round 4; implicit_gemm.hip: igemm_tail): tiles % CUs column tiles at the end of an UNSPLIT launch would run as a last, nearly
    // empty round of lone blocks (ResNet-50's 256 -> 1024 @14^2 b64: 1568 tiles = 6.125 per CU, the CUs that draw a seventh finish 13 us after
    // the others).  Those column tiles [tail_nt0, n_tiles) are cut into split_k pieces along the reduction, which land one (or less) per CU;
    // tail_main > 0 = number of unsplit blocks, and the grid is tail_main + pieces (both multiples of 8: decode() keeps an XCD's pieces behind
    // ITS main blocks in dispatch order).  Pieces write partial[s][K][part_ntot] at column n - part_col0; the reduce kernel finishes them.
    int tail_main, tail_nt0;
    int part_ntot, part_col0; // row pitch / first column of `partial` (plain split-K: Ntot, 0)
    // optional residual (same layout as out), added before the activation: out = act(conv + bias + residual);
    // carried as a byte offset from `out` so the store path needs no second pointer table
    int has_residual;
    ptrdiff_t residual_delta;
    // MODE 3 / 4 (depthwise 3x3 fused into the 1x1 convolution that consumes it): `in` is the DEPTHWISE layer's input [N][C][H][W],
    // the B operand is act(dw3x3(in) + dw_bias) computed on the fly; OH/OW are the depthwise output = pointwise input / output dims
    const float* dw_w12; // [C][12]: 9 taps + 3 unused (depthwise_init's 16-byte tap rows)
    const float* dw_bias;
    int dw_stride, dw_relu;
    // TWIN policy (two 1x1 convolutions of the same input as one GEMM, ResNet's projection shortcut + the first layer of the main
    // branch): rows [0, twin_rows) of the concatenated filter matrix are the first layer and go to `out` ([N][twin_rows][OHW], `relu`),
    // rows [twin_rows, K) are the second layer and go to `out2` ([N][K - twin_rows][OHW], `relu2`).  twin_rows is a multiple of the
    // row tile, so a block writes one of the two tensors only.
    float* out2;
    int twin_rows, relu2;
    // MODE 5 (1x1 / stride 1 on planes whose size is not a multiple of 4 -- ResNet-50's 7 x 7 stage): Ntot counts pixel SLOTS, rag_gpi = ceil(OHW / 4)
    // groups of 4 per image; group g of an image covers pixels min(4 g, OHW - 4) ... + 3 (the last group is shifted back inside the image)
    int rag_gpi;
};

constexpr int kDwFusedMaxC = 256; // channels whose taps the fused route keeps in LDS (12 floats each)

// MODE 0: generic gather (any kernel / stride / pad)
// MODE 1: 1x1, pad 0, any stride (no tap decode, no bounds checks)
// MODE 2: 1x1, stride 1, pad 0, OH*OW % 4 == 0 : the column matrix IS the input -> 16-byte loads
// MODE 5: 1x1, stride 1, pad 0, OH*OW % 4 != 0 (and >= 4) -- ResNet-50's 7 x 7 stage (round 4).  MODE 1 serves such planes with four scalar
//         loads per B float4 and four scalar stores (+ four scalar residual loads) per accumulator float4.  Here the GEMM's columns are pixel
//         SLOTS: every image gets ceil(OHW / 4) groups of 4, the last group is SHIFTED BACK to the image's last four pixels (so every access
//         stays inside the tensor) -- one unaligned 16-byte load per B float4, one unaligned 16-byte store / residual load per accumulator
//         float4; the shifted group recomputes up to three pixels of its neighbour and stores only its new ones.  6 % more columns at 7 x 7
//         (52 slots for 49 pixels), a quarter of the memory instructions.  Same values as MODE 1 bit for bit (same k order per output).
// MODE 3 / 4: 1x1 stride 1 on the OUTPUT of a 3x3 depthwise convolution (MODE 3: depthwise stride 1, MODE 4: stride 2; pad_left =
//         pad_top = 1, W % 4 == 0, OW % 4 == 0, C <= kDwFusedMaxC) that is never written: the B loader fetches the depthwise INPUT
//         patch of its 4 output pixels -- per channel 3 rows x (4-byte, 16-byte, 4- or 16-byte) loads -- and `finish` does the 36
//         FMAs at LDS-write time, a k-tile after the loads were issued.  MobileNet's first pairs are HBM bound: the pair's
//         compulsory traffic drops from in + 2*mid + out to in + out.  (Taking the two halo columns from the neighbouring lanes with
//         shuffles -- 6 loads instead of 9 -- was measured: faster on the 64-row tile, slower on the 128-row one: the route is bound
//         by VALU issue next to the MFMAs, not by the address pipe.)
template <int MODE, bool TWIN = false>
struct ConvGemmPolicy
{
    using Params = ConvGemmParams;
    static constexpr bool DW = MODE == 3 || MODE == 4; // depthwise 3x3 computed into the B tile
    static constexpr int EXTRA_LDS_FLOATS = DW ? kDwFusedMaxC * 12 : 0;
    // MODE 3 / 4: the depthwise taps (9) + bias (slot 9) of every channel -> LDS, once per block
    static __device__ void stage_extra(const Params& p, float* extra, int tid, int threads)
    {
        if (!DW) return;
        for (int i = tid; i < p.C * 12; i += threads)
        {
            const int c = i / 12, e = i - c * 12;
            extra[i] = e == 9 ? (p.dw_bias ? p.dw_bias[c] : 0.f) : p.dw_w12[i];
        }
    }

    // bias of output row m for the prologue preload (split-K pieces store raw partial sums: no bias)
    // (tail split: the unsplit blocks carry split == split_k; the pieces' stores ignore the bias)
    static __device__ float bias_at(const Params& p, int m) { return (p.has_bias && (p.split_k <= 1 || p.tail_main) && m < p.K) ? p.bias[m] : 0.f; }
    static __device__ int k_cut(const Params& p, int split) { return (int)((long long)split * p.k_tiles / p.split_k); }
    static __device__ int k_first(const Params& p, int split) { return (p.tail_main && split >= p.split_k) ? 0 : k_cut(p, split); }
    static __device__ int k_count(const Params& p, int split)
    {
        return (p.tail_main && split >= p.split_k) ? p.k_tiles : k_cut(p, split + 1) - k_cut(p, split);
    }
    // block -> (row tile, column tile, split): the plain order of gemm_core.h, or the tail-split order
    static __device__ void decode(const Params& p, int& mt, int& nt, int& split)
    {
        if (!p.tail_main)
        {
            int vid = xcd_remap(blockIdx.x, p.batches * p.m_tiles * p.n_tiles);
            mt = vid % p.m_tiles;
            vid /= p.m_tiles;
            nt = vid % p.n_tiles;
            split = vid / p.n_tiles;
            return;
        }
        // XCD x = blockIdx % 8 runs its eighth of the main tiles, then its eighth of the pieces (the pieces of one tile next to each other)
        const int x = blockIdx.x & 7, local = blockIdx.x >> 3, mp = p.tail_main >> 3, tp = ((int)gridDim.x - p.tail_main) >> 3;
        if (local < mp)
        {
            const int id = x * mp + local;
            mt = id % p.m_tiles;
            nt = id / p.m_tiles;
            split = p.split_k;
        }
        else
        {
            const int q = x * tp + (local - mp), tt = q / p.split_k;
            split = q - tt * p.split_k;
            mt = tt % p.m_tiles;
            nt = p.tail_nt0 + tt / p.m_tiles;
        }
    }

    struct ALoad
    {
        const float* base;
        // panel-major weights: the rows a block streams are ONE contiguous run of memory (k-tile after k-tile), which is what
        // the InnerProduct shapes need from HBM (VGG fc6: 32 panels of 12.8 MB, each read once by the blocks of one panel)
        __device__ ALoad(const Params& p, int split, int m4)
            : base(p.Wt + ((size_t)(m4 / p.bm) * p.Kdp + (size_t)k_first(p, split) * kConvKTile) * p.bm + (m4 % p.bm))
        {
        }
        __device__ float4 load(const Params& p, int krow) const
        {
            return *reinterpret_cast<const float4*>(base + (size_t)krow * p.bm); // Wt zero padded in both dims
        }
    };

    // MODE 3 / 4: the depthwise input patch of 4 output pixels of one channel, rows y-1, y, y+1 (x0 = input column of tap 1 of output 0)
    struct DwRaw1 // stride 1: columns x0-1 .. x0+4
    {
        float l[3];
        float4 c[3];
        float r[3];
    };
    struct DwRaw2 // stride 2: columns x0-1 .. x0+7
    {
        float l[3];
        float4 c[3];
        float4 r[3];
    };
    struct BLoad
    {
        typedef typename std::conditional<MODE == 3, DwRaw1, typename std::conditional<MODE == 4, DwRaw2, float4>::type>::type Raw;
        const float* ptr[(MODE >= 2) ? 1 : 4]; // &in[n][0][iy0][ix0] of each of the 4 columns (may point before the plane)
        int iy0[MODE == 0 ? 4 : 1], ix0[MODE == 0 ? 4 : 1]; // MODE 3 / 4: [0] = first tap row / first centre column of the patch
        unsigned valid; // bit e: column n4+e < Ntot
        int koff;       // first reduction row of this K split
        __device__ BLoad(const Params& p, int split, int n4)
        {
            valid = 0;
            koff = k_first(p, split) * kConvKTile;
            if (DW)
            {
                // 4 consecutive columns are 4 consecutive pixels of one output row (OW % 4 == 0, n4 % 4 == 0)
                const int cc = n4 < p.Ntot ? n4 : 0;
                const int img = cc / p.OHW, rem = cc - img * p.OHW;
                const int oy = rem / p.OW, ox = rem - oy * p.OW;
                valid = n4 < p.Ntot ? 0xfu : 0u;
                iy0[0] = oy * p.dw_stride - 1; // input row of tap row 0 (pad_top = 1)
                ix0[0] = ox * p.dw_stride;     // input column under output 0, tap column 1 (pad_left = 1)
                ptr[0] = p.in + ((size_t)img * p.C) * p.HW;
            }
            else if (MODE == 2)
            {
                // 4 consecutive columns stay inside one image (OHW % 4 == 0, n4 % 4 == 0)
                const int img = n4 / p.OHW, rem = n4 - img * p.OHW;
                valid = n4 < p.Ntot ? 0xfu : 0u;
                ptr[0] = p.in + ((size_t)img * p.C) * p.HW + rem;
            }
            else if (MODE == 5)
            {
                // pixel slots: group (n4 / 4) % gpi of image n4 / (4 gpi), shifted back inside the image if it is the last one
                const int cc = n4 < p.Ntot ? n4 : 0;
                const int spi = 4 * p.rag_gpi, img = cc / spi, g4 = cc - img * spi;
                valid = n4 < p.Ntot ? 0xfu : 0u;
                ptr[0] = p.in + ((size_t)img * p.C) * p.HW + min(g4, p.OHW - 4);
            }
            else
            {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                {
                    const int col = n4 + e;
                    const bool ok = col < p.Ntot;
                    const int cc = ok ? col : 0;
                    const int img = cc / p.OHW, rem = cc - img * p.OHW;
                    const int oy = rem / p.OW, ox = rem - oy * p.OW;
                    const int y0 = oy * p.SH - p.PT, x0 = ox * p.SW - p.PL;
                    if (MODE == 0)
                    {
                        iy0[e] = y0;
                        ix0[e] = x0;
                    }
                    ptr[e] = p.in + ((size_t)img * p.C) * p.HW + (ptrdiff_t)y0 * p.W + x0;
                    valid |= ok ? (1u << e) : 0u;
                }
            }
        }
        // Unconditional loads from clamped addresses; `ok` says which of the 4 values are real (see gemm_core.h).
        // MODE 3 / 4: act(dw3x3 + bias) of the 4 pixels, taps accumulated in (m, n) order like depthwise3x3_direct_kernel
        __device__ float4 finish(const Params& p, const Raw& raw, int krow_in_split, const float* extra) const
        {
            if constexpr (!DW)
                return raw;
#ifdef FHIP_DWPW_ABLATE // measurement builds only (tools/dwpw_ab.sh): bit 0 = no depthwise arithmetic
            else if constexpr ((FHIP_DWPW_ABLATE & 1) != 0)
                return raw.c[1];
#endif
            else
            {
                const int c = min(krow_in_split + koff, p.C - 1);
                const float* w = extra + c * 12;
                constexpr int S = MODE == 3 ? 1 : 2;
                float out[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 3; ++m)
                {
                    if ((unsigned)(iy0[0] + m) >= (unsigned)p.H) continue; // padding row
                    // t[j] = input column ix0 - 1 + j
                    constexpr int NT = S == 1 ? 6 : 9;
                    float t[NT];
                    t[0] = ix0[0] > 0 ? raw.l[m] : 0.f;
                    t[1] = raw.c[m].x;
                    t[2] = raw.c[m].y;
                    t[3] = raw.c[m].z;
                    t[4] = raw.c[m].w;
                    if constexpr (S == 1)
                        t[5] = raw.r[m];
                    else
                    {
                        t[5] = raw.r[m].x;
                        t[6] = raw.r[m].y;
                        t[7] = raw.r[m].z;
                        t[8] = raw.r[m].w;
                    }
#pragma unroll
                    for (int j = 5; j < NT; ++j) t[j] = (ix0[0] - 1 + j < p.W) ? t[j] : 0.f; // right padding
                    const float w0 = w[m * 3], w1 = w[m * 3 + 1], w2 = w[m * 3 + 2];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                    {
                        out[e] += t[S * e] * w0;
                        out[e] += t[S * e + 1] * w1;
                        out[e] += t[S * e + 2] * w2;
                    }
                }
                const float b = w[9];
                float4 v = make_float4(out[0] + b, out[1] + b, out[2] + b, out[3] + b);
                if (p.dw_relu)
                {
                    v.x = fmaxf(v.x, 0.f);
                    v.y = fmaxf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f);
                    v.w = fmaxf(v.w, 0.f);
                }
                return v;
            }
        }
        __device__ Raw load(const Params& p, int krow_in_split, unsigned& ok) const
        {
            if constexpr (DW)
            {
                const int krow = krow_in_split + koff;
                ok = krow < p.Kd ? valid : 0u;
                const float* plane = ptr[0] + (size_t)min(krow, p.Kd - 1) * p.HW;
                Raw raw;
                const int x0 = ix0[0];
                const int xl = max(x0 - 1, 0);                        // x0 == 0: the left tap is padding, masked in finish
                const int xr = min(x0 + 4, p.W - (MODE == 3 ? 1 : 4)); // past the row: masked in finish
#pragma unroll
                for (int m = 0; m < 3; ++m)
                {
                    const int y = min(max(iy0[0] + m, 0), p.H - 1); // padding rows re-read a real row and are skipped in finish
                    const float* row = plane + (size_t)y * p.W;
#ifdef FHIP_DWPW_ABLATE // bit 1 = no halo loads (3 aligned float4 per channel instead of 9 mixed loads), bit 2 = the centre row only
                    if ((FHIP_DWPW_ABLATE & 4) && m != 1)
                    {
                        raw.l[m] = 0.f;
                        raw.c[m] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if constexpr (MODE == 3) raw.r[m] = 0.f;
                        else raw.r[m] = make_float4(0.f, 0.f, 0.f, 0.f);
                        continue;
                    }
                    if (FHIP_DWPW_ABLATE & 2)
                    {
                        raw.l[m] = 0.f;
                        raw.c[m] = *reinterpret_cast<const float4*>(row + x0);
                        if constexpr (MODE == 3) raw.r[m] = 0.f;
                        else raw.r[m] = make_float4(0.f, 0.f, 0.f, 0.f);
                        continue;
                    }
#endif
                    raw.l[m] = row[xl];
                    raw.c[m] = *reinterpret_cast<const float4*>(row + x0);
                    if constexpr (MODE == 3)
                        raw.r[m] = row[xr];
                    else
                        raw.r[m] = *reinterpret_cast<const float4*>(row + xr);
                }
                return raw;
            }
            else
                return load_plain(p, krow_in_split, ok);
        }
        __device__ float4 load_plain(const Params& p, int krow_in_split, unsigned& ok) const
        {
            const int krow = krow_in_split + koff;
            const bool kin = krow < p.Kd;
            const int kr = min(krow, p.Kd - 1);
            if (MODE == 2)
            {
                ok = kin ? valid : 0u;
                return *reinterpret_cast<const float4*>((valid ? ptr[0] : p.in) + (size_t)kr * p.HW);
            }
            if (MODE == 5)
            {
                ok = kin ? valid : 0u;
                const f32x4u v = *reinterpret_cast<const f32x4u*>(ptr[0] + (size_t)kr * p.HW); // 4-byte aligned: one global_load_dwordx4
                return make_float4(v.x, v.y, v.z, v.w);
            }
            float v[4];
            if (MODE == 1)
            {
                ok = kin ? valid : 0u;
                const size_t koff = (size_t)kr * p.HW;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ptr[e][koff]; // ptr[e] of an invalid column points at column 0
            }
            else
            {
                const int c = kr / p.KHW, r = kr - c * p.KHW;
                const int u = r / p.KW, w = r - u * p.KW;
                const ptrdiff_t koff = (ptrdiff_t)c * p.HW + (ptrdiff_t)u * p.W + w;
                ok = 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                {
                    const bool in = kin && (valid & (1u << e)) && ((unsigned)(iy0[e] + u) < (unsigned)p.H) &&
                                    ((unsigned)(ix0[e] + w) < (unsigned)p.W);
                    ok |= in ? (1u << e) : 0u;
                    v[e] = *(in ? ptr[e] + koff : p.in); // a padding tap reads in[0] and is zeroed at LDS-write time
                }
            }
            return make_float4(v[0], v[1], v[2], v[3]);
        }
    };

    struct Store
    {
        float* ptr[4]; // &out[img][0][rem] of each of the 4 columns
        unsigned valid;
        bool wide; // the 4 columns are one aligned 16-byte piece of one image
        float* part; // split-K: &partial[split][0][n4]
        float* ptr2[TWIN ? 4 : 1]; // TWIN: &out2[img][0][rem] - twin_rows * OHW, so that row m of the GEMM is ptr2[e] + m * OHW
        int first_new;             // MODE 5: first component of this slot group that no other group covers (0 but for an image's last group)
        __device__ Store(const Params& p, int split, int n4)
        {
            part = (p.split_k > 1 && (!p.tail_main || split < p.split_k)) ? p.partial + (size_t)split * p.K * p.part_ntot + (n4 - p.part_col0) : nullptr;
            valid = 0;
            first_new = 0;
            if (MODE == 5)
            {
                const int cc = n4 < p.Ntot ? n4 : 0;
                const int spi = 4 * p.rag_gpi, img = cc / spi, g4 = cc - img * spi, off = min(g4, p.OHW - 4);
                valid = n4 < p.Ntot ? 0xfu : 0u;
                first_new = g4 - off;
                ptr[0] = p.out + ((size_t)img * p.K) * p.OHW + off;
                ptr[1] = ptr[0] + 1;
                ptr[2] = ptr[0] + 2;
                ptr[3] = ptr[0] + 3;
                wide = false;
                return;
            }
            const int k_first = TWIN ? p.twin_rows : p.K; // channels of the tensor `out` points to
#pragma unroll
            for (int e = 0; e < 4; ++e)
            {
                const int col = n4 + e;
                const bool ok = col < p.Ntot;
                const int cc = ok ? col : 0;
                const int img = cc / p.OHW, rem = cc - img * p.OHW;
                ptr[e] = p.out + ((size_t)img * k_first) * p.OHW + rem;
                if (TWIN) ptr2[e] = p.out2 + ((long long)img * (p.K - p.twin_rows) - p.twin_rows) * p.OHW + rem;
                valid |= ok ? (1u << e) : 0u;
            }
            wide = (valid == 0xfu) && ((p.OHW & 3) == 0) && (ptr[3] == ptr[0] + 3);
        }
        __device__ void put4(const Params& p, int m, float4 v) const
        {
            put4b(p, m, v, (p.has_bias && !part && m < p.K) ? p.bias[m] : 0.f, residual4(p, m));
        }
        // the residual operand of this row's 4 columns where they are one aligned 16-byte piece (the ResNet shapes); the ragged case is
        // read element by element in put4b
        __device__ float4 residual4(const Params& p, int m) const
        {
            if (MODE == 5)
            {
                if (p.has_residual && valid && !part && m < p.K)
                {
                    const f32x4u r = *reinterpret_cast<const f32x4u*>(reinterpret_cast<const char*>(ptr[0] + (size_t)m * p.OHW) + p.residual_delta);
                    return make_float4(r.x, r.y, r.z, r.w);
                }
                return make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (p.has_residual && wide && !part && m < p.K)
                return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(ptr[0] + (size_t)m * p.OHW) + p.residual_delta);
            return make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __device__ void put4b(const Params& p, int m, float4 v, float b, float4 r) const
        {
            if (m >= p.K) return;
            if (part)
            {
                float* d = part + (size_t)m * p.part_ntot;
                if (valid == 0xfu && (p.part_ntot & 3) == 0)
                    *reinterpret_cast<float4*>(d) = v;
                else
                {
                    if (valid & 1u) d[0] = v.x;
                    if (valid & 2u) d[1] = v.y;
                    if (valid & 4u) d[2] = v.z;
                    if (valid & 8u) d[3] = v.w;
                }
                return;
            }
            v.x += b;
            v.y += b;
            v.z += b;
            v.w += b;
            const size_t moff = (size_t)m * p.OHW;
            if (MODE == 5)
            {
                if (!valid) return;
                if (p.has_residual)
                {
                    v.x += r.x;
                    v.y += r.y;
                    v.z += r.z;
                    v.w += r.w;
                }
                if (p.relu)
                {
                    v.x = fmaxf(v.x, 0.f);
                    v.y = fmaxf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f);
                    v.w = fmaxf(v.w, 0.f);
                }
                if (first_new == 0)
                {
                    f32x4u o;
                    o.x = v.x;
                    o.y = v.y;
                    o.z = v.z;
                    o.w = v.w;
                    *reinterpret_cast<f32x4u*>(ptr[0] + moff) = o; // 4-byte aligned: one global_store_dwordx4
                }
                else
                {
                    // an image's last group: components below first_new are pixels the group before already owns
                    if (first_new <= 1) ptr[0][moff + 1] = v.y;
                    if (first_new <= 2) ptr[0][moff + 2] = v.z;
                    ptr[0][moff + 3] = v.w;
                }
                return;
            }
            if (TWIN)
            {
                // no residual, no split-K in this form (igemm_twin_forward)
                const bool second = m >= p.twin_rows;
                if (second ? p.relu2 : p.relu)
                {
                    v.x = fmaxf(v.x, 0.f);
                    v.y = fmaxf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f);
                    v.w = fmaxf(v.w, 0.f);
                }
                // (selected element by element: a pointer to one of the two register arrays would send both to scratch memory)
                if (wide)
                    stg4_act<1>((second ? ptr2[0] : ptr[0]) + moff, v);
                else
                {
                    if (valid & 1u) (second ? ptr2[0] : ptr[0])[moff] = v.x;
                    if (valid & 2u) (second ? ptr2[1] : ptr[1])[moff] = v.y;
                    if (valid & 4u) (second ? ptr2[2] : ptr[2])[moff] = v.z;
                    if (valid & 8u) (second ? ptr2[3] : ptr[3])[moff] = v.w;
                }
                return;
            }
            if (p.has_residual)
            {
                auto res = [&](const float* o) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(o) + p.residual_delta); };
                if (wide)
                {
                    v.x += r.x;
                    v.y += r.y;
                    v.z += r.z;
                    v.w += r.w;
                }
                else
                {
                    if (valid & 1u) v.x += res(ptr[0] + moff);
                    if (valid & 2u) v.y += res(ptr[1] + moff);
                    if (valid & 4u) v.z += res(ptr[2] + moff);
                    if (valid & 8u) v.w += res(ptr[3] + moff);
                }
            }
            if (p.relu)
            {
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
            }
            if (wide)
                stg4_act<1>(ptr[0] + moff, v);
            else
            {
                if (valid & 1u) ptr[0][moff] = v.x;
                if (valid & 2u) ptr[1][moff] = v.y;
                if (valid & 4u) ptr[2][moff] = v.z;
                if (valid & 8u) ptr[3][moff] = v.w;
            }
        }
    };
};

} // namespace fhip
