// dwpw_band.h -- depthwise 3x3 + the 1x1 convolution behind it as ONE kernel: band-staged, wave-specialised (round 4).
//
// Reference functions replaced: booster::depthwise (src/booster/avx/booster.cpp:136-160, avx/depthwise.cpp) followed by IM2COL_Forward of
// the 1x1 layer (avx/booster.cpp:83-102 + avx/sgemm.cpp:377-433) -- MobileNet's dw -> pw pairs.  ConvGemmPolicy<3|4> (round 2) already runs
// such a pair as one launch, but every thread computes its 16-byte piece of the GEMM's B operand from NINE mixed global loads and ~66
// vector-ALU instructions and then joins the MFMAs: tools/dwpw_ab.sh (round 4, ablation builds, MobileNet-V1 b256) put ~50 us of each
// 250 - 380 us pair on the six halo loads and ~50 us on the arithmetic, and the pairs ran at 0.35 of either roofline.  A first band-staged
// version of this file, in which all waves of a block alternated between the depthwise phase and the GEMM phase, measured the two phases
// strictly ADDITIVE (conv4 pair: 394 us = 158 us of MFMAs at their peak rate + 90 us of depthwise arithmetic + staging; co-resident blocks
// run in lock step, so nothing overlapped).  Hence the structure here, organised around the depthwise layer's geometry:
//
//   * a block owns R whole rows of the pair's output of ONE image (R * OW = 224 pixels: 2 rows of 112, 4 of 56, 8 of 28) x ALL output channels
//     of its channel block, and walks the input channels in chunks of CH; its waves are SPECIALISED:
//   * PRODUCER waves (PW of them) fetch, per chunk, the input band under those rows -- CH channels x ((R - 1) S + 3) rows x W columns -- with
//     fully coalesced, unconditional 16-byte row loads, three chunks ahead, into one of two LDS bands whose pad columns and out-of-image rows
//     ARE zeros; the depthwise arithmetic then needs no masks: a thread keeps one channel's 9 taps + bias in registers and produces 4
//     consecutive outputs from one aligned ds_read_b128 (two at stride 2) and two (one) scalars per tap row -- 36 FMAs in the depthwise
//     kernels' (row, tap) order, bias, ReLU -- into the k-major B tile of the NEXT chunk;
//   * CONSUMER waves (one per 32 output channels) run the 1x1 GEMM of the CURRENT chunk on v_mfma_f32_32x32x2_f32: a wave owns 32 output
//     channels x all 224 pixels -- the first 128 as four INTERLEAVED pixel sets {4 l + t} (one ds_read_b128 feeds four MFMAs and the
//     accumulators hold four consecutive pixels: dwordx4 stores, no transpose, as in stream_gemm.h), the last 96 as three plain 32-column
//     tiles -- with the A operand straight from the streamed kernel's packed image (CH / 2 coalesced dwords per lane and chunk, a chunk ahead);
//   * a producer and a consumer share every SIMD, so the vector ALU work of chunk i + 1 issues under the MFMAs of chunk i; ONE barrier per chunk.
// HBM sees the pair's input once (+ the two halo rows per band, from L2 -- which takes the round-5 item order below: with round 4's order
// the halo rows came over the fabric again, 1.35x the algorithmic bytes) and its output once; the depthwise output never exists.
//
// WHERE IT IS USED (round 4, measured): the kernel is memory-parallelism bound -- a persistent block keeps at most two chunks of requests in
// flight -- so it pays where a pair is HBM-bound and the block is small enough for TWO blocks per CU: MobileNet-V1's first pair (32 -> 64
// channels on 112 x 112, the pair ConvGemmPolicy<3> refuses because its GEMM is 2 k-tiles deep): 406 us as two kernels -> 369 us with one
// block per CU -> 344 us with two (tools/dwpw_bench.py, MobileNet-V1 b256).  On the 64- and 128-channel pairs every structure tried here lost
// to ConvGemmPolicy<3|4> (tools/experiments/dwpw_band.h keeps that history and the phase-timeline instrumentation).
#pragma once

#include <type_traits>
#include <utility>

#include "common.h"

#ifndef FHIP_BAND_ORDER
#define FHIP_BAND_ORDER 1 // 0: a contiguous share of the items per block (round 4), 1: rounds of XCD-contiguous items (round 5)
#endif
#ifndef FHIP_BAND_ABLATE
#define FHIP_BAND_ABLATE 0 // measurement builds only (tools/dwpw_ab.sh): 1 no depthwise arithmetic, 2 no MFMAs, 4 no band fetch, 8 no stores
#endif

namespace fhip
{

template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant in the body
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

struct DwPwBandParams
{
    const float* in;      // [N][C][H][W]   the depthwise layer's input
    const float* dw_w12;  // [C][12]        9 taps + 3 unused (depthwise_init's 16-byte tap rows)
    const float* dw_bias; // [C] or nullptr
    const float* wp;      // [K / 32][C / 2][64]  the pointwise filters as the MFMA A-operand image (stream_pack_weights_kernel)
    const float* pw_bias; // [K] or nullptr
    float* out;           // [N][K][OH][OW]
    int N, K, H, OH;
    int dw_relu, pw_relu;
    int groups;  // row groups per image = ceil(OH / R)
    int m_tiles; // blocks of 32 * CW output channels = K / (32 * CW)
    int bands;   // N * groups * m_tiles work items; a (persistent) block takes a contiguous share of them
};

// W: input row width, S: stride, R: output rows per block, C: channels, CH: channels per chunk, CW: consumer waves (32 output channels each),
// PW: producer waves
template <int W_, int S_, int R_, int C_, int CH_, int CW_, int PW_, int BPC_ = 1>
struct DwPwBandShape
{
    static constexpr int W = W_, S = S_, R = R_, C = C_, CH = CH_, CW = CW_, PW = PW_;
    static constexpr int BPC = BPC_; // persistent blocks per CU (their LDS and registers must fit that many)
    static constexpr int OW = W / S, BN = R * OW;
    static constexpr int BNP = 288; // B-tile row pitch: 224 columns + slack, = 32 banks mod 64 so that the two k rows a wave reads at once never collide
    static constexpr int RI = (R - 1) * S + 3; // input rows under R output rows
    static constexpr int LW = W + 8;           // band row: [0..2] unused, [3] = x -1 (zero), [4 .. W + 3] = x 0 .. W - 1, [W + 4] = x W (zero)
    static constexpr int W4 = W / 4, Q4 = BN / 4;
    static constexpr int THREADS = 64 * (2 * CW + PW), PT = 64 * PW; // consumers: CW waves on pixels 0 .. 127, CW waves on pixels 128 .. 223
    static constexpr int NCH = C / CH;
    static constexpr int STAGE_ITEMS = CH * RI * W4, SR = (STAGE_ITEMS + PT - 1) / PT;
    static constexpr int CP = RI * LW + 32; // band channel pitch: + 32 banks, so the two channels a producer wave reads at once do not collide
    static constexpr int BAND_FLOATS = CH * CP, BT_FLOATS = CH * BNP, TAP_FLOATS = C * 12;
    static constexpr int LDS_FLOATS = 2 * BAND_FLOATS + 2 * BT_FLOATS + TAP_FLOATS;
    static_assert(BN == 224, "the wave tiling below is written for 224 pixels per block (128 interleaved + 96 plain)");
    static_assert(W % 4 == 0 && OW % 4 == 0 && C % CH == 0 && CH % 2 == 0 && PT % CH == 0 && (C / CH) % 2 == 0, "alignment; an even number of chunks keeps the buffer parity");
};

template <class SH>
__global__ __launch_bounds__(SH::THREADS, SH::BPC * SH::THREADS / 256) void dwpw_band_kernel(const DwPwBandParams q)
{
    constexpr int W = SH::W, S = SH::S, R = SH::R, C = SH::C, CH = SH::CH, OW = SH::OW, RI = SH::RI, LW = SH::LW, BNP = SH::BNP;
    constexpr int PT = SH::PT, NCH = SH::NCH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const band0 = smem;                                          // 2 x [CH][RI][LW]
    float* const bt0 = smem + 2 * SH::BAND_FLOATS;                      // 2 x [CH][BNP]
    float* const taps = smem + 2 * SH::BAND_FLOATS + 2 * SH::BT_FLOATS; // [C][12]: 9 taps, [9] = bias

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    // roles: waves 0 .. CW-1 = consumers of pixels 0 .. 127 (interleaved sets), CW .. 2CW-1 = consumers of pixels 128 .. 223 (plain tiles),
    // the rest = producers.  Waves go to SIMDs round robin, so with CW a multiple of 4 every SIMD carries one of each kind: the two consumers'
    // MFMA streams cover each other's LDS latency, the producer's vector ALU work issues under both.
    const bool producer = wave >= 2 * SH::CW;
    const bool plain = wave >= SH::CW;       // (consumers) the 96-pixel group
    const int cw = plain ? wave - SH::CW : wave; // consumer index inside its group = 32-row group of output channels inside the channel block
    const int ptid = tid - 64 * 2 * SH::CW;   // producers: 0 .. PT - 1
    // PERSISTENT blocks: a block takes a contiguous share of the work items (image, row group, block of output channels -- the channel blocks of
    // one band are neighbours: they read the same input band at the same time) and runs them as ONE chunk pipeline, so that the producers are
    // already two chunks into the next item while the consumers store the last one: no block start / end is ever exposed.
    // ITEM ORDER (round 5).  Neighbouring bands share two of their four input rows.  With a contiguous share of the items per block (round 4)
    // the blocks that run at the same time are half an image apart, a band's halo rows are long gone from the 4 MB L2 when its neighbour comes
    // round, and every input row crosses the fabric twice (rocprofv3, MobileNet-V1 b256: 846 MB fetched for a 411 MB input).  Here the grid
    // walks the items in rounds of gridDim.x, and inside a round the XCD blockIdx % 8 owns a contiguous eighth: at any moment an XCD's blocks
    // work on consecutive bands, so a band's halo rows are its neighbour's rows, requested within microseconds of each other into the same L2.
    // item(t) = item0 + t * step for t < n_items.
    int item0, step, n_items;
    if (FHIP_BAND_ORDER == 1 && (gridDim.x & 7) == 0)
    {
        item0 = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        step = gridDim.x;
        n_items = item0 < q.bands ? (q.bands - item0 + step - 1) / step : 0;
    }
    else
    {
        item0 = (int)((long long)q.bands * blockIdx.x / gridDim.x);
        step = 1;
        n_items = (int)((long long)q.bands * (blockIdx.x + 1) / gridDim.x) - item0;
    }
    if (n_items <= 0) return;
    const int item1 = item0 + n_items * step; // (exclusive bound of the block's walk)
    const long long total_chunks = (long long)n_items * NCH;
    // ---- common prologue: pad columns of both bands (never written by the stager), taps + bias of all channels
    for (int i = tid; i < 2 * CH * RI; i += SH::THREADS)
    {
        const int c2 = i / RI, row = i - c2 * RI; // c2: channel slot over both buffers
        float* const rp = band0 + (c2 / CH) * SH::BAND_FLOATS + (c2 % CH) * SH::CP + row * LW;
        rp[3] = 0.f;
        rp[W + 4] = 0.f;
    }
    for (int i = tid; i < C * 12; i += SH::THREADS)
    {
        const int c = i / 12, e = i - c * 12;
        float v = q.dw_w12[i];
        if (e == 9) v = q.dw_bias ? q.dw_bias[c] : 0.f;
        taps[i] = v;
    }

    // ---- producer state.  Every load is UNCONDITIONAL from a clamped address (gemm_core.h's rule: a load under a branch makes hipcc wait
    // vmcnt(0) right behind it, which turns the requests of a chunk into as many serial round trips); rows outside the image are zeroed at
    // LDS-write time.
    // TWO sets of request registers: chunk j's requests go out at iteration j - 4 and are written to LDS at iteration j - 2, so a request
    // has two whole chunk periods to land (one set, requested one period ahead, left every chunk waiting for its HBM round trip: the phases
    // of the first wave-specialised version were still additive)
    float4 sr[2][SH::SR];
    unsigned sr_ok[2] = {0, 0}; // bit u: request u is real data
    int f_item = item0, f_ch = 0; // the fetch cursor: next chunk to request
    auto fetch = [&](auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        const int band_id = f_item / q.m_tiles;
        const int n = band_id / q.groups, g = band_id - n * q.groups;
        const int iy0 = g * R * S - 1; // input row of band row 0
        const float* const img = q.in + ((size_t)n * C + (size_t)f_ch * CH) * q.H * W;
        sr_ok[SET] = 0;
#pragma unroll
        for (int u = 0; u < SH::SR; ++u)
        {
            const int it = min(ptid + u * PT, SH::STAGE_ITEMS - 1);
            const int c = it / (RI * SH::W4), rem = it - c * (RI * SH::W4);
            const int row = rem / SH::W4, x4 = rem - row * SH::W4;
            const int iy = iy0 + row;
            sr_ok[SET] |= ((unsigned)iy < (unsigned)q.H) ? (1u << u) : 0u;
            const int iyc = min(max(iy, 0), q.H - 1);
            if (FHIP_BAND_ABLATE & 4)
                sr[SET][u] = make_float4(1.f, 2.f, 3.f, 4.f);
            else
                sr[SET][u] = *reinterpret_cast<const float4*>(img + ((size_t)c * q.H + iyc) * W + 4 * x4);
        }
        if (++f_ch == NCH)
        {
            f_ch = 0;
            f_item += step;
        }
    };
    auto stash = [&](auto ch_c) { // chunk parity = band buffer = request set
        constexpr int ch = decltype(ch_c)::value, SET = ch & 1;
        float* const band = band0 + (ch & 1) * SH::BAND_FLOATS;
#pragma unroll
        for (int u = 0; u < SH::SR; ++u)
        {
            const int it = ptid + u * PT;
            const int c = it / (RI * SH::W4), rem = it - c * (RI * SH::W4);
            const int row = rem / SH::W4, x4 = rem - row * SH::W4;
            const bool ok = (sr_ok[SET] >> u) & 1u;
            const float4 v = make_float4(ok ? sr[SET][u].x : 0.f, ok ? sr[SET][u].y : 0.f, ok ? sr[SET][u].z : 0.f, ok ? sr[SET][u].w : 0.f);
            if (it < SH::STAGE_ITEMS) *reinterpret_cast<float4*>(band + c * SH::CP + row * LW + 4 + 4 * x4) = v;
        }
    };
    // depthwise 3x3 of chunk ch: band[ch & 1] -> B tile bt[ch & 1] ([CH][BNP], k-major); a thread keeps ONE channel of the chunk
    auto depthwise = [&](int ch) {
        constexpr int TPC = PT / CH;
        const float* const band = band0 + (ch & 1) * SH::BAND_FLOATS;
        float* const bt = bt0 + (ch & 1) * SH::BT_FLOATS;
        const int c = ptid / TPC, sub = ptid - c * TPC;
        const float* tp = taps + (ch * CH + c) * 12;
        const float4 t0 = *reinterpret_cast<const float4*>(tp), t1 = *reinterpret_cast<const float4*>(tp + 4), t2 = *reinterpret_cast<const float4*>(tp + 8);
        const float wgt[3][3] = {{t0.x, t0.y, t0.z}, {t0.w, t1.x, t1.y}, {t1.z, t1.w, t2.x}};
        const float b = t2.y;
        for (int q4 = sub; q4 < SH::Q4; q4 += TPC)
        {
            if (FHIP_BAND_ABLATE & 1)
            {
                *reinterpret_cast<float4*>(bt + c * BNP + 4 * q4) = make_float4(b, wgt[0][0], wgt[1][1], wgt[2][2]);
                continue;
            }
            const int r = (4 * q4) / OW, x0 = 4 * q4 - r * OW;
            const float* bp = band + c * SH::CP + (r * S) * LW + S * x0 + 3; // band column of input x = S x0 - 1
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 3; ++m)
            {
                constexpr int NT = S == 1 ? 6 : 9;
                float t[NT];
                const float* row = bp + m * LW;
                t[0] = row[0];
                const float4 v = *reinterpret_cast<const float4*>(row + 1);
                t[1] = v.x;
                t[2] = v.y;
                t[3] = v.z;
                t[4] = v.w;
                if constexpr (S == 1)
                    t[5] = row[5];
                else
                {
                    const float4 v2 = *reinterpret_cast<const float4*>(row + 5);
                    t[5] = v2.x;
                    t[6] = v2.y;
                    t[7] = v2.z;
                    t[8] = v2.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                {
                    o[e] += t[S * e] * wgt[m][0];
                    o[e] += t[S * e + 1] * wgt[m][1];
                    o[e] += t[S * e + 2] * wgt[m][2];
                }
            }
            float4 res = make_float4(o[0] + b, o[1] + b, o[2] + b, o[3] + b);
            if (q.dw_relu)
            {
                res.x = fmaxf(res.x, 0.f);
                res.y = fmaxf(res.y, 0.f);
                res.z = fmaxf(res.z, 0.f);
                res.w = fmaxf(res.w, 0.f);
            }
            *reinterpret_cast<float4*>(bt + c * BNP + 4 * q4) = res;
        }
    };

    // The two roles are separate top-level branches, each with its own copy of the chunk loop and the SAME number of barriers, so that the
    // register allocator sees the producer's request registers and the consumer's 112 accumulators as the disjoint live ranges they are
    // (one interleaved loop with `if (producer)` inside cost 256 registers + scratch: hipcc cannot know that the predicate is wave-uniform
    // and constant).  s_barrier counts waves, not code locations.  Global chunk j = item * NCH + ch; buffers alternate with j (NCH is even, so
    // the parity is the compile-time ch & 1); one barrier behind every chunk but the very last.
    __syncthreads(); // pads, taps
    if (producer)
    {
        // ---- pipeline prologue: band 0 in LDS, its B tile, band 1 in LDS, chunks 2 and 3 requested
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        fetch(I0{});
        stash(I0{});
        if (total_chunks > 1) fetch(I1{});
        if (total_chunks > 2) fetch(I0{});
        __syncthreads(); // band 0 (producers only read it; the consumers just arrive)
        depthwise(0);
        if (total_chunks > 1) stash(I1{});
        if (total_chunks > 3) fetch(I1{});
        __syncthreads(); // B tile 0, band 1
        long long j = 0;
        for (int item = item0; item < item1; item += step)
        {
            static_for<NCH>([&](auto ch_c) {
                constexpr int ch = decltype(ch_c)::value;
                // chunk j + 1's B tile (its band was completed before the last barrier), chunk j + 2's band into the buffer chunk j's left
                // (from the request set of its parity), chunk j + 4 requested into that set
                if (j + 1 < total_chunks) depthwise((ch + 1) % NCH);
                if (j + 2 < total_chunks) stash(std::integral_constant<int, (ch + 2) % NCH>{});
                if (j + 4 < total_chunks) fetch(std::integral_constant<int, ch & 1>{});
                if (j + 1 < total_chunks) __syncthreads(); // B tile j + 1 and band j + 2 are complete; B tile j and band j + 1 are free
                ++j;
            });
        }
        return;
    }

    // ---- consumers.  NG = accumulators of the wave: 4 interleaved pixel sets {4 l + t} of pixels 0 .. 127 (one ds_read_b128 per k step, the
    // accumulators hold four consecutive pixels: dwordx4 stores), or 3 plain 32-pixel tiles of pixels 128 .. 223 (three ds_read_b32, dword stores)
    auto consume = [&](auto ng_c) {
        constexpr int NG = decltype(ng_c)::value;
        f32x16 acc[NG];
        // the A operand -- this wave's 32 output channels x all C input channels, C / 2 dwords per lane of the packed image -- stays in
        // REGISTERS across chunks and across items; it is re-read only when the block of output channels changes (never, with one channel
        // block).  (Requesting it chunk by chunk "a chunk ahead" did not survive hipcc: the loads were sunk behind the MFMAs that still read
        // the registers they overwrite, i.e. to the end of the chunk, and every chunk then began with an exposed L2 round trip.)
        float a_reg[C / 2];
        int mg_loaded = (item0 % q.m_tiles) * SH::CW + cw;
        {
            const float* const ap0 = q.wp + (size_t)mg_loaded * (C / 2) * 64 + lane;
#pragma unroll
            for (int j = 0; j < C / 2; ++j) a_reg[j] = ap0[(size_t)j * 64];
        }
#pragma unroll
        for (int i = 0; i < NG; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        __syncthreads(); // band 0
        __syncthreads(); // B tile 0, band 1
        long long j = 0;
        for (int item = item0; item < item1; item += step)
        {
            const int mt = item % q.m_tiles, band_id = item / q.m_tiles;
            const int n = band_id / q.groups, g = band_id - n * q.groups;
            const int r0 = g * R;            // first output row of the item
            const int mg = mt * SH::CW + cw; // this wave's 32-row group of output channels
            if (mg != mg_loaded)
            {
                mg_loaded = mg;
                const float* const ap = q.wp + (size_t)mg * (C / 2) * 64 + lane;
#pragma unroll
                for (int jj = 0; jj < C / 2; ++jj) a_reg[jj] = ap[(size_t)jj * 64];
            }
            // the pointwise bias of this lane's 16 rows: requested now, used by the stores behind the item's last chunk
            const int kbase = 32 * mg + 4 * half;
            float pb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pb[r] = q.pw_bias ? q.pw_bias[kbase + (r & 3) + 8 * (r >> 2)] : 0.f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch, ++j)
            {
                const float* const bt = bt0 + (ch & 1) * SH::BT_FLOATS + half * BNP + (NG == 4 ? 4 * l31 : 128 + l31);
#pragma unroll
                for (int kp = 0; kp < CH / 2; ++kp)
                {
                    const float a = a_reg[ch * (CH / 2) + kp];
                    const float* brow = bt + (2 * kp) * BNP;
                    if (FHIP_BAND_ABLATE & 2)
                    {
                        acc[0][kp & 15] += a * brow[0];
                        continue;
                    }
                    if constexpr (NG == 4)
                    {
                        const float4 b4 = *reinterpret_cast<const float4*>(brow);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b4.x, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b4.y, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b4.z, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b4.w, acc[3], 0, 0, 0);
                    }
                    else
                    {
                        const float b0 = brow[0], b1 = brow[32], b2 = brow[64];
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
                    }
                }
                if (j + 1 < total_chunks) __syncthreads();
            }

            // ---- the item's stores (the producers are already at work on the next item's chunks).  C/D layout of the 32x32 MFMA:
            // column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
            float* const obase = q.out + ((size_t)n * q.K + kbase) * q.OH * OW + (size_t)r0 * OW;
            const size_t kstride = (size_t)q.OH * OW;
            const int rows_left = q.OH - r0; // output rows of this item that exist
            if (!((FHIP_BAND_ABLATE & 8) && l31 != 0))
            {
                if constexpr (NG == 4)
                {
                    const int pix = 4 * l31, prow = pix / OW;
                    if (prow < rows_left)
                    {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                        {
                            const int row = (r & 3) + 8 * (r >> 2);
                            const float b = pb[r];
                            float4 v = make_float4(acc[0][r] + b, acc[1][r] + b, acc[2][r] + b, acc[3][r] + b);
                            if (q.pw_relu)
                            {
                                v.x = fmaxf(v.x, 0.f);
                                v.y = fmaxf(v.y, 0.f);
                                v.z = fmaxf(v.z, 0.f);
                                v.w = fmaxf(v.w, 0.f);
                            }
                            stg4_act<8>(obase + (size_t)row * kstride + pix, v);
                        }
                    }
                }
                else
                {
#pragma unroll
                    for (int jx = 0; jx < 3; ++jx)
                    {
                        const int pix2 = 128 + 32 * jx + l31, prow2 = pix2 / OW;
                        if (prow2 < rows_left)
                        {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                            {
                                const int row = (r & 3) + 8 * (r >> 2);
                                float v = acc[jx][r] + pb[r];
                                if (q.pw_relu) v = fmaxf(v, 0.f);
                                obase[(size_t)row * kstride + pix2] = v;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NG; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
    };
    if (plain)
        consume(std::integral_constant<int, 3>{});
    else
        consume(std::integral_constant<int, 4>{});
}

} // namespace fhip
