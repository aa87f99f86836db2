// wino_chain_band.h -- EXPERIMENT (tools only, round 4; measured, not in the library): the chained Winograd transform
// (output transform of layer L -> input transform of layer L + 1, winograd_f63.hip wino_chain_kernel) in BANDS of consumer tile rows for the
// 112 x 112 planes that fill a block's LDS on their own.  Bit-identical V' (tests/test_wino_chain_gpu.py and test_baseline_shapes_gpu.py
// passed with it wired in).  VGG-16 b32, bench.py stage timers, three interleaved rounds per build on one box (tools/variant_ab.sh):
//     whole planes (product) ............. chained transforms 1.015 - 1.022 ms per step, 9 272 images/s
//     2 bands per plane ................... 1.046 - 1.054 ms, 9 224
//     4 bands behind a pooling, 2 else .... 1.073 - 1.085 ms, 9 143
//     4 bands per plane ................... 1.105 - 1.109 ms, 9 103
// More, smaller, out-of-phase blocks (5 per CU instead of 2) lose to the whole-plane block: the producer tile rows two bands share are read
// twice (+ 5 ... 10 % of M), a band's M / V' runs are half or a quarter as long, and a pooled band uses 90 of its 512 lanes in phase 2.
// To build it: include this file from winograd_f63.hip behind wino_chain_kernel and paste the host part (at the end, under #if 0) in front of
// the FHIP_CHAIN launch in winograd_output_to_next_input.
#pragma once

// ---------------------------------------------------------------------------------------------------
// The chained transform for planes too large to share a block (112 x 112: one plane = 54.7 KB of LDS, two 6-wave blocks per CU), in BANDS of
// consumer tile rows (round 4).  A block owns band b of one (image, channel) plane: the consumer's tile rows [c0, c1), hence LDS rows
// 6 c0 ... 6 c1 + 1 of the zero-bordered activation plane, hence the producer tile rows that cover activation rows 6 c0 - 1 ... 6 c1 (one or two
// producer tile rows are transformed by both neighbouring bands: + 5 ... 10 % of M reads, from L2).  Same arithmetic per value as
// wino_chain_kernel -- V' is bit-identical -- but a band is 15 - 30 KB: five blocks share a CU instead of two, a pooled 224 -> 112 boundary
// needs ONE pass over its producer tiles instead of three latency-serialised ones, and the blocks of a CU drift out of phase, so that
// loads, arithmetic and stores of different bands overlap (the whole-plane blocks ran the chip in lock step: 3.9 - 4.0 TB/s).
struct WinoChainBand
{
    int bands; // per plane
    int CB;    // consumer tile rows per band
};

template <bool HAS_BIAS, bool RELU, bool POOL>
__global__ __launch_bounds__(512, POOL ? 5 : 4) void wino_chain_band_kernel(float* __restrict__ Vn, const float* __restrict__ M, const float* __restrict__ bias,
                                                                          const WinoChain g, const WinoChainBand bd)
{
    extern __shared__ __attribute__((aligned(16))) float smem[]; // [6 CB + 2][LDW]
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int item = xcd_remap(blockIdx.x, gridDim.x);
    const int plane = item / bd.bands, b = item - plane * bd.bands;
    const int k = plane / g.N, n = plane - k * g.N;
    const int TY2 = g.T2 / g.TX2;
    const int c0 = b * bd.CB, c1 = min(c0 + bd.CB, TY2);
    if (c0 >= c1) return;
    const int rows_b = 6 * (c1 - c0) + 2;                     // LDS rows of this band: global LDS rows 6 c0 ... 6 c1 + 1
    const int A0 = max(6 * c0 - 1, 0), A1 = min(6 * c1 + 1, g.AH); // activation rows the band holds
    constexpr int RPT = POOL ? 3 : 6;                          // activation rows (and columns) a producer tile yields
    const int CW = RPT * g.TX;                                 // activation columns the producer tiles cover
    // everything phase 1 does not write -- border rows / columns, rows beyond the image -- is zeroed; the two sets are disjoint: no barrier
    for (int i = tid; i < rows_b * g.LDW; i += nthreads)
    {
        const int lr = i / g.LDW, col = i - lr * g.LDW;
        const int ay = 6 * c0 + lr - 1;
        if (ay < A0 || ay >= A1 || col < 2 || col >= 2 + CW) smem[i] = 0.f;
    }
    // ---- phase 1: the producer tiles whose rows meet [A0, A1)
    const int p0 = A0 / RPT, p1 = (A1 - 1) / RPT;
    const size_t xi_stride = g.Lm.xis;
    for (int w = tid; w < (p1 - p0 + 1) * g.TX; w += nthreads)
    {
        const int ty = p0 + w / g.TX, tx = w - (w / g.TX) * g.TX;
        const float* mp = M + (size_t)k * g.Lm.bp + g.Lm.col(n * g.T + ty * g.TX + tx);
        float m[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) m[i][j] = mp[(size_t)(i * 8 + j) * xi_stride];
        float tmp[6][8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            at6(m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j], m[6][j], m[7][j], tmp[0][j], tmp[1][j], tmp[2][j], tmp[3][j], tmp[4][j], tmp[5][j]);
        const float bv = HAS_BIAS ? bias[k] : 0.f;
        float prev0 = 0.f, prev1 = 0.f, prev2 = 0.f;
#pragma unroll
        for (int a = 0; a < 6; ++a)
        {
            float y[6];
            at6(tmp[a][0], tmp[a][1], tmp[a][2], tmp[a][3], tmp[a][4], tmp[a][5], tmp[a][6], tmp[a][7], y[0], y[1], y[2], y[3], y[4], y[5]);
#pragma unroll
            for (int bb = 0; bb < 6; ++bb)
            {
                float v = y[bb] + bv;
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
                    const int ay = 3 * ty + (a >> 1);
                    if (ay >= A0 && ay < A1)
                    {
                        float* row = smem + (size_t)(ay + 1 - 6 * c0) * g.LDW + 2 + 3 * tx;
                        row[0] = (3 * tx < g.AW) ? fmaxf(prev0, h0) : 0.f;
                        row[1] = (3 * tx + 1 < g.AW) ? fmaxf(prev1, h1) : 0.f;
                        row[2] = (3 * tx + 2 < g.AW) ? fmaxf(prev2, h2) : 0.f;
                    }
                }
                continue;
            }
            const int ay = 6 * ty + a;
            if (ay >= A0 && ay < A1)
            {
                float* row = smem + (size_t)(ay + 1 - 6 * c0) * g.LDW + 2 + 6 * tx; // even offset: 8-byte aligned pairs
#pragma unroll
                for (int bb = 0; bb < 6; bb += 2)
                {
                    const float v0 = (6 * tx + bb < g.AW) ? y[bb] : 0.f, v1 = (6 * tx + bb + 1 < g.AW) ? y[bb + 1] : 0.f;
                    *reinterpret_cast<float2*>(row + bb) = make_float2(v0, v1);
                }
            }
        }
    }
    __syncthreads();
    // ---- phase 2: the band's consumer tiles: window = band rows 6 (ty2 - c0) ... + 7, columns 6 tx2 + 1 ... + 8
    const size_t xi_stride2 = g.Lv2.xis;
    for (int w = tid; w < (c1 - c0) * g.TX2; w += nthreads)
    {
        const int tyl = w / g.TX2, tx = w - tyl * g.TX2;
        const float* lp = smem + (size_t)(6 * tyl) * g.LDW + 6 * tx + 1;
        float d[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) d[i][j] = lp[(size_t)i * g.LDW + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) bt8(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], d[6][j], d[7][j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) bt8(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5], d[i][6], d[i][7]);
        float* vp = Vn + (size_t)k * g.Lv2.bp + g.Lv2.col(n * g.T2 + (c0 + tyl) * g.TX2 + tx);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) vp[(size_t)(i * 8 + j) * xi_stride2] = d[i][j];
    }
}


#if 0 // ---- host part, as it stood in winograd_output_to_next_input (bands = (pool && g.T > 512) ? 4 : 2)
#ifndef FHIP_CHAIN_NO_BANDS // (measurement builds: whole planes only)
    if (g.ppb == 1 && plane_bytes > 32 * 1024)
    {
        // planes that fill a block's LDS on their own (112 x 112): bands of consumer tile rows (wino_chain_band_kernel) -- 2 per plane, 4 behind
        // a fused pooling (four times as many producer tiles as consumer tiles: one pass of <= 512 lanes per band)
        WinoChainBand bd;
        bd.bands = FHIP_CHAIN_BANDS_EXPR;
        bd.CB = ceil_div(pn.tiles_y, bd.bands);
        bd.bands = ceil_div(pn.tiles_y, bd.CB);
        const int rpt = pool ? 3 : 6;
        const int prod_rows = ceil_div(6 * bd.CB + 2, rpt) + 1; // producer tile rows a band meets, at most
        const int bwork = std::max(prod_rows * g.TX, bd.CB * g.TX2);
        const unsigned bthreads = (unsigned)std::min(512, std::max(64, (bwork + 63) / 64 * 64));
        const size_t blds = (size_t)(6 * bd.CB + 2) * g.LDW * sizeof(float);
        const long long bgrid = planes * bd.bands;
        if (bgrid <= 0x7fffffffLL)
        {
#define FHIP_CHAINB(B_, R_, P_) hipLaunchKernelGGL((wino_chain_band_kernel<B_, R_, P_>), dim3((unsigned)bgrid), dim3(bthreads), blds, s, vn, m, bias, g, bd)
            if (pool)
            {
                if (has_bias && relu) FHIP_CHAINB(true, true, true);
                else if (has_bias) FHIP_CHAINB(true, false, true);
                else if (relu) FHIP_CHAINB(false, true, true);
                else FHIP_CHAINB(false, false, true);
            }
            else
            {
                if (has_bias && relu) FHIP_CHAINB(true, true, false);
                else if (has_bias) FHIP_CHAINB(true, false, false);
                else if (relu) FHIP_CHAINB(false, true, false);
                else FHIP_CHAINB(false, false, false);
            }
#undef FHIP_CHAINB
            FHIP_CHECK_HIP(hipGetLastError());
            return FHIP_OK;
        }
    }
#endif
#endif
