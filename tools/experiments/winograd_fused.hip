// winograd_fused.hip -- Winograd F(6x6,3x3): tile GEMM + output transform in ONE kernel, for the layers whose
// non-fused pipeline is HBM-bound (C, K <= 128: arithmetic intensity 2KC/(4(K+C)) <= 32 FLOP/B against a ridge of
// ~20-25 FLOP/B, and the M round trip is 2 x 64*K*P*4 bytes -- 1.5 GB for VGG-16 conv1_2 at batch 32).
// Same math as winograd_f63.hip (reference TensorGEMM avx/winograd_kernels_F63.cpp:518-692 followed by
// winogradOutputTransform<relu,bias> :1088-1269); what changes is that M never exists in HBM:
//
//   * a block owns 32 output channels x 32 tiles (columns p) for ALL 64 frequency points xi; its 8 waves own 8 xi each
//     (one row i of the 8x8 frequency grid), i.e. 8 independent 32x32 fp32-MFMA accumulators = 128 VGPRs per lane,
//     2 waves per SIMD, one block per CU;
//   * the reduction over input channels runs in chunks of 4: U[xi][c..c+3][32 k] and V[xi][c..c+3][32 p] for all 64 xi
//     are 32 KB each in LDS, double buffered (128 KB), prefetched two chunks ahead through registers exactly like
//     gemm_core.h (unconditional loads, LDS write at the top of the next iteration, one barrier per chunk);
//   * after the last chunk the accumulators are exchanged through LDS (two passes of 16 output channels: 64 xi x 16 x
//     32 floats = 128 KB, the operand buffers are dead by then), every thread gathers the 64 xi of one (k, tile),
//     applies A^T m A, bias, ReLU and stores its clipped 6x6 block.
#include "common.h"

namespace fhip
{

__device__ __forceinline__ void at6f(float m0, float m1, float m2, float m3, float m4, float m5, float m6, float m7, float& s0,
                                     float& s1, float& s2, float& s3, float& s4, float& s5)
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

struct WinoFusedParams
{
    const float* U; // [64][Cp][Kp]
    const float* V; // [64][C][Pp]
    const float* bias;
    float* out;
    int C, K, Cp, Kp, P, Pp;
    int OH, OW, TX, T;
    int k_groups, p_groups, chunks; // K/32 (rounded up), P/32 (rounded up), ceil(C/4)
};

constexpr int kFusedCB = 4;                          // input channels per chunk
constexpr int kFusedBuf = 64 * kFusedCB * 32;        // floats per operand buffer (32 KB)
constexpr int kFusedLds = 4 * kFusedBuf;             // U, V double buffered = 128 KB = the exchange buffer

template <bool HAS_BIAS, bool RELU>
__global__ __launch_bounds__(512, 2) void wino_fused_gemm_output_kernel(const WinoFusedParams q)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Us0 = lds;                 // Us[buf][xi][c][32 k]
    float* const Vs0 = lds + 2 * kFusedBuf; // Vs[buf][xi][c][32 p]

    const int nwg = q.k_groups * q.p_groups;
    int vid = xcd_remap(blockIdx.x, nwg);
    const int kg = vid % q.k_groups, pg = vid / q.k_groups;
    const int k0 = kg * 32, p0 = pg * 32;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    // loader mapping: float4 number f = tid + 512*i (i = 0..3) of a [64*4 rows][32 floats] operand chunk:
    // row = f / 8 = (xi, c), piece = f % 8
    const int piece = tid & 7;
    const float* usrc[4];
    const float* vsrc[4];
    int vc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const int row = (tid >> 3) + 64 * i; // 0..255
        const int xi = row >> 2, c = row & 3;
        vc[i] = c;
        usrc[i] = q.U + ((size_t)xi * q.Cp + c) * q.Kp + k0 + 4 * piece;
        vsrc[i] = q.V + (size_t)xi * q.C * q.Pp + p0 + 4 * piece; // + channel * Pp at load time (clamped)
    }
    // (plain named registers + macros rather than arrays captured by lambdas: with arrays hipcc kept the prefetch
    //  registers in scratch memory here)
    float4 pu0, pu1, pu2, pu3, pv0, pv1, pv2, pv3;
#define FUSED_FETCH1(I, PU, PV)                                                                                  \
    PU = *reinterpret_cast<const float4*>(usrc[I] + (size_t)c0_ * q.Kp); /* U is zero padded to Cp >= 4*chunks */ \
    PV = *reinterpret_cast<const float4*>(vsrc[I] + (size_t)min(c0_ + vc[I], q.C - 1) * q.Pp);
    // rows past C re-read row C-1; they meet all-zero rows of U, so they contribute exactly 0
#define FUSED_FETCH(CHUNK)                 \
    {                                      \
        const int c0_ = (CHUNK)*kFusedCB;  \
        FUSED_FETCH1(0, pu0, pv0)          \
        FUSED_FETCH1(1, pu1, pv1)          \
        FUSED_FETCH1(2, pu2, pv2)          \
        FUSED_FETCH1(3, pu3, pv3)          \
    }
#define FUSED_STASH1(I, PU, PV)                                                              \
    *reinterpret_cast<float4*>(Us0 + (buf_)*kFusedBuf + 4 * (tid + 512 * I)) = PU;           \
    *reinterpret_cast<float4*>(Vs0 + (buf_)*kFusedBuf + 4 * (tid + 512 * I)) = PV;
#define FUSED_STASH(BUF)            \
    {                               \
        const int buf_ = (BUF);     \
        FUSED_STASH1(0, pu0, pv0)   \
        FUSED_STASH1(1, pu1, pv1)   \
        FUSED_STASH1(2, pu2, pv2)   \
        FUSED_STASH1(3, pu3, pv3)   \
    }

    f32x16 acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;

    FUSED_FETCH(0)
    FUSED_STASH(0)
    if (q.chunks > 1) FUSED_FETCH(1)
    __syncthreads();

    // operand fragment of xi = 8*wave + e, channel pair cp: row (xi*4 + 2*cp + half), element l31
    const int frag = (wave * 8 * kFusedCB + half) * 32 + l31;
    int cur = 0;
    for (int ch = 0; ch < q.chunks; ++ch)
    {
        if (ch + 1 < q.chunks) FUSED_STASH(cur ^ 1)
        if (ch + 2 < q.chunks) FUSED_FETCH(ch + 2)
        const float* us = Us0 + cur * kFusedBuf + frag;
        const float* vs = Vs0 + cur * kFusedBuf + frag;
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int cp = 0; cp < kFusedCB / 2; ++cp)
            {
                const float a = us[(e * kFusedCB + 2 * cp) * 32];
                const float b = vs[(e * kFusedCB + 2 * cp) * 32];
                acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[e], 0, 0, 0);
            }
        __syncthreads();
        cur ^= 1;
    }

#undef FUSED_FETCH
#undef FUSED_FETCH1
#undef FUSED_STASH
#undef FUSED_STASH1
    // ---- exchange + output transform, two passes of 16 output channels (accumulator registers 8h .. 8h+7 hold rows
    // (rr & 3) + 8 * (rr >> 2) + 4 * half + 16 h of the 32x32 tile, column = l31)
    float* const Ms = lds; // [64 xi][16 rows][32 tiles]
    const int o_row = tid >> 5, o_t = tid & 31;
    const int p = p0 + o_t;
    const bool p_ok = p < q.P;
    const int pc = p_ok ? p : 0;
    const int n = pc / q.T, t = pc - n * q.T;
    const int ty = t / q.TX, tx = t - ty * q.TX;
    const int oy0 = ty * 6, ox0 = tx * 6;
#pragma unroll
    for (int h = 0; h < 2; ++h)
    {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr)
                Ms[(wave * 8 + e) * 512 + ((rr & 3) + 8 * (rr >> 2) + 4 * half) * 32 + l31] = acc[e][8 * h + rr];
        __syncthreads();
        const int k = k0 + 16 * h + o_row;
        float m[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) m[i][j] = Ms[(i * 8 + j) * 512 + tid];
        float tmp[6][8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            at6f(m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j], m[6][j], m[7][j], tmp[0][j], tmp[1][j], tmp[2][j], tmp[3][j],
                 tmp[4][j], tmp[5][j]);
        if (p_ok && k < q.K)
        {
            const float b = HAS_BIAS ? q.bias[k] : 0.f;
            float* op = q.out + (((size_t)n * q.K + k) * q.OH + oy0) * q.OW + ox0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
            {
                float y[6];
                at6f(tmp[a][0], tmp[a][1], tmp[a][2], tmp[a][3], tmp[a][4], tmp[a][5], tmp[a][6], tmp[a][7], y[0], y[1], y[2], y[3],
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
        __syncthreads();
    }
}

int winograd_plan(const fhip_conv_param& p, int batch, fhip_winograd_plan* plan);

// Tile GEMM + output transform fused; V as produced by the input transform, U as produced by Init.
int winograd_fused_gemm_output(const fhip_conv_param& p, int batch, float* output, const float* u, const float* v, const float* bias,
                               hipStream_t s)
{
    fhip_winograd_plan pl;
    int rc = winograd_plan(p, batch, &pl);
    if (rc) return rc;
    const bool has_bias = p.bias_term != 0, relu = p.activation == FHIP_ACT_RELU;
    if (has_bias && !bias) return fail(FHIP_E_BADARG, "bias_term set but bias_arr is NULL");
    WinoFusedParams q;
    q.U = u;
    q.V = v;
    q.bias = bias;
    q.out = output;
    q.C = p.input_channels;
    q.K = p.output_channels;
    q.Cp = pl.in_channels_padded;
    q.Kp = pl.out_channels_padded;
    q.P = pl.columns;
    q.Pp = pl.columns_padded;
    q.OH = p.output_h;
    q.OW = p.output_w;
    q.TX = pl.tiles_x;
    q.T = pl.tiles_per_image;
    q.k_groups = ceil_div(q.K, 32);
    q.p_groups = ceil_div(q.P, 32);
    q.chunks = ceil_div(q.C, kFusedCB);
    const size_t lds = (size_t)kFusedLds * sizeof(float);
    static bool attr_done = false;
    if (!attr_done)
    {
        // 128 KiB of dynamic LDS needs the opt-in attribute (default cap 64 KiB)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_fused_gemm_output_kernel<true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_fused_gemm_output_kernel<true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_fused_gemm_output_kernel<false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_fused_gemm_output_kernel<false, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    StageTimer tm(FHIP_STAGE_WINO_GEMM, s);
    const dim3 grid(q.k_groups * q.p_groups), block(512);
    if (has_bias && relu)
        hipLaunchKernelGGL((wino_fused_gemm_output_kernel<true, true>), grid, block, lds, s, q);
    else if (has_bias)
        hipLaunchKernelGGL((wino_fused_gemm_output_kernel<true, false>), grid, block, lds, s, q);
    else if (relu)
        hipLaunchKernelGGL((wino_fused_gemm_output_kernel<false, true>), grid, block, lds, s, q);
    else
        hipLaunchKernelGGL((wino_fused_gemm_output_kernel<false, false>), grid, block, lds, s, q);
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

} // namespace fhip
