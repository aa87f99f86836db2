// stream_split.h -- EXPERIMENT (tools only, round 4): the register-streamed 1x1 GEMM of stream_gemm_exp.h with the reduction split over
// ANY number of waves.
//
// Why: a wave tile is 32 output channels x 128 pixels, so ResNet-50 b64's bottleneck layers decompose into 784 (1024 -> 256 @14x14) or
// 1568 (512 -> 128 @28x28) waves for the chip's 1024 SIMDs -- 0.77 of them busy on average, whatever the kernel does inside the wave, and
// a 2- or 4-way split (round 2's experiment) keeps exactly that ratio (1568 / 2048, 3136 / 4096).  An S-way split with S = 3, 5, 7 ... makes
// tiles * S land just under a multiple of the SIMD count, and the S-times shorter waves shrink what the last partial round costs.
//
// A block = MGB m-groups x S pieces of one pixel tile (MGB * S waves).  Wave (g, cp) reduces chunks [cp * Q / S, (cp + 1) * Q / S) of the
// Q = C / 2 / D request-ring chunks of its m-group (pieces differ by at most one chunk), then the S partial accumulators of an m-group are
// added in the FIXED order 0 .. S-1 through LDS -- in two passes of 8 accumulator registers (8 KB per wave and pass) -- every wave
// finishing the rows r = cp, cp + S, ... of a pass: bias, ReLU, one dwordx4 store per row and lane as in the unsplit kernel.
#pragma once
#include "stream_gemm_exp.h"

namespace fhip
{

template <int D, int S, int MGB>
__global__ __launch_bounds__(64 * S * MGB, 4) void stream_pw_splitn_kernel(const StreamParams q)
{
    extern __shared__ __attribute__((aligned(16))) float red[]; // [MGB][S][8 regs][64 lanes] float4
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = wave / S, cp = wave - grp * S;
    const int mg_blocks = q.mgroups / MGB;
    int vid = blockIdx.x;
    {
        const int nwg = gridDim.x, qx = nwg / 8, rx = nwg % 8, xcd = vid % 8, local = vid / 8;
        vid = ((xcd < rx) ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + local;
    }
    const int pt = vid / mg_blocks, mg = (vid - pt * mg_blocks) * MGB + grp;
    const int half = lane >> 5, l31 = lane & 31;
    const long long g = (long long)pt * 128 + 4 * l31;
    const bool ok = g < q.total_px;
    const long long gc = ok ? g : 0;
    const int n = (int)(gc / q.HW), p = (int)(gc - (long long)n * q.HW);
    const int Q = q.C / 2 / D;                                 // chunks of D steps
    const int c0 = cp * Q / S, c1 = (cp + 1) * Q / S;          // this wave's chunks (Q >= S: at least one)
    const int J = (c1 - c0) * D, j_first = c0 * D;
    const float* bp = q.in + ((size_t)n * q.C + half + (size_t)2 * j_first) * q.HW + p;
    const float* ap = q.wp + ((size_t)mg * (q.C / 2) + (size_t)j_first) * 64 + lane;
    const size_t bstep = (size_t)2 * q.HW;

    f32x16s acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4s b[D];
    float a[D];
#pragma unroll
    for (int u = 0; u < D; ++u)
    {
        STREAM_LD4(b[u], bp + (size_t)u * bstep);
        STREAM_LD1(a[u], ap + (size_t)u * 64);
    }
    const float* bnext = bp + (size_t)D * bstep;
    const float* anext = ap + (size_t)D * 64;
    for (int j0 = 0; j0 < J - D; j0 += D)
    {
#pragma unroll
        for (int u = 0; u < D; ++u)
        {
            STREAM_WAIT(2 * D - 2, b[u], a[u]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].w, acc[3], 0, 0, 0);
            STREAM_LD4(b[u], bnext + (size_t)u * bstep);
            STREAM_LD1(a[u], anext + (size_t)u * 64);
        }
        bnext += (size_t)D * bstep;
        anext += (size_t)D * 64;
    }
#pragma unroll
    for (int u = 0; u < D; ++u)
    {
        STREAM_WAIT(2 * (D - u) - 2, b[u], a[u]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].w, acc[3], 0, 0, 0);
    }

    // ---- fixed-order reduction through LDS, two passes of 8 accumulator registers
    float4* const slots = reinterpret_cast<float4*>(red) + (size_t)grp * S * 8 * 64;
    float* const op = q.out + ((size_t)n * q.K + 32 * mg + 4 * half) * q.HW + p;
    const float* const bsp = q.bias + 32 * mg + 4 * half;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
    {
        if (pass) __syncthreads(); // pass 0's readers are done with the slots
#pragma unroll
        for (int r = 0; r < 8; ++r)
        {
            const int rr = pass * 8 + r;
            slots[(cp * 8 + r) * 64 + lane] = make_float4(acc[0][rr], acc[1][rr], acc[2][rr], acc[3][rr]);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r)
        {
            if ((r % S) != cp) continue; // wave-uniform
            float4 v = slots[(0 * 8 + r) * 64 + lane];
#pragma unroll
            for (int s = 1; s < S; ++s)
            {
                const float4 w = slots[(s * 8 + r) * 64 + lane];
                v.x += w.x;
                v.y += w.y;
                v.z += w.z;
                v.w += w.w;
            }
            const int rr = pass * 8 + r;
            const int row = (rr & 3) + 8 * (rr >> 2);
            const float bs = bsp[row];
            v.x += bs;
            v.y += bs;
            v.z += bs;
            v.w += bs;
            if (q.relu)
            {
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
            }
            if (ok) *reinterpret_cast<float4*>(op + (size_t)row * q.HW) = v;
        }
    }
}

template <int D, int S, int MGB>
static void launch_splitn(const StreamParams& q)
{
    const int mg_blocks = q.mgroups / MGB;
    const size_t lds = (size_t)MGB * S * 8 * 64 * 16;
    hipLaunchKernelGGL((stream_pw_splitn_kernel<D, S, MGB>), dim3((unsigned)(q.px_tiles * mg_blocks)), dim3(64 * S * MGB), lds, 0, q);
}

} // namespace fhip
