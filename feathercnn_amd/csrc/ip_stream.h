// ip_stream.h -- InnerProduct at small batch as a weight-streaming kernel (round 3).
//
// feather::InnerProductLayer (reference src/layers/inner_product_layer.h:28-171; its GEMV, booster/avx/sgemv.cpp:317-395, is the batch-1
// case) is y[n][k] = sum_c W[k][c] x[n][c] + b[k]: at the batches the nets are run with (VGG-16 b32: fc6 = 4096 x 25088) every weight is
// used for 32 columns only, so the layer is a read-only HBM stream of W (fc6: 411 MB) -- and read-only streams reach 7.3-7.7 TB/s on this
// chip (tools/l2_probe.hip) where the LDS-tiled GEMM it ran on got 3.7-4.1 (its 64 x 32 tile still spends half its matrix-pipe time on
// set-up, barriers and epilogues of short-lived blocks).  Here:
//   * MFMA orientation D[32 output rows][32 batch columns] += A[32 x 2] B[2 x 32] (v_mfma_f32_32x32x2_f32, exact fp32);
//   * W is packed ONCE (Init) as the A-operand image wp[m-group][k-octet][lane][4]: lane l's float4 holds W[32 g + l % 32][8 q + 2 e + l / 32],
//     e = 0 .. 3 -- one coalesced 16-byte load per lane feeds four MFMAs; the activations are re-packed per call the same way
//     (xq[k-octet][lane][4] = x[l % 32][8 q + 2 e + l / 32], zero beyond the batch; 3 MB for fc6, L2-resident, read by every m-group);
//   * a wave owns one m-group (32 output rows) and one piece of the reduction; no LDS, no barriers, two independent accumulators; the next
//     trip's eight float4 are requested before this trip's sixteen MFMAs are issued (two register sets, ~110 VGPRs, 4 waves per SIMD);
//   * pieces write raw partial sums [piece][K][batch]; ip_reduce_kernel adds them in piece order (deterministic) with bias / ReLU.
// Measured (tools/ip_stream_bench.hip, batch 32): fc6 82 us (5.0 TB/s of weights; the same loop without MFMAs or activations: 66 us), fc7
// 15.8 us, fc8 6.0 us; about 3 blocks per CU is the best piece count (more pieces: more partial sums; fewer: idle CUs).
#pragma once

#include "common.h"

namespace fhip
{

struct IpStreamParams
{
    const float* wp; // [Kg][KQ][64][4]
    const float* xq; // [KQ][64][4]
    float* partial;  // [S][K][batch]
    int K, Kg, KQ, S, batch;
};

// W[K][C] -> wp[g][q][lane][e] = W[32 g + lane % 32][8 q + 2 e + lane / 32] (zero past K / C)
__global__ __launch_bounds__(256) void ip_pack_weights_kernel(float* __restrict__ wp, const float* __restrict__ w, int K, int C, int Kg, int KQ)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x; // one float4 of wp per thread
    if (i >= (long long)Kg * KQ * 64) return;
    const int lane = (int)(i & 63);
    const long long gq = i >> 6;
    const int q = (int)(gq % KQ), g = (int)(gq / KQ);
    const int m = 32 * g + (lane & 31), k0 = 8 * q + (lane >> 5);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (m < K && k0 + 2 * e < C) ? w[(size_t)m * C + k0 + 2 * e] : 0.f;
    reinterpret_cast<float4*>(wp)[i] = make_float4(v[0], v[1], v[2], v[3]);
}

// x[batch][C] -> xq[q][lane][e] = x[lane % 32][8 q + 2 e + lane / 32] (zero past batch / C).  A block transposes 8 octets (64 input features) of
// every image through LDS: global reads run along 256-byte row segments of x, global writes are whole float4.
constexpr int kIpPackOctets = 8;
__global__ __launch_bounds__(256) void ip_pack_input_kernel(float* __restrict__ xq, const float* __restrict__ x, int batch, int C, int KQ)
{
    __shared__ float t[32][kIpPackOctets * 8 + 1];
    const int q0 = blockIdx.x * kIpPackOctets, c0 = q0 * 8;
    const int col = threadIdx.x & 63, row0 = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
        const int n = 4 * i + row0, c = c0 + col;
        t[n][col] = (n < batch && c < C) ? x[(size_t)n * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
    {
        const int ql = 4 * i + row0, lane = col;
        if (q0 + ql >= KQ) continue;
        const int n = lane & 31, kb = 8 * ql + (lane >> 5);
        reinterpret_cast<float4*>(xq)[(size_t)(q0 + ql) * 64 + lane] = make_float4(t[n][kb], t[n][kb + 2], t[n][kb + 4], t[n][kb + 6]);
    }
}

#ifndef FHIP_IP_NT
#define FHIP_IP_NT 0 // 1: the weight image is requested with the non-temporal hint (every float4 of it is read once per call); measurement switch
#endif
#if FHIP_IP_NT
#define FHIP_IP_LDW(p) __builtin_nontemporal_load(p)
#else
#define FHIP_IP_LDW(p) (*(p))
#endif
#define FHIP_IP_MFMA4(A, B)                                                  \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((A).x, (B).x, acc0, 0, 0, 0); \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((A).y, (B).y, acc1, 0, 0, 0); \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((A).z, (B).z, acc0, 0, 0, 0); \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((A).w, (B).w, acc1, 0, 0, 0)

template <int UNR>
__global__ __launch_bounds__(256) void ip_stream_kernel(const IpStreamParams p)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // block = 4 consecutive m-groups of one piece (they read the same activation lines at about the same time)
    const int mgb = (p.Kg + 3) / 4;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int s = vid / mgb, g = (vid - s * mgb) * 4 + wave;
    if (g >= p.Kg) return;
    const int q_lo = (int)((long long)s * p.KQ / p.S), q_hi = (int)((long long)(s + 1) * p.KQ / p.S);
    const f32x4* a = reinterpret_cast<const f32x4*>(p.wp) + ((size_t)g * p.KQ + q_lo) * 64 + lane;
    const f32x4* b = reinterpret_cast<const f32x4*>(p.xq) + (size_t)q_lo * 64 + lane;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const int trips = (q_hi - q_lo) / UNR, tail = (q_hi - q_lo) - trips * UNR;
    if (trips > 0)
    {
        f32x4 av[UNR], bv[UNR], an[UNR], bn[UNR]; // native vectors: arrays of HIP float4 structs live across the loop end up in scratch
#pragma unroll
        for (int u = 0; u < UNR; ++u)
        {
            av[u] = FHIP_IP_LDW(a + (size_t)u * 64);
            bv[u] = b[(size_t)u * 64];
        }
        for (int t = 1; t < trips; ++t)
        {
            a += (size_t)UNR * 64;
            b += (size_t)UNR * 64;
#pragma unroll
            for (int u = 0; u < UNR; ++u)
            {
                an[u] = FHIP_IP_LDW(a + (size_t)u * 64);
                bn[u] = b[(size_t)u * 64];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
            {
                FHIP_IP_MFMA4(av[u], bv[u]);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
            {
                av[u] = an[u];
                bv[u] = bn[u];
            }
        }
        a += (size_t)UNR * 64;
        b += (size_t)UNR * 64;
#pragma unroll
        for (int u = 0; u < UNR; ++u)
        {
            FHIP_IP_MFMA4(av[u], bv[u]);
        }
    }
    for (int u = 0; u < tail; ++u)
    {
        const f32x4 a1 = FHIP_IP_LDW(a + (size_t)u * 64), b1 = b[(size_t)u * 64];
        FHIP_IP_MFMA4(a1, b1);
    }
    // C/D layout: column (batch index) = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int n = lane & 31, half = lane >> 5;
    if (n >= p.batch) return;
    float* out = p.partial + ((size_t)s * p.K + 32 * g + 4 * half) * p.batch + n;
#pragma unroll
    for (int r = 0; r < 16; ++r)
    {
        const int row = (r & 3) + 8 * (r >> 2);
        if (32 * g + 4 * half + row < p.K) out[(size_t)row * p.batch] = acc0[r] + acc1[r];
    }
}
#undef FHIP_IP_MFMA4

// out[n][m] = act(bias[m] + sum over pieces (in piece order) of partial[s][m][n]); one thread per (m, n), reads coalesced along m * batch + n
__global__ __launch_bounds__(256) void ip_reduce_kernel(float* __restrict__ out, const float* __restrict__ partial, const float* __restrict__ bias, int K,
                                                        int batch, int S, int has_bias, int relu)
{
    const int i = blockIdx.x * 256 + threadIdx.x, total = K * batch;
    if (i >= total) return;
    const int m = i / batch, n = i - m * batch;
    float acc = 0.f;
    int s = 0;
    for (; s + 4 <= S; s += 4)
    {
        const float v0 = partial[(size_t)s * total + i], v1 = partial[(size_t)(s + 1) * total + i], v2 = partial[(size_t)(s + 2) * total + i],
                    v3 = partial[(size_t)(s + 3) * total + i];
        acc = (((acc + v0) + v1) + v2) + v3;
    }
    for (; s < S; ++s) acc += partial[(size_t)s * total + i];
    if (has_bias) acc += bias[m];
    if (relu) acc = fmaxf(acc, 0.f);
    out[(size_t)n * K + m] = acc;
}

} // namespace fhip
