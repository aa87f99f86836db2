// stream_gemm.h -- 1x1 / stride-1 convolution as a register-streamed GEMM: no LDS, no barriers.
//
// Same contraction as the implicit GEMM of implicit_gemm.hip (reference: IM2COL_Forward, avx/booster.cpp:83-102 -- for a 1x1 kernel
// the column matrix IS the input -- followed by packed_sgemm_activation, avx/sgemm.cpp:381-432), organised for layers with a deep
// reduction and a narrow output (ResNet-50's C -> C/4 layers, MobileNet's 512 -> 512): there the LDS-tiled kernel spends its time in
// block prologues / epilogues and split-K, while a wave that simply streams its operands keeps the matrix pipe fed.
//
//   * a wave owns 32 output channels x 128 consecutive pixels (4 per lane; pixels are numbered across the whole batch, a group of 4
//     never straddles two images because Ho*Wo % 4 == 0);
//   * per PAIR of input channels every lane loads one float4 of the activation: lanes 0-31 channel 2j, lanes 32-63 channel 2j+1,
//     pixels 4l .. 4l+3 -- component t of it is the B operand of MFMA t (v_mfma_f32_32x32x2_f32: B[k = lane / 32][n = lane % 32]),
//     so the four MFMAs of a step compute the four interleaved pixel sets {4n + t} and a 16-byte load feeds 4 x 64 MFMA cycles;
//   * the weights are pre-packed as the A-operand image wp[m-group][channel pair][lane] = W[32 g + lane % 32][2 j + lane / 32]:
//     one coalesced dword per lane and step (L2 / L1 resident: <= 2 MB per layer);
//   * the accumulators hold, per register, four CONSECUTIVE pixels across (acc0 .. acc3): the result leaves as 16 dwordx4 stores,
//     512 contiguous bytes per half-wave, bias + ReLU applied on the way -- no transpose, no LDS;
//   * loads are inline asm with counted waits: hipcc sinks ordinary loads towards their uses (it kept 2 of 8 float4 in flight).
//     vmcnt retires in order, so with P requests outstanding `s_waitcnt vmcnt(P - 2)` says the oldest two (one B float4, one A dword)
//     have landed; the registers are operands of the wait, so their uses cannot be hoisted above it.  What the language does NOT
//     guarantee is that hipcc leaves a destination alone between its load and that wait (a register copy or a spill there would
//     read stale data): that obligation is checked on the generated code of the shipped library by tools/check_stream_isa.py
//     (a vmcnt-queue replay of every instantiation: no instruction touches a register with a load in flight, no scratch
//     instructions), run by tests/test_boundary.py on every build -- re-run it after any compiler change.
//
// Measured against gemm_mfma_kernel<128x64, ConvGemmPolicy<2>> (tools/stream_bench.hip, MI355X): 1024 -> 256 @14x14 b64 70 vs 80 us
// (+ 12 us of split-K reduce), 512 -> 128 @28x28 b64 72 vs 79, 512 -> 512 @14x14 b256 231 vs 240, 256 -> 256 @28x28 b256 232 vs 250;
// slower on shallow reductions (C <= 128) and on wide outputs (K >= 1024: every m-group re-reads the activation), which stay on
// the LDS-tiled kernel (stream_profitable).
#pragma once
#include "common.h"

namespace fhip
{
typedef float st_f32x16 __attribute__((ext_vector_type(16)));
typedef float st_f32x4 __attribute__((ext_vector_type(4)));

struct StreamGemmParams
{
    const float* in;   // [N][C][HW]
    const float* wp;   // [K / 32][C / 2][64]
    const float* bias; // [K] (read only when the kernel is instantiated with BIAS)
    float* out;        // [N][K][HW]
    int C, K, HW;
    long long total_px; // N * HW  (RAGGED: N * 4 * ceil(HW / 4) pixel SLOTS)
    int mgroups;        // K / 32
    int px_tiles;       // ceil(total_px / 128)
    int gpi;            // RAGGED: pixel groups per image = ceil(HW / 4)
};

__global__ __launch_bounds__(256) void stream_pack_weights_kernel(float* __restrict__ wp, const float* __restrict__ w, int K, int C)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)K * C) return;
    const int lane = (int)(i & 63);
    const long long rest = i >> 6;
    const int J = C / 2;
    const int j = (int)(rest % J), mg = (int)(rest / J);
    wp[i] = w[(size_t)(32 * mg + (lane & 31)) * C + 2 * j + (lane >> 5)];
}

#define FHIP_ST_LD4(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr))
#define FHIP_ST_LD1(dst, ptr) asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(ptr))
#define FHIP_ST_WAIT(n, b_, a_) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b_), "+v"(a_) : "n"(n))
#define FHIP_ST_MFMA4(a_, b_)                                                    \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_.x, acc[0], 0, 0, 0);    \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_.y, acc[1], 0, 0, 0);    \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_.z, acc[2], 0, 0, 0);    \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_.w, acc[3], 0, 0, 0)

// D = depth of the request ring (steps in flight per wave); C / 2 must be a multiple of D.  Block = 4 waves = 4 consecutive m-groups of
// one pixel tile (they read the same activation lines at about the same time: L1 / L2 hits); no wave ever waits for another.
// RAGGED: Ho*Wo is not a multiple of 4 (ResNet-50's 7x7 stage: 49).  Every image then has ceil(HW / 4) pixel groups; the last one
// would run past the image, so it is loaded SHIFTED BACK to the image's last four pixels (all loads stay inside the tensor) and only
// its new pixels -- the trailing components -- are stored, one dword each.  Rows are then only 4-byte aligned: the float4 loads and
// stores are unaligned, which the hardware takes.
template <int D, bool BIAS, bool RELU, bool RAGGED = false>
__global__ __launch_bounds__(256) void stream_gemm_kernel(const StreamGemmParams q)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mg_blocks = (q.mgroups + 3) / 4;
    const int vid = xcd_remap(blockIdx.x, gridDim.x); // the m-groups of one pixel tile land on ONE XCD (one L2 fetches the tile)
    const int pt = vid / mg_blocks, mg = (vid - pt * mg_blocks) * 4 + wave;
    if (mg >= q.mgroups) return;
    const int half = lane >> 5, l31 = lane & 31;
    const long long g = (long long)pt * 128 + 4 * l31;
    const bool ok = g < q.total_px;
    const long long gc = ok ? g : 0; // lanes beyond the tensor stream pixel group 0 and store nothing
    int n, p, first_new = 0;         // first_new: first component of this group that no earlier group covers
    if (RAGGED)
    {
        const long long grp = gc >> 2;
        n = (int)(grp / q.gpi);
        const int gi = (int)(grp - (long long)n * q.gpi);
        p = min(4 * gi, q.HW - 4);
        first_new = 4 * gi - p;
    }
    else
    {
        n = (int)(gc / q.HW);
        p = (int)(gc - (long long)n * q.HW);
    }
    const float* bp = q.in + ((size_t)n * q.C + half) * q.HW + p;
    const float* ap = q.wp + (size_t)mg * (q.C / 2) * 64 + lane;
    const size_t bstep = (size_t)2 * q.HW;
    const int J = q.C / 2;

    float bs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bs[r] = BIAS ? q.bias[32 * mg + 4 * half + (r & 3) + 8 * (r >> 2)] : 0.f;

    st_f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    st_f32x4 b[D];
    float a[D];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the bias loads: from here on every outstanding request is one of the ring's
#pragma unroll
    for (int u = 0; u < D; ++u)
    {
        FHIP_ST_LD4(b[u], bp + (size_t)u * bstep);
        FHIP_ST_LD1(a[u], ap + (size_t)u * 64);
    }
    const float* bnext = bp + (size_t)D * bstep;
    const float* anext = ap + (size_t)D * 64;
    for (int j0 = 0; j0 < J - D; j0 += D)
    {
#pragma unroll
        for (int u = 0; u < D; ++u)
        {
            FHIP_ST_WAIT(2 * D - 2, b[u], a[u]);
            FHIP_ST_MFMA4(a[u], b[u]);
            FHIP_ST_LD4(b[u], bnext + (size_t)u * bstep);
            FHIP_ST_LD1(a[u], anext + (size_t)u * 64);
        }
        bnext += (size_t)D * bstep;
        anext += (size_t)D * 64;
    }
    // the last D steps: nothing left to request, the queue drains
#pragma unroll
    for (int u = 0; u < D; ++u)
    {
        FHIP_ST_WAIT(2 * (D - u) - 2, b[u], a[u]);
        FHIP_ST_MFMA4(a[u], b[u]);
    }
    if (!ok) return;
    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* op = q.out + ((size_t)n * q.K + 32 * mg + 4 * half) * q.HW + p;
#pragma unroll
    for (int r = 0; r < 16; ++r)
    {
        const int row = (r & 3) + 8 * (r >> 2);
        float4 v = make_float4(acc[0][r] + bs[r], acc[1][r] + bs[r], acc[2][r] + bs[r], acc[3][r] + bs[r]);
        if (RELU)
        {
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f);
        }
        float* o = op + (size_t)row * q.HW;
        if (!RAGGED || first_new == 0)
            stg4_act<1>(o, v);
        else
        {
            if (first_new <= 1) o[1] = v.y;
            if (first_new <= 2) o[2] = v.z;
            o[3] = v.w;
        }
    }
}
#undef FHIP_ST_LD4
#undef FHIP_ST_LD1
#undef FHIP_ST_WAIT
#undef FHIP_ST_MFMA4
} // namespace fhip
