// wino_gemm_c64.h -- EXPERIMENT (tools only, round 4; measured, not in the library): a persistent streaming Winograd tile GEMM for C, K <= 64.
// tools/gemm_bench.hip (GEMM_C64=1), MI355X: VGG-16 conv1_2 b32 (C = K = 64, P = 46 208) 330 - 384 us with two blocks per CU (415 - 436 with one)
// against 310 - 390 us for the product's one-tile-per-block kernel in the same run; ResNet-50's 56-pixel layers (P = 6 400) 48.5 vs 44.4 us.
// The layer sits on the chip's mixed read / write HBM rate whichever way the tiles are fed; block starts are not what limits it.
#pragma once
#include "wino_gemm_glds.h"

namespace fhip
{

// ---- C <= 64, K <= 64 (VGG-16's conv1_2, ResNet-50's 56-pixel 3x3 layers): a PERSISTENT streaming form (round 4) ------------------------
// With 64 input and 64 output channels a frequency point's GEMM is 16 FLOP per byte of V + M: the layer is a stream of V in and M out that
// the matrix pipe merely keeps up with (VGG-16 conv1_2 b32: 757 MB each way).  One tile per block (gemm_core.h) pays, 23 104 times, a block
// start, a U fetch and an HBM round trip before the first MFMA; the launch ran at 4.7 TB/s where plain copies reach 5.5 - 6.  Here a block
// stays: it owns a contiguous range of the (xi, column tile) list, keeps U_xi (64 x 64 floats, 16 KB) in LDS for as long as xi does not change
// -- once or twice per block --, and streams 64 x 128 V tiles through two LDS buffers with LDS-DMA, the next tile requested before this
// tile's MFMAs.  The accumulators leave through the buffer the tile was read from (free after the barrier), so LDS is 16 + 2 x 32 KB and two
// blocks share a CU; nothing but the two barriers per tile ever waits for the memory system inside the loop.
// Waves 1 x 4: a wave owns all 64 rows x 32 columns (two accumulators), as gemm_mfma_kernel<64 x 128>.  Same k order per output: bit-identical M.
__global__ __launch_bounds__(256, 2) void wino_gemm_c64_kernel(const WinoGemmPolicy::Params prm)
{
    constexpr int BM = 64, BN = 128, CK = 64, EPI_LD = 36;
    constexpr int A_FLOATS = CK * BM, B_FLOATS = CK * BN;
    __shared__ __attribute__((aligned(16))) float lds[A_FLOATS + 2 * B_FLOATS];
    float* const As = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    const long long total = (long long)prm.batches * prm.n_tiles;
    const int first = (int)(total * blockIdx.x / gridDim.x), last = (int)(total * (blockIdx.x + 1) / gridDim.x);
    if (first >= last) return;

    // A (U_xi, [64 c][Kp = 64] floats, rows of 256 B): piece = 4 rows (16 lanes x 16 B each); wave w takes pieces w, w + 4, ...: 16 pieces
    auto issue_a = [&](int xi) {
        const float* src = prm.U + (size_t)xi * prm.Cp * prm.Kp + (size_t)(lane >> 4) * prm.Kp + (lane & 15) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const int piece = wave + 4 * i; // rows 4 piece .. 4 piece + 3
            __builtin_amdgcn_global_load_lds(src + (size_t)(4 * piece) * prm.Kp, (lds_void*)(As + piece * 4 * BM), 16, 0, 0);
        }
    };
    // B (V tile, [64 c][128] floats, rows of 512 B): piece = 2 rows (32 lanes x 16 B each); wave w takes pieces w, w + 4, ...: 32 pieces
    auto issue_b = [&](int item, int buf) {
        const int xi = item / prm.n_tiles, nt = item - xi * prm.n_tiles;
        const float* src = prm.V + (size_t)xi * prm.Lv.xis + prm.Lv.col(nt * BN) + l31 * 4;
        float* dst = lds + A_FLOATS + buf * B_FLOATS;
#pragma unroll
        for (int i = 0; i < 8; ++i)
        {
            const int piece = wave + 4 * i;
            const int r = min(2 * piece + half, prm.C - 1); // rows past C meet zero rows of U
            __builtin_amdgcn_global_load_lds(src + (size_t)r * prm.Lv.bp, (lds_void*)(dst + piece * 2 * BN), 16, 0, 0);
        }
    };

    int xi_cur = first / prm.n_tiles;
    issue_a(xi_cur);
    issue_b(first, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int a_off = half * BM + l31;
    const int b_off = half * BN + wave * 32 + l31;
    const int e_row = lane >> 3, e_c4 = (lane & 7) * 4;
    int cur = 0;
    for (int item = first; item < last; ++item)
    {
        const int xi = item / prm.n_tiles, nt = item - xi * prm.n_tiles;
        if (xi != xi_cur)
        {
            // a new frequency point: every wave is past the previous tile's reads of U (the barrier that ended the last iteration)
            xi_cur = xi;
            issue_a(xi);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const bool more = item + 1 < last;
        if (more) issue_b(item + 1, cur ^ 1);

        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float* const bbuf = lds + A_FLOATS + cur * B_FLOATS;
        const float* as = As + a_off;
        const float* bs = bbuf + b_off;
#pragma unroll 8
        for (int kp = 0; kp < CK / 2; ++kp)
        {
            const float fa0 = as[(2 * kp) * BM], fa1 = as[(2 * kp) * BM + 32], fb = bs[(2 * kp) * BN];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb, acc[1], 0, 0, 0);
        }
        // every wave is done reading this tile: its buffer becomes the transpose scratch (wave-private pieces, in-order LDS queue)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* const scr = bbuf + wave * (32 * EPI_LD);
        float* mbase = prm.M + (size_t)xi * prm.Lm.xis + prm.Lm.col(nt * BN + wave * 32 + e_c4);
#pragma unroll
        for (int i = 0; i < 2; ++i)
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * half) * EPI_LD + l31] = acc[i][r];
#pragma unroll
            for (int q = 0; q < 4; ++q)
            {
                const float4 v = *reinterpret_cast<const float4*>(&scr[(q * 8 + e_row) * EPI_LD + e_c4]);
                const int m = i * 32 + e_row + q * 8;
                if (m < prm.K) *reinterpret_cast<float4*>(mbase + (size_t)m * prm.Lm.bp) = v;
            }
        }
        // the next tile has landed (vmcnt retires in order: at most this tile's 8 stores, issued after its requests, may remain), and the
        // scratch reads are done before the tile after next is requested into this buffer
        // (rows beyond K < 64 skip their stores: then the count of stores behind the requests is not 8 -- drain everything)
        if (prm.K == BM)
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }
}

} // namespace fhip
