// gemm_core_probe.h -- MEASUREMENT BUILDS ONLY (tools/gemm_bench.hip, tools/flat_bench.hip): the product's fp32-MFMA main loop
// (feathercnn_amd/csrc/gemm_core.h) with the ablation switches and per-block clock stamps the investigations of DESIGN.md 3.4 used.
// Never included by the library.  Keep its loop in step with gemm_core.h when that changes.
#pragma once

#include "gemm_core.h" // GemmShape, the policy concept

namespace fhip
{

// ABLATE (measurement builds only, tools/gemm_bench.hip; the product always uses 0):
//   bit 0: no global fetch after the prologue (MFMA + LDS only; results are garbage), bit 1: no accumulator store.
#ifdef FHIP_TIMELINE
// measurement builds only (tools/flat_bench.hip, -DFHIP_TIMELINE, ABLATE bit 5): shader-clock stamps of sampled blocks
static __device__ long long g_core_timeline[64][16];
// ABLATE bit 6: EVERY block logs [hw id | 100 MHz wall clock at start, set-up done, k-tile 0 in LDS, k-loop done, end] -> g_core_blocklog[block][8]
// (the life cycle of all blocks of a launch, grouped per CU on the host: tools/r50_probe.hip)
static __device__ long long* g_core_blocklog;
static __device__ long long* g_core_iterlog;
#endif

// TUNE (round 2, from per-block clock stamps -- tools/flat_bench.hip FLAT_CORE=1: a 128x64x64 tile of ResNet-50's 1x1 layers spent
// ~7000 cycles in its index set-up, ~16000 in the k-loop and 10000-30000 in the epilogue):
//   bit 0: the block's set-up runs at raised wave priority.  The SIMD arbitrates issue slots by priority, then AGE: a freshly
//          launched wave is the youngest on its SIMD and only gets the slots the older, MFMA-issuing waves leave over, so the few
//          hundred VALU instructions in front of its first load took microseconds -- with every other block's prologue latency
//          behind them;
//   bit 1: the lane's bias values are requested in the prologue (Policy::bias_at) instead of one dependent global load in front of
//          every accumulator store (8 serialised L2 round trips per wave and tile).
// Measured on ResNet-50 / MobileNet 1x1 layers: +3 ... +7 % on the shallow ones (C <= 128), nothing on the deep ones.  Two other
// epilogues were measured and dropped: storing straight from the accumulators of an MFMA with swapped operand
// roles (32-byte store pieces: -20 %) and batching the LDS transpose of a whole 32-column piece (block latency -3000 cycles,
// throughput unchanged).
template <class Shape, class Policy, int ABLATE = 0, int TUNE = 3>
__global__ __launch_bounds__(Shape::THREADS, Shape::BLOCKS_PER_CU* Shape::THREADS / 256) void gemm_mfma_probe_kernel(
    const typename Policy::Params prm)
{
    constexpr int BM = Shape::BM, BN = Shape::BN, BK = Shape::BK;
    // ONE LDS object (a second __shared__ object de-pipelines hipcc's waits)
    __shared__ __attribute__((aligned(16))) float lds[Shape::LDS_FLOATS + Policy::EXTRA_LDS_FLOATS];
    float* const extra = lds + Shape::LDS_FLOATS; // Policy::stage_extra's block-wide constants (behind the operand / epilogue area)
    float* const As0 = lds;               // As[buf] = As0 + buf * BK*BM
    float* const Bs0 = lds + 2 * BK * BM; // Bs[buf] = Bs0 + buf * BK*BN

#ifdef FHIP_TIMELINE
    const bool tl_on = (ABLATE & 32) && (blockIdx.x % 97) == 5 && blockIdx.x / 97 < 64 && threadIdx.x == 0;
    int tl_n = 0;
    auto stamp = [&]() {
        if ((ABLATE & 32) && tl_on && tl_n < 16) g_core_timeline[blockIdx.x / 97][tl_n] = clock64();
        if ((ABLATE & 64) && threadIdx.x == 0)
        {
            long long* lg = g_core_blocklog + (size_t)blockIdx.x * 8;
            if (tl_n == 0)
            {
                lg[0] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4); // XCC_ID, HW_ID
                lg[1] = wall_clock64();
            }
            else if (tl_n == 1) lg[2] = wall_clock64();
            else if (tl_n == 2) lg[3] = wall_clock64();
            else if (tl_n == 7) lg[4] = wall_clock64();
            else if (tl_n == 15) lg[5] = wall_clock64();
        }
        ++tl_n;
    };
#else
    int tl_n = 0;
    auto stamp = [&]() {};
    (void)tl_n;
#endif
    stamp();
    if (TUNE & 1) __builtin_amdgcn_s_setprio(3);
    const int nwg = prm.batches * prm.m_tiles * prm.n_tiles;
    int vid = xcd_remap(blockIdx.x, nwg);
    const int mt = vid % prm.m_tiles;
    vid /= prm.m_tiles;
    const int nt = vid % prm.n_tiles;
    const int batch = vid / prm.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int k_tiles = Policy::k_count(prm, batch); // tiles of THIS batch entry (a split-K piece may be uneven)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Shape::WAVES_N, wn = wave % Shape::WAVES_N;
    const int l31 = lane & 31, half = lane >> 5;

    // loader mapping: a thread always fetches the same 4 consecutive m (n) of rows r, r + ROWS_PER_PASS, ...
    const int a_c4 = tid % Shape::A_F4_PER_ROW, a_r = tid / Shape::A_F4_PER_ROW;
    const int b_c4 = tid % Shape::B_F4_PER_ROW, b_r = tid / Shape::B_F4_PER_ROW;
    const typename Policy::ALoad aload(prm, batch, m0 + a_c4 * 4);
    const typename Policy::BLoad bload(prm, batch, n0 + b_c4 * 4);

    typedef typename Policy::BLoad::Raw BRaw;
    float4 pa[Shape::A_PASSES];
    BRaw pb[Shape::B_PASSES];
    unsigned pok[Shape::B_PASSES];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i) pa[i] = aload.load(prm, kt * BK + a_r + i * Shape::A_ROWS_PER_PASS);
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i) pb[i] = bload.load(prm, kt * BK + b_r + i * Shape::B_ROWS_PER_PASS, pok[i]);
    };
    auto stash = [&](int buf, int kt) {
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i)
            *reinterpret_cast<float4*>(&As0[buf * (BK * BM) + (a_r + i * Shape::A_ROWS_PER_PASS) * BM + a_c4 * 4]) = pa[i];
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i)
        {
            float4 v = bload.finish(prm, pb[i], kt * BK + b_r + i * Shape::B_ROWS_PER_PASS, extra);
            v.x = (pok[i] & 1u) ? v.x : 0.f;
            v.y = (pok[i] & 2u) ? v.y : 0.f;
            v.z = (pok[i] & 4u) ? v.z : 0.f;
            v.w = (pok[i] & 8u) ? v.w : 0.f;
            *reinterpret_cast<float4*>(&Bs0[buf * (BK * BN) + (b_r + i * Shape::B_ROWS_PER_PASS) * BN + b_c4 * 4]) = v;
        }
    };

    // ABLATE bit 8 (DEEP): a second register set, so the global loads of a k-tile have TWO iterations to land (a block whose iteration is
    // shorter than the load latency is latency-bound whatever the matrix pipe does)
    constexpr bool DEEP = (ABLATE & 256) != 0;
    float4 pa2[Shape::A_PASSES];
    BRaw pb2[Shape::B_PASSES];
    unsigned pok2[Shape::B_PASSES];
    auto fetch2 = [&](int kt) {
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i) pa2[i] = aload.load(prm, kt * BK + a_r + i * Shape::A_ROWS_PER_PASS);
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i) pb2[i] = bload.load(prm, kt * BK + b_r + i * Shape::B_ROWS_PER_PASS, pok2[i]);
    };
    auto stash2 = [&](int buf, int kt) {
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i)
            *reinterpret_cast<float4*>(&As0[buf * (BK * BM) + (a_r + i * Shape::A_ROWS_PER_PASS) * BM + a_c4 * 4]) = pa2[i];
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i)
        {
            float4 v = bload.finish(prm, pb2[i], kt * BK + b_r + i * Shape::B_ROWS_PER_PASS, extra);
            v.x = (pok2[i] & 1u) ? v.x : 0.f;
            v.y = (pok2[i] & 2u) ? v.y : 0.f;
            v.z = (pok2[i] & 4u) ? v.z : 0.f;
            v.w = (pok2[i] & 8u) ? v.w : 0.f;
            *reinterpret_cast<float4*>(&Bs0[buf * (BK * BN) + (b_r + i * Shape::B_ROWS_PER_PASS) * BN + b_c4 * 4]) = v;
        }
    };

    f32x16 acc[Shape::TM][Shape::TN];
#pragma unroll
    for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
        for (int j = 0; j < Shape::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: k-tile 0 -> LDS buffer 0, k-tile 1 -> registers.  Both tiles' loads are issued back to back
    // (a second register set for a moment), so the block pays ONE global round trip before its first MFMA, not two.
    stamp(); // setup done
    fetch(0);
    if (Policy::EXTRA_LDS_FLOATS > 0)
    {
        Policy::stage_extra(prm, extra, tid, Shape::THREADS); // behind the first operand requests; the first finish() needs it
        __syncthreads();
    }
    // bias of the rows this lane will store (row = .. + i*32 + q*8 + (lane >> 3)), requested behind the first operand tile
    float bias_r[Shape::TM][4];
#pragma unroll
    for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            bias_r[i][q] = (TUNE & 2) ? Policy::bias_at(prm, m0 + wm * Shape::WTM + i * 32 + q * 8 + (lane >> 3)) : 0.f;
    if (DEEP)
    {
        // set 1 (pa) holds tile 0 now; tile 1 -> set 2, tile 0 -> LDS, tile 2 -> set 1
        if (k_tiles > 1) fetch2(1);
        stash(0, 0);
        if (k_tiles > 2) fetch(2);
    }
    else if (k_tiles > 1)
    {
        float4 qa[Shape::A_PASSES];
        BRaw qb[Shape::B_PASSES];
        unsigned qok[Shape::B_PASSES];
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i) qa[i] = aload.load(prm, BK + a_r + i * Shape::A_ROWS_PER_PASS);
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i) qb[i] = bload.load(prm, BK + b_r + i * Shape::B_ROWS_PER_PASS, qok[i]);
        stash(0, 0); // waits for tile 0's loads only (vmcnt counts in order)
#pragma unroll
        for (int i = 0; i < Shape::A_PASSES; ++i) pa[i] = qa[i];
#pragma unroll
        for (int i = 0; i < Shape::B_PASSES; ++i)
        {
            pb[i] = qb[i];
            pok[i] = qok[i];
        }
    }
    else
        stash(0, 0);
    if (TUNE & 1) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    stamp(); // k-tile 0 in LDS

    const int a_off = half * BM + wm * Shape::WTM + l31;
    const int b_off = half * BN + wn * Shape::WTN + l31;
    int cur = 0;
    for (int kt = 0; kt < k_tiles; ++kt)
    {
        // k-tile kt+1 (in registers since the previous iteration) -> the other LDS buffer; k-tile kt+2 -> registers
        if (DEEP)
        {
            // odd tiles live in set 2, even tiles in set 1; the set just emptied takes tile kt+3
            if (kt & 1)
            {
                if (kt + 1 < k_tiles) stash(cur ^ 1, kt + 1);
                if (kt + 3 < k_tiles) fetch(kt + 3);
            }
            else
            {
                if (kt + 1 < k_tiles) stash2(cur ^ 1, kt + 1);
                if (kt + 3 < k_tiles) fetch2(kt + 3);
            }
        }
        else
        {
            if (kt + 1 < k_tiles) stash(cur ^ 1, kt + 1);
            if (kt + 2 < k_tiles && !(ABLATE & 1)) fetch(kt + 2);
        }

        const float* as = As0 + cur * (BK * BM) + a_off;
        const float* bs = Bs0 + cur * (BK * BN) + b_off;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp)
        {
            float fa[Shape::TM], fbv[Shape::TN];
#pragma unroll
            for (int i = 0; i < Shape::TM; ++i) fa[i] = as[(2 * kp) * BM + i * 32];
#pragma unroll
            for (int j = 0; j < Shape::TN; ++j) fbv[j] = bs[(2 * kp) * BN + j * 32];
#pragma unroll
            for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
                for (int j = 0; j < Shape::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fbv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
        if (kt < 4) stamp();
#ifdef FHIP_TIMELINE
        // ABLATE bit 7: wall clock at the end of EVERY k-tile of every block -> g_core_iterlog[block][64]
        if ((ABLATE & 128) && threadIdx.x == 0 && kt < 64) g_core_iterlog[(size_t)blockIdx.x * 64 + kt] = wall_clock64();
#endif
    }
    tl_n = 7;
    stamp(); // k-loop done

    // ---- epilogue.  After the last barrier nobody reads the operand buffers any more: each wave transposes its
    // 32x32 MFMA tiles through a private piece of them (C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2)
    // + 4 * (lane >> 5)) and stores 4 consecutive columns per lane.  Wave-private + in-order LDS queue: no barrier.
    if (ABLATE & 2)
    {
#pragma unroll
        for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
            for (int j = 0; j < Shape::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[i][j][r]));
        return;
    }
    float* const scr = lds + wave * (32 * Shape::EPI_LD);
    const int e_row = lane >> 3, e_c4 = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < Shape::TN; ++j)
    {
        const typename Policy::Store st(prm, batch, n0 + wn * Shape::WTN + j * 32 + e_c4);
        if (j == 0) stamp(); // [8] store descriptor built
#pragma unroll
        for (int i = 0; i < Shape::TM; ++i)
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * half) * Shape::EPI_LD + l31] = acc[i][j][r];
            if (j == 0 && i == 0) stamp(); // [9] first transpose written
            const int mbase = m0 + wm * Shape::WTM + i * 32 + e_row;
#pragma unroll
            for (int q = 0; q < 4; ++q)
            {
                const float4 v = *reinterpret_cast<const float4*>(&scr[(q * 8 + e_row) * Shape::EPI_LD + e_c4]);
                if (TUNE & 2)
                    st.put4b(prm, mbase + q * 8, v, bias_r[i][q], st.residual4(prm, mbase + q * 8));
                else
                    st.put4(prm, mbase + q * 8, v);
                if (j == 0 && i == 0) stamp(); // [10..13] after each store of the first 32x32 piece
            }
        }
    }
    tl_n = 15;
    stamp();
}

} // namespace fhip
