// flat_gemm.h -- the fp32-MFMA GEMM main loop of round 2: LDS-DMA operand staging through a RING of D k-tile stages whose
// loads run ahead ACROSS output-tile boundaries.  Serves the same two GEMMs as gemm_core.h:
//   * Winograd tile GEMM  M_xi[K x P] = U_xi[K x C] * V_xi[C x P]   (reference TensorGEMM,
//     src/booster/avx/winograd_kernels_F63.cpp:518-692), and
//   * the 1x1 implicit-GEMM convolution out[K x N*Ho*Wo] = W[K x C] * in  (reference IM2COL_Forward +
//     packed_sgemm_activation, avx/booster.cpp:83-102, avx/sgemm.cpp:377-433; for a 1x1 kernel booster::im2col is a copy --
//     or a strided sub-sampling -- of the input, so the column matrix is addressed in place).
//
// Why (round-1 profile): the nets' GEMMs are SHORT in the reduction -- C = 64..512 is 4..32 k-tiles -- so a block that owns one
// output tile spends a large part of its life in the prologue (one HBM round trip before the first MFMA) and in the epilogue,
// and ResNet-50's 1x1 layers (6.6 GFLOP = 42 us at peak each) ran at 46 % MFMA-busy.  Here a block walks `tpb` consecutive
// n-tiles of one (batch entry, m-tile); the (tile, k-tile) pairs form ONE flat sequence of stages, stage f's operands are
// requested D-1 stages before they are used whatever tile they belong to, so the loads of tile t+1 are in flight while tile t
// finishes its MFMAs and stores, and only the first tile of a block pays a prologue.
//
// Mechanics (cdna_hip_programming.md "Pipelining across barriers", MI355X_MICROARCH.md item 7):
//   * global_load_lds writes 64 lanes x 16 B (or x 4 B for gathered columns) linearly into LDS: the stage image is the k-major
//     [BK][BM] | [BK][BN] layout the one-VGPR MFMA operand reads want (conflict-free ds_read_b32, see gemm_core.h);
//   * vmcnt retires in order and counts stores too: the wait in front of stage s allows exactly the operations issued after
//     stage s's loads -- the loads of the stages behind it plus the accumulator stores of tile epilogues in that window;
//   * raw s_barrier (a __syncthreads() would drain the DMA queue), one per stage: RAW (every wave's pieces of stage s landed)
//     and WAR (the stage refilled next was read by all waves in the previous iteration) in one;
//   * no VGPR-destination global load inside the loop (hipcc answers one with vmcnt(0), which would drain the ring): the lane's
//     bias values are requested right behind the prologue's operand requests and waited for once, ahead of the loop;
//   * accumulators leave straight from the MFMA registers as 16-byte stores (operand roles swapped, see the kernel): no LDS
//     scratch, so the ring is all the LDS a block needs.
#pragma once

#include "common.h"

namespace fhip
{

#ifndef FHIP_LDS_VOID_DEFINED
#define FHIP_LDS_VOID_DEFINED
typedef __attribute__((address_space(3))) void lds_void;
#endif

template <int BM_, int BN_, int WAVES_M_, int WAVES_N_, int D_, int OCC_>
struct FlatShape
{
    static constexpr int BM = BM_, BN = BN_, BK = 16, D = D_, OCC = OCC_;
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
    static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    static constexpr int TM = WTM / 32, TN = WTN / 32;
    static constexpr int STAGE = BK * (BM + BN); // floats
    static constexpr int LDS_FLOATS = D * STAGE; // the ring (wave_gemm.h adds its own scratch)
    static constexpr int S = TM * TN * 4;        // 16-byte stores per lane and output tile
    static_assert(WAVES_M * WAVES_N == 4, "four waves");
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && BM % 64 == 0 && BN % 16 == 0, "tile shape");
    static_assert(D >= 2 && D <= 5, "ring depth");
};

__device__ __forceinline__ void flat_wait(int n)
{
    // n is wave-uniform; the immediate must be a literal
    switch (__builtin_amdgcn_readfirstlane(n))
    {
#define FHIP_W(i) \
    case i: asm volatile("s_waitcnt vmcnt(" #i ") lgkmcnt(0)" ::: "memory"); break;
        FHIP_W(0) FHIP_W(1) FHIP_W(2) FHIP_W(3) FHIP_W(4) FHIP_W(5) FHIP_W(6) FHIP_W(7) FHIP_W(8) FHIP_W(9) FHIP_W(10) FHIP_W(11)
        FHIP_W(12) FHIP_W(13) FHIP_W(14) FHIP_W(15) FHIP_W(16) FHIP_W(17) FHIP_W(18) FHIP_W(19) FHIP_W(20) FHIP_W(21) FHIP_W(22)
        FHIP_W(23) FHIP_W(24) FHIP_W(25) FHIP_W(26) FHIP_W(27) FHIP_W(28) FHIP_W(29) FHIP_W(30) FHIP_W(31) FHIP_W(32) FHIP_W(33)
        FHIP_W(34) FHIP_W(35) FHIP_W(36) FHIP_W(37) FHIP_W(38) FHIP_W(39) FHIP_W(40) FHIP_W(41) FHIP_W(42) FHIP_W(43) FHIP_W(44)
        FHIP_W(45) FHIP_W(46) FHIP_W(47) FHIP_W(48)
#undef FHIP_W
        default: asm volatile("s_waitcnt vmcnt(48) lgkmcnt(0)" ::: "memory"); break; // stricter than asked: always safe
    }
}

// measurement builds only (ABLATE bit 3): every 64th block adds its shader-clock and 100 MHz wall-clock ticks here, which gives the
// effective shader clock the kernel really ran at (the chip clocks to its power budget)
static __device__ unsigned long long g_flat_clock_probe[2];

// measurement builds only (ABLATE bit 5): shader-clock stamps of sampled blocks: [sample][wave][16] = start, ring primed, after the
// wait / barrier / MFMAs of the first stages, end
static __device__ long long g_flat_timeline[64][4][16];

// one wave-wide LDS-DMA request: 64 lanes x VEC floats from per-lane global addresses to `dst` + lane * VEC (dst wave-uniform)
template <int VEC>
__device__ __forceinline__ void flat_request(const float* src, float* dst)
{
    if constexpr (VEC == 4)
        __builtin_amdgcn_global_load_lds(src, (lds_void*)dst, 16, 0, 0);
    else
        __builtin_amdgcn_global_load_lds(src, (lds_void*)dst, 4, 0, 0);
}

// Policy concept (static members; VEC = floats per lane of a B-operand request: 4 = dwordx4, 1 = dword gather):
//   struct Params { int batches, m_tiles, n_tiles, k_tiles, tpb; ... };
//   const float* a_ptr(p, batch, m, krow)   address of A[krow][m]; consecutive krow are lda(p) floats apart, 4 consecutive m contiguous
//   const float* b_ptr(p, batch, n)         address of B[0][n] (n clamped into the matrix); consecutive rows are ldb(p) floats apart
//   int krows(p)                            rows of B that exist (rows beyond re-read the last one and meet zero rows of A)
//   const float* bias(p), int rows(p), int cols(p)   bias vector or nullptr, number of real output rows / columns
//   struct Out { Out(p, batch, n4); void put4(p, m, v, bias_m) }   4 consecutive output columns of row m
// ABLATE (measurement builds only, tools/flat_bench.hip; the product always uses 0): bit 0 = every tile re-requests the block's FIRST
// B tile (no new HBM reads), bit 1 = no accumulator stores, bit 2 = no MFMAs, bit 3 = clock probe, bit 4 = no operand requests
// after the prologue (the loop reads stale LDS: pure LDS + MFMA + synchronisation), bit 5 = timeline stamps.
//
// Epilogue (round-2 timeline probe: the LDS-transposed epilogue of gemm_core.h costs ~3400 cycles of serialised LDS round trips
// per tile, and a bias fetched ahead of the first operand request adds a whole memory round trip to every block): the MFMA is
// issued with the operand ROLES SWAPPED -- the B-tile fragment as srcA, the A-tile fragment as srcB -- so the accumulator is the
// transposed 32 x 32 piece: lane l holds row m = l & 31 and, per register quad g, the four CONSECUTIVE columns
// n = 8g + 4 (l >> 5) .. +3.  A quad is one 16-byte store straight into the output row: no LDS scratch, no transpose, and the
// lane's bias is one value per 32-row piece, loaded after the ring is primed and first used after the k-loop.
template <class Shape, class Policy, int ABLATE = 0>
__global__ __launch_bounds__(256, Shape::OCC) void flat_gemm_kernel(const typename Policy::Params prm)
{
    constexpr int BM = Shape::BM, BN = Shape::BN, BK = Shape::BK, D = Shape::D, STAGE = Shape::STAGE;
    constexpr int VEC = Policy::VEC;
    constexpr int GA = BK * BM / 256 / 4;                         // A requests per wave and stage (16-byte lanes)
    constexpr int B_PIECES = BK * BN / (64 * VEC);                // wave-wide B requests per stage
    constexpr int GB = (B_PIECES + 3) / 4;
    __shared__ __attribute__((aligned(16))) float lds[D * STAGE]; // ONE LDS object (a second one de-pipelines hipcc's waits)

    const long long probe_c0 = (ABLATE & 8) ? clock64() : 0, probe_w0 = (ABLATE & 8) ? wall_clock64() : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool tl_on = (ABLATE & 32) && (blockIdx.x % 97) == 5 && blockIdx.x / 97 < 64;
    long long* const tl = tl_on ? &g_flat_timeline[blockIdx.x / 97][wave][0] : nullptr;
    int tl_n = 0;
    auto stamp = [&]() {
        if ((ABLATE & 32) && tl_on && lane == 0 && tl_n < 16) tl[tl_n] = clock64();
        ++tl_n;
    };
    stamp();

    const int n_groups = (prm.n_tiles + prm.tpb - 1) / prm.tpb;
    const int nwg = prm.batches * prm.m_tiles * n_groups;
    int vid = xcd_remap(blockIdx.x, nwg);
    const int mt = vid % prm.m_tiles;
    vid /= prm.m_tiles;
    const int grp = vid % n_groups;
    const int batch = vid / n_groups;
    const int m0 = mt * BM;
    const int t0 = grp * prm.tpb;
    const int ntl = min(prm.tpb, prm.n_tiles - t0);
    const int k_tiles = prm.k_tiles;
    const int total = ntl * k_tiles;

    const int wm = wave / Shape::WAVES_N, wn = wave % Shape::WAVES_N;
    const int l31 = lane & 31, half = lane >> 5;

    // ---- what this lane requests in every stage (invariant: position inside the stage image)
    const float* a_src[GA];
#pragma unroll
    for (int g = 0; g < GA; ++g)
    {
        const int o = (wave + 4 * g) * 256 + lane * 4;
        a_src[g] = Policy::a_ptr(prm, batch, m0 + o % BM, o / BM);
    }
    const size_t a_step = (size_t)BK * Policy::lda(prm);
    int b_row[GB], b_col[GB];
    int gw = GA; // requests this wave really issues per stage (wave-uniform)
#pragma unroll
    for (int g = 0; g < GB; ++g)
    {
        const int o = (wave + 4 * g) * (64 * VEC) + lane * VEC;
        b_row[g] = o / BN;
        b_col[g] = o % BN;
        if (wave + 4 * g < B_PIECES) ++gw;
    }
    const size_t ldb = Policy::ldb(prm);
    const int krows = Policy::krows(prm);
    const float* b_src[GB];
    auto set_issue_tile = [&](int t) {
#pragma unroll
        for (int g = 0; g < GB; ++g) b_src[g] = Policy::b_ptr(prm, batch, (t0 + ((ABLATE & 1) ? 0 : t)) * BN + b_col[g]);
    };

    // issue side of the flat sequence
    int it_i = 0, kt_i = 0, buf_i = 0;
    set_issue_tile(0);
    auto issue_next = [&]() {
        float* base = lds + buf_i * STAGE;
#pragma unroll
        for (int g = 0; g < GA; ++g) flat_request<4>(a_src[g] + (size_t)kt_i * a_step, base + (wave + 4 * g) * 256);
#pragma unroll
        for (int g = 0; g < GB; ++g)
            if (wave + 4 * g < B_PIECES)
            {
                const int r = min(kt_i * BK + b_row[g], krows - 1);
                flat_request<VEC>(b_src[g] + (size_t)r * ldb, base + BK * BM + (wave + 4 * g) * (64 * VEC));
            }
        buf_i = buf_i + 1 == D ? 0 : buf_i + 1;
        if (++kt_i == k_tiles)
        {
            kt_i = 0;
            if (++it_i < ntl) set_issue_tile(it_i);
        }
    };
#pragma unroll
    for (int f = 0; f < D - 1; ++f)
        if (f < total) issue_next();
    stamp(); // ring primed (requests issued)

    // the lane's bias values: requested BEHIND the first operand requests, first used after a whole k-loop
    const int out_rows = Policy::rows(prm), out_cols = Policy::cols(prm);
    float bv[Shape::TM];
    {
        const float* bp = Policy::bias(prm);
#pragma unroll
        for (int i = 0; i < Shape::TM; ++i)
        {
            const int m = m0 + wm * Shape::WTM + i * 32 + l31;
            bv[i] = bp ? bp[min(m, out_rows - 1)] : 0.f;
        }
    }

    f32x16 acc[Shape::TM][Shape::TN];
#pragma unroll
    for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
        for (int j = 0; j < Shape::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_off = half * BM + wm * Shape::WTM + l31;
    const int b_off = BK * BM + half * BN + wn * Shape::WTN + l31;
    // wave-uniform copies for the store bookkeeping (SGPRs: the conditions below become scalar branches)
    const int wm_u = __builtin_amdgcn_readfirstlane(wm), wn_u = __builtin_amdgcn_readfirstlane(wn);

    // Consume the bias registers HERE: hipcc places its s_waitcnt vmcnt(0) for them in front of this statement -- once, ahead of the
    // loop, where it only waits for the prologue's stages -- instead of in front of their first use in the epilogue, where it
    // would drain the ring at every tile.
#pragma unroll
    for (int i = 0; i < Shape::TM; ++i) asm volatile("" ::"v"(bv[i]));

    int cur = 0, kt = 0, it = 0;
    int hist[D - 1]; // hist[i]: store instructions this wave issued in iteration s-1-i (they sit behind the loads we wait for)
#pragma unroll
    for (int i = 0; i < D - 1; ++i) hist[i] = 0;
    for (int s = 0; s < total; ++s)
    {
        // Operations issued after stage s's loads: the loads of the `later` stages behind it and the epilogue stores of the last
        // D-1 iterations.  The count must never be ABOVE the truth (that would under-wait); every store counted below is issued
        // under a wave-uniform condition with at least one active lane, so it is a lower bound (exact for 16-byte stores).
        // (The bias loads of the prologue sit behind the prologue's stages too: the first D-1 waits are stricter by TM.  Safe.)
        const int later = min(total - 1 - s, D - 2);
        int behind = later * gw;
#pragma unroll
        for (int i = 0; i < D - 1; ++i) behind += hist[i];
        flat_wait(behind);
        if (s < 4) stamp(); // my pieces landed
        __builtin_amdgcn_s_barrier();
        if (s < 4) stamp(); // everybody's pieces landed
        if (s + D - 1 < total && !(ABLATE & 16)) issue_next();

        const float* as = lds + cur * STAGE + a_off;
        const float* bs = lds + cur * STAGE + b_off;
#pragma unroll
        for (int kp = 0; kp < ((ABLATE & 4) ? 0 : BK / 2); ++kp)
        {
            float fa[Shape::TM], fb[Shape::TN];
#pragma unroll
            for (int i = 0; i < Shape::TM; ++i) fa[i] = as[(2 * kp) * BM + i * 32];
#pragma unroll
            for (int j = 0; j < Shape::TN; ++j) fb[j] = bs[(2 * kp) * BN + j * 32];
#pragma unroll
            for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
                for (int j = 0; j < Shape::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j], fa[i], acc[i][j], 0, 0, 0); // roles swapped: acc = (A B)^T piece
        }
        if (s < 4) stamp(); // MFMAs issued

        int stores = 0;
        const bool end = kt == k_tiles - 1;
        if (end)
        {
            // ---- tile epilogue: acc[i][j][4g + e] = out[m = .. + i*32 + l31][n = .. + j*32 + 8g + 4*half + e]
            const int n0 = (t0 + it) * BN;
#pragma unroll
            for (int j = 0; j < Shape::TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                {
                    const int nq = n0 + wn_u * Shape::WTN + j * 32 + 8 * g; // first column of this instruction's 8 (wave-uniform)
                    if (nq < out_cols)
                    {
                        const typename Policy::Out st(prm, batch, nq + 4 * half);
#pragma unroll
                        for (int i = 0; i < Shape::TM; ++i)
                        {
                            const int mi = m0 + wm_u * Shape::WTM + i * 32; // first row of this instruction's 32 (wave-uniform)
                            if (mi < out_rows)
                            {
                                const float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                                if (ABLATE & 2)
                                    asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                                else
                                {
                                    st.put4(prm, mi + l31, v, bv[i]);
                                    ++stores;
                                }
                            }
                        }
                    }
                }
#pragma unroll
            for (int i = 0; i < Shape::TM; ++i)
#pragma unroll
                for (int j = 0; j < Shape::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            ++it;
            kt = 0;
        }
        else
            ++kt;
#pragma unroll
        for (int i = D - 2; i > 0; --i) hist[i] = hist[i - 1];
        hist[0] = stores;
        cur = cur + 1 == D ? 0 : cur + 1;
    }
    if ((ABLATE & 32) && tl_on && lane == 0) tl[15] = clock64();
    if ((ABLATE & 8) && threadIdx.x == 0 && (blockIdx.x & 63) == 0)
    {
        atomicAdd(&g_flat_clock_probe[0], (unsigned long long)(clock64() - probe_c0));
        atomicAdd(&g_flat_clock_probe[1], (unsigned long long)(wall_clock64() - probe_w0));
    }
}

// ---- Winograd tile GEMM policy (layouts of winograd_f63.hip: U[xi][Cp][Kp], V[xi][C][Pp], M[xi][K][Pp]) ------------------
struct FlatWinoParams
{
    int batches, m_tiles, n_tiles, k_tiles, tpb;
    const float* U;
    const float* V;
    float* M;
    int C, K, Cp, Kp, Pp;
};

struct FlatWinoPolicy
{
    using Params = FlatWinoParams;
    static constexpr int VEC = 4;
    static __device__ const float* a_ptr(const Params& p, int xi, int m, int krow)
    {
        return p.U + ((size_t)xi * p.Cp + krow) * p.Kp + m; // zero padded in both dimensions
    }
    static __device__ size_t lda(const Params& p) { return p.Kp; }
    static __device__ const float* b_ptr(const Params& p, int xi, int n) { return p.V + (size_t)xi * p.C * p.Pp + min(n, p.Pp - 4); }
    static __device__ size_t ldb(const Params& p) { return p.Pp; }
    static __device__ int krows(const Params& p) { return p.C; }
    static __device__ const float* bias(const Params&) { return nullptr; }
    static __device__ int rows(const Params& p) { return p.K; }
    static __device__ int cols(const Params& p) { return p.Pp; }
    struct Out
    {
        float* base;
        __device__ Out(const Params& p, int xi, int n4) : base(n4 < p.Pp ? p.M + (size_t)xi * p.K * p.Pp + n4 : nullptr) {}
        __device__ void put4(const Params& p, int m, float4 v, float) const
        {
            if (base && m < p.K) *reinterpret_cast<float4*>(base + (size_t)m * p.Pp) = v;
        }
    };
};

// ---- 1x1 convolution policy -----------------------------------------------------------------------------------------------
struct FlatConvParams
{
    int batches, m_tiles, n_tiles, k_tiles, tpb;
    const float* Wt; // panel-major packed weights [Kp / bm][Kdp][bm] (igemm_pack_weights_kernel)
    const float* in;
    float* out;
    const float* bias;
    int C, K, Kdp, bm;
    int HWin, W;       // input plane size and row length
    int OW, OHW;       // output row length and plane size
    int SH, SW;
    int Ntot;          // N * OHW
    int relu, has_residual;
    ptrdiff_t residual_delta; // byte offset from `out` to the residual tensor (same layout)
};

// VEC = 4: stride 1 and OHW % 4 == 0 -- 4 consecutive columns are 16 contiguous, aligned bytes of one image's plane.
// VEC = 1: any stride / plane size: one column per lane.
template <int VEC_>
struct FlatConvPolicy
{
    using Params = FlatConvParams;
    static constexpr int VEC = VEC_;
    static __device__ const float* a_ptr(const Params& p, int, int m, int krow)
    {
        return p.Wt + ((size_t)(m / p.bm) * p.Kdp + krow) * p.bm + (m % p.bm);
    }
    static __device__ size_t lda(const Params& p) { return p.bm; }
    static __device__ const float* b_ptr(const Params& p, int, int n)
    {
        n = min(n, p.Ntot - VEC);
        const int img = n / p.OHW, rem = n - img * p.OHW;
        const float* plane0 = p.in + (size_t)img * p.C * p.HWin;
        if (VEC == 4) return plane0 + rem;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        return plane0 + (size_t)(oy * p.SH) * p.W + ox * p.SW;
    }
    static __device__ size_t ldb(const Params& p) { return p.HWin; }
    static __device__ int krows(const Params& p) { return p.C; }
    static __device__ const float* bias(const Params& p) { return p.bias; }
    static __device__ int rows(const Params& p) { return p.K; }
    static __device__ int cols(const Params& p) { return p.Ntot; }

    struct Out
    {
        float* ptr[VEC == 4 ? 1 : 4]; // &out[img][0][rem] of the column(s)
        unsigned valid;
        bool wide;
        __device__ Out(const Params& p, int, int n4)
        {
            if (VEC == 4)
            {
                const int img = n4 / p.OHW, rem = n4 - img * p.OHW;
                valid = n4 < p.Ntot ? 0xfu : 0u;
                wide = valid != 0;
                ptr[0] = p.out + (size_t)img * p.K * p.OHW + rem;
            }
            else
            {
                valid = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                {
                    const int col = n4 + e;
                    const bool ok = col < p.Ntot;
                    const int cc = ok ? col : 0;
                    const int img = cc / p.OHW, rem = cc - img * p.OHW;
                    ptr[e] = p.out + (size_t)img * p.K * p.OHW + rem;
                    valid |= ok ? (1u << e) : 0u;
                }
                wide = (valid == 0xfu) && ((p.OHW & 3) == 0) && (ptr[VEC == 4 ? 0 : 3] == ptr[0] + 3);
            }
        }
        __device__ void put4(const Params& p, int m, float4 v, float b) const
        {
            if (m >= p.K || !valid) return;
            v.x += b;
            v.y += b;
            v.z += b;
            v.w += b;
            const size_t moff = (size_t)m * p.OHW;
            if (p.has_residual)
            {
                if (wide)
                {
                    const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(ptr[0] + moff) + p.residual_delta);
                    v.x += r.x;
                    v.y += r.y;
                    v.z += r.z;
                    v.w += r.w;
                }
                else
                {
                    auto res = [&](const float* o) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(o) + p.residual_delta); };
                    if (valid & 1u) v.x += res(ptr[0] + moff);
                    if (valid & 2u) v.y += res(ptr[VEC == 4 ? 0 : 1] + moff);
                    if (valid & 4u) v.z += res(ptr[VEC == 4 ? 0 : 2] + moff);
                    if (valid & 8u) v.w += res(ptr[VEC == 4 ? 0 : 3] + moff);
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
                *reinterpret_cast<float4*>(ptr[0] + moff) = v;
            else
            {
                if (valid & 1u) ptr[0][moff] = v.x;
                if (valid & 2u) ptr[VEC == 4 ? 0 : 1][moff] = v.y;
                if (valid & 4u) ptr[VEC == 4 ? 0 : 2][moff] = v.z;
                if (valid & 8u) ptr[VEC == 4 ? 0 : 3][moff] = v.w;
            }
        }
    };
};

} // namespace fhip
