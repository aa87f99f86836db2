// wino_gemm_policy.h -- operand loaders / accumulator store of the Winograd tile GEMM
//   M_xi[K x P] = U_xi[K x C] * V_xi[C x P], xi = 0..63   (reference TensorGEMM,
//   src/booster/avx/winograd_kernels_F63.cpp:518-692) for the shared main loop in gemm_core.h.
#pragma once

#include "gemm_core.h"
#include "wino_layout.h"

namespace fhip
{

struct WinoGemmParams
{
    int batches, m_tiles, n_tiles, k_tiles;
    const float* U;
    const float* V;
    float* M;
    int C, K, Cp, Kp, Pp;
    WinoLayout Lv, Lm; // where V (rows = C) and M (rows = K) live; the column tile divides their column block (wino_layout.h)
    // wino_gemm_glds_kernel only (round 5): tiles [tail_first, batches * m_tiles * n_tiles) are cut into tail_parts (2 / 4) ROW pieces of 64 / 32
    // rows, one block each -- the remainder of the tile count over the CU count spread over all CUs instead of a whole tile more on some
    // (wino_gemm_row_split).  tail_parts = 1: every block is a whole tile.
    int tail_first = 0, tail_parts = 1;
};

// NT bit 0: V loads carry `nt` (every V element is read by exactly one block when the row tile covers all of K: m_tiles = 1);
//    bit 1: M stores carry `nt` (M is written once and read once, by the next launch)
template <int NT>
struct WinoGemmPolicyT
{
    typedef WinoGemmParams Params;
    static constexpr int EXTRA_LDS_FLOATS = 0;
    static __device__ void stage_extra(const Params&, float*, int, int) {}
    static __device__ int k_count(const Params& p, int) { return p.k_tiles; }
    static __device__ float bias_at(const Params&, int) { return 0.f; } // the bias is the output transform's business
    struct ALoad
    {
        const float* base;
        __device__ ALoad(const Params& p, int xi, int m4) : base(p.U + (size_t)xi * p.Cp * p.Kp + m4) {}
        __device__ float4 load(const Params& p, int krow) const
        {
            return *reinterpret_cast<const float4*>(base + (size_t)krow * p.Kp); // U is zero padded to [Cp][Kp]
        }
    };
    struct BLoad
    {
        typedef float4 Raw;
        __device__ float4 finish(const Params&, const Raw& r, int, const float*) const { return r; }
        const float* base;
        __device__ BLoad(const Params& p, int xi, int n4) : base(p.V + (size_t)xi * p.Lv.xis + p.Lv.col(n4)) {}
        __device__ float4 load(const Params& p, int krow, unsigned& ok) const
        {
            // unconditional: rows past C re-read row C-1 and are zeroed at LDS-write time
            ok = krow < p.C ? 0xfu : 0u;
            const float* q = base + (size_t)min(krow, p.C - 1) * p.Lv.bp;
            if constexpr (NT & 1) return ldg4_nt(q);
            return *reinterpret_cast<const float4*>(q);
        }
    };
    struct Store
    {
        float* base;
        __device__ Store(const Params& p, int xi, int n4) : base(p.M + (size_t)xi * p.Lm.xis + p.Lm.col(n4)) {}
        __device__ void put4(const Params& p, int m, float4 v) const
        {
            if (m >= p.K) return; // Pp is a multiple of the column tile
            if constexpr (NT & 2)
                stg4_nt(base + (size_t)m * p.Lm.bp, v);
            else
                *reinterpret_cast<float4*>(base + (size_t)m * p.Lm.bp) = v;
        }
        __device__ float4 residual4(const Params&, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
        __device__ void put4b(const Params& p, int m, float4 v, float, float4) const { put4(p, m, v); }
    };
};

// Cache policy of the M stores, by the size of M (winograd_tile_gemm): FHIP_M_NT_BIG for an M of at least FHIP_M_NT_BYTES, else FHIP_M_NT_SMALL
// (0 plain, 2 `nt`).  Round 5, tools/variant_ab.sh: an M of hundreds of MB cannot stay in any cache until the next launch
// reads it -- `nt` stores take 1.5 - 2.5 % off the MFMA-bound launches and 3 - 7 % off the HBM-bound ones (C = 64) -- while a small M (ResNet-50:
// <= 105 MB) written `nt` makes the output transform that reads it 6 - 13 % slower.  Write-through (`sc1`) stores measured no better than `nt`
// (inline asm: needs hand-placed wait states, and without them corrupted M; buffer stores with per-lane bases: 15 - 90 % slower) and are not kept.
#ifndef FHIP_M_NT_BIG
#define FHIP_M_NT_BIG 2
#endif
#ifndef FHIP_M_NT_SMALL
#define FHIP_M_NT_SMALL 0
#endif
// V loads of the register-staged kernel (K <= 64, or C < 128: one row tile covers all of K, so every V element is read by exactly one block)
#ifndef FHIP_V_NT_ONCE
#define FHIP_V_NT_ONCE 1
#endif
// ... and only for a V of at least this many bytes (a smaller V may still be in the memory-side cache, where the transform before just wrote it)
#ifndef FHIP_V_NT_BYTES
#define FHIP_V_NT_BYTES 0u
#endif
#ifndef FHIP_M_NT_BYTES
#define FHIP_M_NT_BYTES (150u << 20)
#endif
typedef WinoGemmPolicyT<0> WinoGemmPolicy; // parameter block + plain accesses; the launches pick WinoGemmPolicyT<FHIP_M_NT_*>

} // namespace fhip
