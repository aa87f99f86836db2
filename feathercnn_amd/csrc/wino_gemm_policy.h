// wino_gemm_policy.h -- operand loaders / accumulator store of the Winograd tile GEMM
//   M_xi[K x P] = U_xi[K x C] * V_xi[C x P], xi = 0..63   (reference TensorGEMM,
//   src/booster/avx/winograd_kernels_F63.cpp:518-692) for the shared main loop in gemm_core.h.
#pragma once

#include "gemm_core.h"
#include "wino_layout.h"

namespace fhip
{

struct WinoGemmPolicy
{
    struct Params
    {
        int batches, m_tiles, n_tiles, k_tiles;
        const float* U;
        const float* V;
        float* M;
        int C, K, Cp, Kp, Pp;
        WinoLayout Lv, Lm; // where V (rows = C) and M (rows = K) live; the column tile divides their column block (wino_layout.h)
    };
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
            return *reinterpret_cast<const float4*>(base + (size_t)min(krow, p.C - 1) * p.Lv.bp);
        }
    };
    struct Store
    {
        float* base;
        __device__ Store(const Params& p, int xi, int n4) : base(p.M + (size_t)xi * p.Lm.xis + p.Lm.col(n4)) {}
        __device__ void put4(const Params& p, int m, float4 v) const
        {
            if (m < p.K) *reinterpret_cast<float4*>(base + (size_t)m * p.Lm.bp) = v; // Pp is a multiple of the column tile
        }
        __device__ float4 residual4(const Params&, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
        __device__ void put4b(const Params& p, int m, float4 v, float, float4) const { put4(p, m, v); }
    };
};


} // namespace fhip
