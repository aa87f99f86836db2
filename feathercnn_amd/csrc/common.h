// common.h -- shared host/device helpers of the feather_hip kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "feather_hip/feather_hip.h"

namespace fhip
{

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kXcds = 8; // MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (speed only, never correctness)

// Record `msg` as the thread's last error and return `code`.
int fail(int code, const char* msg);
int fail_hip(hipError_t e, const char* what);

#define FHIP_CHECK_HIP(expr)                                       \
    do                                                             \
    {                                                              \
        hipError_t e__ = (expr);                                   \
        if (e__ != hipSuccess) return ::fhip::fail_hip(e__, #expr); \
    } while (0)

// Stage timing (feather_hip.h fhip_stage_timing_*): RAII pair of events around one kernel launch.
struct StageTimer
{
    StageTimer(int stage, hipStream_t s);
    ~StageTimer();
    int stage;
    hipStream_t stream;
    int slot;
};

// Compute units of the current device (cached per device; 256 on MI355X).
int device_compute_units();
// LDS bytes per CU of the current device (cached per device; 160 KiB on MI355X): the residency estimates of the persistent grids use it
size_t device_lds_bytes();

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
inline size_t round_up_sz(size_t a, size_t b) { return (a + b - 1) / b * b; }

// XCD-aware bijective remap of a 1-D grid: consecutive virtual ids land on the same XCD, so blocks that
// share operand panels share that XCD's private 4 MiB L2 (cdna_hip_programming.md T1, bijective form).
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg / kXcds, r = nwg % kXcds;
    const int xcd = bid % kXcds, local = bid / kXcds;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + local;
}

__device__ __forceinline__ float apply_act(float v, bool relu) { return relu ? fmaxf(v, 0.f) : v; }

// 16-byte global accesses with the non-temporal hint (`nt`), for streams that pass through the cache hierarchy exactly once.
__device__ __forceinline__ float4 ldg4_nt(const float* p)
{
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stg4_nt(float* p, float4 v)
{
    f32x4 w;
    w.x = v.x;
    w.y = v.y;
    w.z = v.z;
    w.w = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(p));
}
// 16-byte stores of ACTIVATIONS (layer outputs).  FHIP_ACT_NT is a mask of the kernel families whose output stores carry `nt`:
// 1 = implicit-GEMM / streamed 1x1 convolutions, 2 = the staged Winograd output transform, 4 = depthwise kernels, 8 = the band-staged
// depthwise + pointwise kernel (measured per build with tools/variant_ab.sh).
#ifndef FHIP_ACT_NT
#define FHIP_ACT_NT 0
#endif
template <int FAMILY>
__device__ __forceinline__ void stg4_act(float* p, float4 v)
{
    if constexpr ((FHIP_ACT_NT & FAMILY) != 0)
        stg4_nt(p, v);
    else
        *reinterpret_cast<float4*>(p) = v;
}

// Dword accesses of the Winograd transforms to the scratch tensors V / M.  Every element of V and M is written once by one launch and
// read once by the next one: FHIP_XFORM_NT bit 0 puts the non-temporal hint on those loads, bit 1 on those stores (measured per build with
// tools/variant_ab.sh; 0 = plain accesses).
#ifndef FHIP_XFORM_NT
#define FHIP_XFORM_NT 0
#endif
__device__ __forceinline__ float ld_scratch(const float* p)
{
    if constexpr ((FHIP_XFORM_NT & 1) != 0) return __builtin_nontemporal_load(p);
    return *p;
}
__device__ __forceinline__ void st_scratch(float* p, float v)
{
    if constexpr ((FHIP_XFORM_NT & 2) != 0)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// Conv geometry as the kernels want it (decoded once on the host from fhip_conv_param).
struct ConvGeom
{
    int C, K, H, W, KH, KW, OH, OW, SH, SW, PL, PT;
    int N;      // batch
    int has_bias;
    int relu;
};

inline ConvGeom make_geom(const fhip_conv_param& p, int batch)
{
    ConvGeom g;
    g.C = p.input_channels;
    g.K = p.output_channels;
    g.H = p.input_h;
    g.W = p.input_w;
    g.KH = p.kernel_h;
    g.KW = p.kernel_w;
    g.OH = p.output_h;
    g.OW = p.output_w;
    g.SH = p.stride_h > 0 ? p.stride_h : 1;
    g.SW = p.stride_w > 0 ? p.stride_w : 1;
    g.PL = p.pad_left;
    g.PT = p.pad_top;
    g.N = batch;
    g.has_bias = p.bias_term != 0;
    g.relu = p.activation == FHIP_ACT_RELU;
    return g;
}

} // namespace fhip
