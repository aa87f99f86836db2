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
