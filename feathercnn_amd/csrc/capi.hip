// capi.hip -- the extern "C" surface declared in include/feather_hip/feather_hip.h: algorithm selection,
// buffer sizing, Init and Forward dispatch (the GPU counterpart of reference src/booster/avx/booster.cpp),
// error reporting and per-stage event timing.
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace fhip
{
// kernels (winograd_f63.hip, implicit_gemm.hip, depthwise.hip)
int winograd_plan(const fhip_conv_param& p, int batch, fhip_winograd_plan* plan);
int winograd_transform_kernel(const fhip_conv_param& p, float* u, const float* kernel, hipStream_t s);
int winograd_input_transform(const fhip_conv_param& p, int batch, float* v, const float* input, hipStream_t s);
int winograd_tile_gemm(const fhip_conv_param& p, int batch, float* m, const float* u, const float* v, hipStream_t s);
int winograd_output_transform(const fhip_conv_param& p, int batch, float* output, const float* m, const float* bias,
                              hipStream_t s);
int winograd_fused_gemm_output(const fhip_conv_param& p, int batch, float* output, const float* u, const float* v, const float* bias,
                               hipStream_t s);
void igemm_packed_dims(const fhip_conv_param& p, int* kd_padded, int* k_padded);
int igemm_init(const fhip_conv_param& p, float* packed, const float* kernel, hipStream_t s);
int igemm_forward(const fhip_conv_param& p, int batch, float* out, const float* in, const float* packed, const float* bias,
                  float* buffer, bool force_no_act, hipStream_t s);
size_t igemm_buffer_bytes(const fhip_conv_param& p, int batch);
int depthwise_forward(const fhip_conv_param& p, int batch, float* out, const float* in, const float* kernel, const float* bias,
                      hipStream_t s);

// Winograd cache blocking: bytes of V + M kept in flight per sub-batch (MiB); measured sweep in DESIGN.md 3.2
constexpr long kWinoChunkMB = 0;

// ---- errors -----------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

int fail(int code, const char* msg)
{
    g_last_error = msg ? msg : "";
    return code;
}

int fail_hip(hipError_t e, const char* what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return FHIP_E_HIP;
}

int device_compute_units()
{
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0)
    {
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
        cached[dev] = cu;
    }
    return cached[dev];
}

// ---- stage timing -----------------------------------------------------------------------------------
struct TimedLaunch
{
    int stage;
    hipEvent_t a, b;
};
static std::mutex g_tm_mu;
static bool g_tm_on = false;
static std::vector<TimedLaunch> g_tm_pending;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_tm_pool;

StageTimer::StageTimer(int st, hipStream_t s) : stage(st), stream(s), slot(-1)
{
    if (!g_tm_on) return;
    std::lock_guard<std::mutex> lk(g_tm_mu);
    TimedLaunch t;
    t.stage = st;
    if (!g_tm_pool.empty())
    {
        t.a = g_tm_pool.back().first;
        t.b = g_tm_pool.back().second;
        g_tm_pool.pop_back();
    }
    else if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess)
        return;
    (void)hipEventRecord(t.a, s);
    g_tm_pending.push_back(t);
    slot = (int)g_tm_pending.size() - 1;
}

StageTimer::~StageTimer()
{
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_tm_mu);
    if (slot < (int)g_tm_pending.size()) (void)hipEventRecord(g_tm_pending[slot].b, stream);
}

static bool valid_param(const fhip_conv_param* p)
{
    return p && p->input_channels > 0 && p->output_channels > 0 && p->input_h > 0 && p->input_w > 0 && p->kernel_h > 0 &&
           p->kernel_w > 0 && p->output_h > 0 && p->output_w > 0;
}

} // namespace fhip

using namespace fhip;

extern "C"
{

int fhip_conv_assign_output_dim(fhip_conv_param* p)
{
    if (!p) return fail(FHIP_E_BADARG, "null param");
    if (p->group == 0) p->group = 1;
    if (p->stride_h == 0) p->stride_h = 1;
    if (p->stride_w == 0) p->stride_w = 1;
    p->output_h = (p->input_h + p->pad_top + p->pad_bottom - p->kernel_h) / p->stride_h + 1;
    p->output_w = (p->input_w + p->pad_left + p->pad_right - p->kernel_w) / p->stride_w + 1;
    if (p->group == p->input_channels) p->output_channels = p->input_channels;
    return FHIP_OK;
}

double fhip_conv_flops(const fhip_conv_param* p)
{
    if (!p || p->group == 0) return 0.0;
    return 2.0 * p->output_channels * p->input_channels * p->output_h * p->output_w * p->kernel_h * p->kernel_w / p->group;
}

int fhip_conv_select_algo(const fhip_conv_param* p, int* algo)
{
    if (!p || !algo) return fail(FHIP_E_BADARG, "null argument");
    if (p->group == p->input_channels)
        *algo = FHIP_DEPTHWISE;
    else if (p->group == 1 && p->kernel_h == 3 && p->kernel_w == 3 && p->stride_h == 1 && p->stride_w == 1 && p->input_h > 8 &&
             p->input_w > 8 && p->output_channels % 4 == 0 && p->input_channels % 4 == 0)
        *algo = FHIP_WINOGRADF63;
    else if (p->group == 1)
        *algo = FHIP_IM2COL;
    else
    {
        *algo = -1;
        return fail(FHIP_E_UNSUPPORTED, "Partial group conv is not supported (same as the reference, avx/booster.cpp:304-308)");
    }
    return FHIP_OK;
}

int fhip_winograd_f63_plan(const fhip_conv_param* p, int batch, fhip_winograd_plan* plan)
{
    if (!valid_param(p) || !plan) return fail(FHIP_E_BADARG, "bad param");
    return winograd_plan(*p, batch, plan);
}

int fhip_conv_get_buffer_size(const fhip_conv_param* p, int algo, int batch, size_t* buffer_bytes, size_t* packed_bytes)
{
    if (!valid_param(p) || !buffer_bytes || !packed_bytes || batch < 1) return fail(FHIP_E_BADARG, "bad argument");
    switch (algo)
    {
        case FHIP_NAIVE:
        case FHIP_IM2COL:
        {
            if (p->group > 1) return fail(FHIP_E_UNSUPPORTED, "implicit GEMM handles group == 1 only");
            int kdp, kp;
            igemm_packed_dims(*p, &kdp, &kp);
            *buffer_bytes = igemm_buffer_bytes(*p, batch); // no column matrix; split-K partial sums for under-filled grids only
            *packed_bytes = (size_t)kdp * kp * sizeof(float);
            return FHIP_OK;
        }
        case FHIP_DEPTHWISE:
            *buffer_bytes = 0; // no padded copy
            *packed_bytes = (size_t)p->group * p->kernel_h * p->kernel_w * sizeof(float);
            return FHIP_OK;
        case FHIP_WINOGRADF63:
        {
            fhip_winograd_plan pl;
            int rc = winograd_plan(*p, batch, &pl);
            if (rc) return rc;
            *buffer_bytes = pl.v_bytes + pl.m_bytes;
            *packed_bytes = pl.u_bytes;
            return FHIP_OK;
        }
        default: return fail(FHIP_E_UNSUPPORTED, "This algo is not supported on gfx950 (nor on AVX2, avx/booster.cpp:348-354)");
    }
}

int fhip_conv_init(const fhip_conv_param* p, int algo, float* packed, const float* kernel, void* stream)
{
    if (!valid_param(p) || !packed || !kernel) return fail(FHIP_E_BADARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    switch (algo)
    {
        case FHIP_NAIVE:
        case FHIP_IM2COL: return igemm_init(*p, packed, kernel, s);
        case FHIP_DEPTHWISE:
        {
            StageTimer tm(FHIP_STAGE_INIT, s);
            FHIP_CHECK_HIP(hipMemcpyAsync(packed, kernel, (size_t)p->group * p->kernel_h * p->kernel_w * sizeof(float),
                                          hipMemcpyDeviceToDevice, s));
            return FHIP_OK;
        }
        case FHIP_WINOGRADF63: return winograd_transform_kernel(*p, packed, kernel, s);
        default: return fail(FHIP_E_UNSUPPORTED, "This algo is not supported on gfx950");
    }
}

int fhip_conv_forward(const fhip_conv_param* p, int algo, int batch, float* output, const float* input, const float* packed,
                      float* buffer, const float* bias, void* stream)
{
    if (!valid_param(p) || !output || !input || !packed || batch < 1) return fail(FHIP_E_BADARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    switch (algo)
    {
        case FHIP_NAIVE: return igemm_forward(*p, batch, output, input, packed, bias, buffer, true, s);
        case FHIP_IM2COL: return igemm_forward(*p, batch, output, input, packed, bias, buffer, false, s);
        case FHIP_DEPTHWISE: return depthwise_forward(*p, batch, output, input, packed, bias, s);
        case FHIP_WINOGRADF63:
        {
            if (!buffer) return fail(FHIP_E_BADARG, "Winograd needs the scratch buffer");
            // measurement switches: FHIP_WINO_FUSED=1 forces the fused GEMM + output-transform kernel (measured slower,
            // DESIGN.md 3.1); FHIP_WINO_CHUNK_MB overrides the sub-batch size of the cache blocking below (0 = off).
            static const int fused_env = [] {
                const char* e = getenv("FHIP_WINO_FUSED");
                return e ? atoi(e) : 0;
            }();
            static const long chunk_mb = [] {
                const char* e = getenv("FHIP_WINO_CHUNK_MB");
                return e ? atol(e) : (long)kWinoChunkMB;
            }();
            // Cache blocking over the batch: V and M of `nb` images are produced and consumed back to back in the same
            // scratch bytes, so for small enough nb they live in the 256 MiB Infinity Cache instead of making four trips
            // through HBM (V write, V read, M write, M read = 3 GB per VGG conv1_2 batch of 32).
            fhip_winograd_plan one;
            int rc = winograd_plan(*p, 1, &one);
            if (rc) return rc;
            const size_t per_image = (size_t)64 * (p->input_channels + p->output_channels) * one.tiles_per_image * sizeof(float);
            int nb = batch;
            if (chunk_mb > 0)
            {
                const long fit = (long)((size_t)chunk_mb * 1024 * 1024 / (per_image ? per_image : 1));
                nb = (int)std::max(1L, std::min((long)batch, fit));
            }
            const size_t in_img = (size_t)p->input_channels * p->input_h * p->input_w;
            const size_t out_img = (size_t)p->output_channels * p->output_h * p->output_w;
            for (int b0 = 0; b0 < batch; b0 += nb)
            {
                const int bn = std::min(nb, batch - b0);
                fhip_winograd_plan pl;
                if ((rc = winograd_plan(*p, bn, &pl))) return rc;
                float* v = reinterpret_cast<float*>(reinterpret_cast<char*>(buffer) + pl.v_offset_bytes);
                float* m = reinterpret_cast<float*>(reinterpret_cast<char*>(buffer) + pl.m_offset_bytes);
                const float* in_b = input + (size_t)b0 * in_img;
                float* out_b = output + (size_t)b0 * out_img;
                if ((rc = winograd_input_transform(*p, bn, v, in_b, s))) return rc;
                if (fused_env)
                {
                    if ((rc = winograd_fused_gemm_output(*p, bn, out_b, packed, v, bias, s))) return rc;
                    continue;
                }
                if ((rc = winograd_tile_gemm(*p, bn, m, packed, v, s))) return rc;
                if ((rc = winograd_output_transform(*p, bn, out_b, m, bias, s))) return rc;
            }
            return FHIP_OK;
        }
        default: return fail(FHIP_E_UNSUPPORTED, "This algo is not supported on gfx950");
    }
}

int fhip_winograd_f63_transform_kernel(const fhip_conv_param* p, float* u, const float* kernel, void* stream)
{
    if (!valid_param(p) || !u || !kernel) return fail(FHIP_E_BADARG, "bad argument");
    return winograd_transform_kernel(*p, u, kernel, (hipStream_t)stream);
}

int fhip_winograd_f63_input_transform(const fhip_conv_param* p, int batch, float* v, const float* input, void* stream)
{
    if (!valid_param(p) || !v || !input) return fail(FHIP_E_BADARG, "bad argument");
    return winograd_input_transform(*p, batch, v, input, (hipStream_t)stream);
}

int fhip_winograd_f63_tile_gemm(const fhip_conv_param* p, int batch, float* m, const float* u, const float* v, void* stream)
{
    if (!valid_param(p) || !m || !u || !v) return fail(FHIP_E_BADARG, "bad argument");
    return winograd_tile_gemm(*p, batch, m, u, v, (hipStream_t)stream);
}

int fhip_winograd_f63_output_transform(const fhip_conv_param* p, int batch, float* output, const float* m, const float* bias,
                                       void* stream)
{
    if (!valid_param(p) || !output || !m) return fail(FHIP_E_BADARG, "bad argument");
    return winograd_output_transform(*p, batch, output, m, bias, (hipStream_t)stream);
}

int fhip_stage_timing_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_tm_mu);
    g_tm_on = on != 0;
    return FHIP_OK;
}

int fhip_stage_timing_collect(double* ms, long long* launches)
{
    if (!ms || !launches) return fail(FHIP_E_BADARG, "null argument");
    std::lock_guard<std::mutex> lk(g_tm_mu);
    for (int i = 0; i < FHIP_STAGE_COUNT; ++i)
    {
        ms[i] = 0.0;
        launches[i] = 0;
    }
    int rc = FHIP_OK;
    for (auto& t : g_tm_pending)
    {
        float dt = 0.f;
        hipError_t e = hipEventSynchronize(t.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&dt, t.a, t.b);
        if (e != hipSuccess)
            rc = fail_hip(e, "stage timing");
        else if (t.stage >= 0 && t.stage < FHIP_STAGE_COUNT)
        {
            ms[t.stage] += dt;
            launches[t.stage] += 1;
        }
        g_tm_pool.emplace_back(t.a, t.b);
    }
    g_tm_pending.clear();
    return rc;
}

const char* fhip_last_error(void) { return g_last_error.c_str(); }

const char* fhip_version(void) { return "feather_hip 0.1 (gfx950)"; }

int fhip_device_info(char* name, int name_len, int* cus, int* lds_bytes)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(FHIP_E_NODEVICE, "no HIP device");
    int dev = 0;
    FHIP_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    FHIP_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (name && name_len > 0)
    {
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
    return FHIP_OK;
}

} // extern "C"
