// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// A small extern "C" shim (ours) over the *real* reference operator API
// booster::ConvParam / booster::ConvBooster (reference: src/booster/include/booster/booster.h:59-170),
// compiled together with the reference's own AVX sources where they lie under /root/reference
// (see oracle/Makefile; no reference source is copied into this repo).  The reference builds with
// -fvisibility=hidden (reference CMakeLists.txt:64), so this TU is the only thing that exports symbols.
//
// Driving sequence is exactly the one feather::ConvLayer uses (reference src/layers/conv_layer.h:92-172):
//   AssignOutputDim -> SelectAlgo (or ForceSelectAlgo) -> GetBufferSize -> Init -> Forward.
// The reference has no batch dimension (conv_layer.h:107): callers loop over images.
// The reference AVX Winograd path is only safe with num_threads == 1.

#include <booster/booster.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <ctime>

namespace
{
float* aligned_floats(size_t n)
{
    void* p = nullptr;
    // reference kernels use aligned SSE/AVX stores on scratch (winograd_kernels_F63.cpp:467); 128 B like mempool.cpp:41
    if (posix_memalign(&p, 128, (n ? n : 1) * sizeof(float) + 256) != 0) return nullptr;
    memset(p, 0, (n ? n : 1) * sizeof(float) + 256);
    return static_cast<float*>(p);
}

void fill_param(booster::ConvParam& p, const int* g)
{
    // g = {input_channels, output_channels, input_h, input_w, kernel_h, kernel_w, stride_h, stride_w,
    //      pad_left, pad_right, pad_top, pad_bottom, group, bias_term, activation}
    memset(&p, 0, sizeof(p));
    p.input_channels = g[0];
    p.output_channels = g[1];
    p.input_h = g[2];
    p.input_w = g[3];
    p.kernel_h = g[4];
    p.kernel_w = g[5];
    p.stride_h = g[6];
    p.stride_w = g[7];
    p.pad_left = g[8];
    p.pad_right = g[9];
    p.pad_top = g[10];
    p.pad_bottom = g[11];
    p.group = g[12];
    p.bias_term = g[13] != 0;
    p.activation = g[14] ? booster::ReLU : booster::None;
    p.AssignOutputDim();
}

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

struct Prepared
{
    booster::ConvParam p;
    booster::ConvBooster b;
    float* scratch = nullptr;
    float* packed = nullptr;
    float* bias = nullptr;
    int rc = 0;
};

// force_algo < 0 -> reference's own SelectAlgo (avx/booster.cpp:283-310)
int prepare(Prepared& s, const int* geom, int force_algo, const float* weights, const float* bias)
{
    fill_param(s.p, geom);
    int rc = (force_algo < 0) ? s.b.SelectAlgo(&s.p) : s.b.ForceSelectAlgo((booster::ConvAlgo)force_algo);
    if (rc != 0 || !s.b.GetBufferSize || !s.b.Init || !s.b.Forward) return -1;
    int buf = 0, pk = 0;
    s.b.GetBufferSize(&s.p, &buf, &pk);
    s.scratch = aligned_floats((size_t)buf);
    s.packed = aligned_floats((size_t)pk);
    // always hand the reference a valid bias pointer (winogradOutputTransform reads it even without bias,
    // avx/winograd_kernels_F63.cpp:1109)
    s.bias = aligned_floats((size_t)s.p.output_channels);
    if (!s.scratch || !s.packed || !s.bias) return -2;
    if (bias) memcpy(s.bias, bias, sizeof(float) * s.p.output_channels);
    s.b.Init(&s.p, s.packed, const_cast<float*>(weights));
    return 0;
}

void release(Prepared& s)
{
    free(s.scratch);
    free(s.packed);
    free(s.bias);
}
} // namespace

extern "C"
{

// Output dims via the reference's own ConvParam::AssignOutputDim (booster.h:113-125).
int ref_conv_output_dims(const int* geom, int* out_c, int* out_h, int* out_w)
{
    booster::ConvParam p;
    fill_param(p, geom);
    *out_c = p.output_channels;
    *out_h = p.output_h;
    *out_w = p.output_w;
    return 0;
}

// Which algorithm the reference's SelectAlgo picks. ConvBooster::algo is private, so identify it
// through the bound function pointers' behaviour: we compare GetBufferSize results against the
// closed forms of avx/booster.cpp (cheap, pure).  Returns the booster::ConvAlgo enum value or -1.
int ref_conv_select_algo(const int* geom)
{
    booster::ConvParam p;
    fill_param(p, geom);
    booster::ConvBooster b;
    if (b.SelectAlgo(&p) != 0) return -1;
    booster::ConvBooster probe;
    const booster::ConvAlgo cands[] = {booster::DEPTHWISE, booster::WINOGRADF63, booster::IM2COL, booster::NAIVE};
    for (booster::ConvAlgo a : cands)
    {
        probe.ForceSelectAlgo(a);
        if (probe.Forward == b.Forward) return (int)a;
    }
    return -1;
}

// Buffer sizes in FLOAT COUNTS exactly as the reference reports them (booster.h:151).
int ref_conv_buffer_size(const int* geom, int force_algo, int* buffer_floats, int* packed_floats)
{
    booster::ConvParam p;
    fill_param(p, geom);
    booster::ConvBooster b;
    int rc = (force_algo < 0) ? b.SelectAlgo(&p) : b.ForceSelectAlgo((booster::ConvAlgo)force_algo);
    if (rc != 0 || !b.GetBufferSize) return -1;
    return b.GetBufferSize(&p, buffer_floats, packed_floats);
}

// One forward over `batch` images (looping the N=1 reference). input [batch][C][H][W], output [batch][K][Ho][Wo].
int ref_conv_forward(const int* geom, int force_algo, int batch, const float* input, const float* weights,
                     const float* bias, float* output)
{
    Prepared s;
    int rc = prepare(s, geom, force_algo, weights, bias);
    if (rc != 0)
    {
        release(s);
        return rc;
    }
    const size_t in_sz = (size_t)s.p.input_channels * s.p.input_h * s.p.input_w;
    const size_t out_sz = (size_t)s.p.output_channels * s.p.output_h * s.p.output_w;
    float* in_al = aligned_floats(in_sz);
    float* out_al = aligned_floats(out_sz);
    for (int n = 0; n < batch; ++n)
    {
        memcpy(in_al, input + n * in_sz, in_sz * sizeof(float));
        s.b.Forward(&s.p, out_al, in_al, s.packed, s.scratch, s.bias, 1);
        memcpy(output + n * out_sz, out_al, out_sz * sizeof(float));
    }
    free(in_al);
    free(out_al);
    release(s);
    return 0;
}

// CPU-baseline timing of the reference Forward (Init untimed), single thread, 1 image per call:
// `warmup` untimed + `reps` timed forwards; returns best-of-reps seconds per image in *best_s and the mean in *mean_s.
int ref_conv_time(const int* geom, int force_algo, const float* input, const float* weights, const float* bias,
                  int warmup, int reps, double* best_s, double* mean_s)
{
    Prepared s;
    int rc = prepare(s, geom, force_algo, weights, bias);
    if (rc != 0)
    {
        release(s);
        return rc;
    }
    const size_t in_sz = (size_t)s.p.input_channels * s.p.input_h * s.p.input_w;
    const size_t out_sz = (size_t)s.p.output_channels * s.p.output_h * s.p.output_w;
    float* in_al = aligned_floats(in_sz);
    float* out_al = aligned_floats(out_sz);
    memcpy(in_al, input, in_sz * sizeof(float));
    for (int i = 0; i < warmup; ++i) s.b.Forward(&s.p, out_al, in_al, s.packed, s.scratch, s.bias, 1);
    double best = 1e30, tot = 0.0;
    for (int i = 0; i < reps; ++i)
    {
        double t0 = now_s();
        s.b.Forward(&s.p, out_al, in_al, s.packed, s.scratch, s.bias, 1);
        double dt = now_s() - t0;
        tot += dt;
        if (dt < best) best = dt;
    }
    *best_s = best;
    *mean_s = reps > 0 ? tot / reps : 0.0;
    free(in_al);
    free(out_al);
    release(s);
    return 0;
}

double ref_conv_flops(const int* geom)
{
    booster::ConvParam p;
    fill_param(p, geom);
    return p.GetFLOPS(); // booster.h:145-148
}

} // extern "C"
