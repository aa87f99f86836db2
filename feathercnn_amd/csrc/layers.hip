// layers.hip -- SURVEY.md 8(f) rank 1: the layers BETWEEN the convolutions on the device, so a whole forward pass stays
// in HBM.  All of them are HBM-bound elementwise / window kernels; each follows the reference layer it replaces
// (paths relative to /root/reference/src):
//   relu            layers/relu_layer.h:29-41
//   add (+relu)     layers/eltwise_layer.h:69-80 -> booster::add_relu<fuse_relu>, booster/avx/generic_kernels.cpp:138
//   affine (+relu)  layers/batchnorm_layer.h:43-75 (folded alpha/beta), booster::batchnorm<bias,scale,relu>
//                   generic_kernels.cpp:237-279, layers/scale_layer.h + booster::scale<bias> generic_kernels.cpp:203-233
//   pooling         layers/pooling_layer.h:37-88 (max / average / global; NB the window origin subtracts BOTH pads,
//                   :56,:67, the divisor is the number of in-range taps, output dims use ceil, :129-130)
//   softmax         layers/softmax_layer.h:33-53 (over the whole C*H*W of an image)
// InnerProduct needs no kernel of its own: it is a 1x1 convolution over a 1x1 image with C*H*W input channels and runs
// through the implicit-GEMM path (split-K takes care of the tiny N = batch).
#include <float.h>

#include <algorithm>

#include "common.h"
#include "feather_hip/feather_net.h"

namespace fhip
{

// One float4 per thread, no grid-stride loop: the fastest streaming form on this chip (tools/copy_probe.hip: 6.2 TB/s against
// 4.3-5.4 for any looped variant).  Threads [n4, n4 + tail) finish the count % 4 (or unaligned) remainder one float each.
__global__ __launch_bounds__(256) void relu_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n4, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4)
    {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = fmaxf(v.x, 0.f);
        v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f);
        v.w = fmaxf(v.w, 0.f);
        reinterpret_cast<float4*>(y)[i] = v;
        return;
    }
    const size_t t = n4 * 4 + (i - n4);
    if (t < n) y[t] = fmaxf(x[t], 0.f);
}

template <bool RELU>
__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ y, const float* __restrict__ a, const float* __restrict__ b,
                                                 size_t n4, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4)
    {
        const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
        float4 r = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
        if (RELU)
        {
            r.x = fmaxf(r.x, 0.f);
            r.y = fmaxf(r.y, 0.f);
            r.z = fmaxf(r.z, 0.f);
            r.w = fmaxf(r.w, 0.f);
        }
        reinterpret_cast<float4*>(y)[i] = r;
        return;
    }
    const size_t t = n4 * 4 + (i - n4);
    if (t < n)
    {
        const float r = a[t] + b[t];
        y[t] = RELU ? fmaxf(r, 0.f) : r;
    }
}

// y[n][c][:] = x[n][c][:] * mul[c] + add[c]  (+ ReLU); one float4 (VEC: HW % 4 == 0, so it stays inside a plane) or one
// float per lane, no loop
template <bool RELU, bool VEC>
__global__ __launch_bounds__(256) void affine_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ mul,
                                                    const float* __restrict__ add, int C, int HW, size_t total)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t e = VEC ? i * 4 : i;
    const int c = (int)((e / HW) % C);
    const float m = mul[c], a = add ? add[c] : 0.f;
    if (VEC)
    {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = v.x * m + a;
        v.y = v.y * m + a;
        v.z = v.z * m + a;
        v.w = v.w * m + a;
        if (RELU)
        {
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f);
        }
        reinterpret_cast<float4*>(y)[i] = v;
    }
    else
    {
        const float v = x[i] * m + a;
        y[i] = RELU ? fmaxf(v, 0.f) : v;
    }
}

struct PoolParams
{
    int planes, H, W, OH, OW, KH, KW, SH, SW;
    int off_y, off_x; // window origin offset = pad_top + pad_bottom / pad_left + pad_right (reference quirk)
    int average;
};

__global__ __launch_bounds__(256) void pooling_kernel(float* __restrict__ y, const float* __restrict__ x, const PoolParams q, long long total)
{
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256)
    {
        const int ox = (int)(idx % q.OW);
        const long long t = idx / q.OW;
        const int oy = (int)(t % q.OH);
        const long long plane = t / q.OH;
        const float* xp = x + plane * q.H * q.W;
        const int y0 = oy * q.SH - q.off_y, x0 = ox * q.SW - q.off_x;
        const int ya = max(y0, 0), yb = min(y0 + q.KH, q.H);
        const int xa = max(x0, 0), xb = min(x0 + q.KW, q.W);
        float total_v = q.average ? 0.f : -FLT_MAX;
        int counter = 0;
        for (int yy = ya; yy < yb; ++yy)
            for (int xx = xa; xx < xb; ++xx)
            {
                const float v = xp[yy * q.W + xx];
                if (q.average)
                {
                    total_v += v;
                    ++counter;
                }
                else
                    total_v = total_v > v ? total_v : v;
            }
        y[idx] = q.average ? total_v / counter : total_v; // empty window: 0/0 = NaN exactly like the reference
    }
}

// 3x3 / stride 2 / unpadded MAX pooling (ResNet pool1, SqueezeNet): 4 consecutive outputs per lane from 3 rows of 9 inputs, read as
// two 16-byte vectors + one scalar per row (the generic kernel issues 36 scalar loads for the same 4 outputs).  Needs W % 4 == 0
// (aligned vectors); the clipped last window of ceil mode is handled by the column / row guards.
__global__ __launch_bounds__(256) void maxpool3s2_kernel(float* __restrict__ y, const float* __restrict__ x, int planes, int H, int W, int OH,
                                                        int OW, int ow4, long long total)
{
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256)
    {
        const int xq = (int)(idx % ow4);
        const long long t = idx / ow4;
        const int oy = (int)(t % OH);
        const long long plane = t / OH;
        const int ox = xq * 4, ix = ox * 2;
        const float* xp = x + plane * H * W;
        float m0 = -FLT_MAX, m1 = -FLT_MAX, m2 = -FLT_MAX, m3 = -FLT_MAX;
        // all nine loads unconditional, from clamped addresses (a load under a branch is waited for on the spot): a row past the image repeats
        // the last one (the maximum does not change), columns past it are replaced by -FLT_MAX after the load
        const bool has_b = ix + 4 < W, has_c = ix + 8 < W;
        float4 a[3], b[3];
        float c[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
        {
            const float* row = xp + (size_t)min(oy * 2 + r, H - 1) * W + ix;
            a[r] = *reinterpret_cast<const float4*>(row); // ix + 3 < W always (W % 4 == 0)
            b[r] = *reinterpret_cast<const float4*>(row + (has_b ? 4 : 0));
            c[r] = row[has_c ? 8 : 0];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
        {
            if (!has_b) b[r] = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
            if (!has_c) c[r] = -FLT_MAX;
            m0 = fmaxf(m0, fmaxf(fmaxf(a[r].x, a[r].y), a[r].z));
            m1 = fmaxf(m1, fmaxf(fmaxf(a[r].z, a[r].w), b[r].x));
            m2 = fmaxf(m2, fmaxf(fmaxf(b[r].x, b[r].y), b[r].z));
            m3 = fmaxf(m3, fmaxf(fmaxf(b[r].z, b[r].w), c[r]));
        }
        float* yp = y + (plane * OH + oy) * OW + ox;
        if ((OW & 3) == 0) *reinterpret_cast<float4*>(yp) = make_float4(m0, m1, m2, m3); // whole, aligned quads
        else
        {
            yp[0] = m0;
            if (ox + 1 < OW) yp[1] = m1;
            if (ox + 2 < OW) yp[2] = m2;
            if (ox + 3 < OW) yp[3] = m3;
        }
    }
}

// global pooling: one wave per (n, c) plane, lanes stride over the plane (coalesced), butterfly reduction
template <bool AVG>
__global__ __launch_bounds__(256) void plane_reduce_kernel(float* __restrict__ y, const float* __restrict__ x, int planes, int HW)
{
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (plane >= planes) return;
    const float* xp = x + (size_t)plane * HW;
    float v = AVG ? 0.f : -FLT_MAX;
    for (int i = lane; i < HW; i += 64) v = AVG ? v + xp[i] : fmaxf(v, xp[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
        const float t = __shfl_xor(v, o);
        v = AVG ? v + t : fmaxf(v, t);
    }
    if (lane == 0) y[plane] = AVG ? v / HW : v;
}

// global pooling of SMALL planes (HW < 128; the 7 x 7 planes in front of every classifier): the wave-per-plane form above spends a whole
// wave, one 4-byte load per lane and a six-step butterfly on 49 values (MobileNet-V1 b256: 262 144 planes, 34 us for 51 MB).  Here a block copies
// 128 consecutive planes -- one contiguous, 16-byte aligned run of the tensor -- to LDS with coalesced float4 loads (plane stride HW | 1: odd,
// conflict-free) and lane t sums plane t in index order, the order of the reference's loop (layers/pooling_layer.h:60-75).
constexpr int kPlaneReducePB = 128;
template <bool AVG>
__global__ __launch_bounds__(256) void plane_reduce_small_kernel(float* __restrict__ y, const float* __restrict__ x, int planes, int HW)
{
    extern __shared__ __attribute__((aligned(16))) float smem[]; // [PB][HW | 1]
    const int tid = threadIdx.x, hwp = HW | 1;
    const int plane0 = blockIdx.x * kPlaneReducePB, np = min(kPlaneReducePB, planes - plane0);
    const float* src = x + (size_t)plane0 * HW; // 128 * HW floats per block: 16-byte aligned
    const int count = np * HW, n4 = count >> 2;
    for (int i = tid; i < n4; i += 256)
    {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        const float e[4] = {v.x, v.y, v.z, v.w};
        int pl = (4 * i) / HW, r = 4 * i - pl * HW;
#pragma unroll
        for (int c = 0; c < 4; ++c)
        {
            smem[pl * hwp + r] = e[c];
            if (++r == HW)
            {
                r = 0;
                ++pl;
            }
        }
    }
    for (int i = (n4 << 2) + tid; i < count; i += 256) smem[(i / HW) * hwp + (i - (i / HW) * HW)] = src[i];
    __syncthreads();
    if (tid < np)
    {
        const float* p = smem + tid * hwp;
        float v = AVG ? 0.f : -FLT_MAX;
        for (int i = 0; i < HW; ++i) v = AVG ? v + p[i] : fmaxf(v, p[i]);
        y[plane0 + tid] = AVG ? v / HW : v;
    }
}

// one block per image: max, exp-sum, normalise over `cols` values
__global__ __launch_bounds__(256) void softmax_kernel(float* __restrict__ y, const float* __restrict__ x, int cols)
{
    __shared__ float red[256];
    const float* xp = x + (size_t)blockIdx.x * cols;
    float* yp = y + (size_t)blockIdx.x * cols;
    float m = -FLT_MAX;
    for (int i = threadIdx.x; i < cols; i += 256) m = fmaxf(m, xp[i]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1)
    {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    m = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < cols; i += 256)
    {
        const float e = expf(xp[i] - m);
        yp[i] = e;
        sum += e;
    }
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1)
    {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    sum = red[0];
    for (int i = threadIdx.x; i < cols; i += 256) yp[i] = yp[i] / sum;
}

// threads needed: one per float4 plus one per leftover float
static unsigned ew_grid(size_t n4, size_t n) { return (unsigned)((n4 + (n - n4 * 4) + 255) / 256); }

} // namespace fhip

using namespace fhip;

extern "C"
{

int fhip_relu(float* y, const float* x, size_t count, void* stream)
{
    if (!y || !x) return fail(FHIP_E_BADARG, "null pointer");
    if (count == 0) return FHIP_OK;
    const bool vec = (((uintptr_t)y | (uintptr_t)x) & 15) == 0;
    const size_t n4 = vec ? count / 4 : 0;
    hipLaunchKernelGGL(relu_kernel, dim3(ew_grid(n4, count)), dim3(256), 0, (hipStream_t)stream, y, x, n4, count);
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

int fhip_add(float* y, const float* a, const float* b, size_t count, int relu, void* stream)
{
    if (!y || !a || !b) return fail(FHIP_E_BADARG, "null pointer");
    if (count == 0) return FHIP_OK;
    const bool vec = (((uintptr_t)y | (uintptr_t)a | (uintptr_t)b) & 15) == 0;
    const size_t n4 = vec ? count / 4 : 0;
    const dim3 grid(ew_grid(n4, count));
    if (relu)
        hipLaunchKernelGGL(add_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, y, a, b, n4, count);
    else
        hipLaunchKernelGGL(add_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, y, a, b, n4, count);
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

int fhip_affine(float* y, const float* x, const float* mul, const float* add, int batch, int channels, int hw, int relu, void* stream)
{
    if (!y || !x || !mul || batch < 1 || channels < 1 || hw < 1) return fail(FHIP_E_BADARG, "bad argument");
    const size_t count = (size_t)batch * channels * hw;
    const bool vec = (hw % 4) == 0 && (((uintptr_t)y | (uintptr_t)x) & 15) == 0;
    const size_t total = vec ? count / 4 : count;
    if ((total + 255) / 256 > 0x7fffffffULL) return fail(FHIP_E_BADARG, "tensor too large");
    const dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
    if (relu && vec)
        hipLaunchKernelGGL((affine_kernel<true, true>), grid, dim3(256), 0, s, y, x, mul, add, channels, hw, total);
    else if (relu)
        hipLaunchKernelGGL((affine_kernel<true, false>), grid, dim3(256), 0, s, y, x, mul, add, channels, hw, total);
    else if (vec)
        hipLaunchKernelGGL((affine_kernel<false, true>), grid, dim3(256), 0, s, y, x, mul, add, channels, hw, total);
    else
        hipLaunchKernelGGL((affine_kernel<false, false>), grid, dim3(256), 0, s, y, x, mul, add, channels, hw, total);
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

int fhip_pooling_output_dim(const fhip_pool_param* p, int* out_h, int* out_w)
{
    if (!p || !out_h || !out_w) return fail(FHIP_E_BADARG, "null argument");
    if (p->global_pooling)
    {
        *out_h = 1;
        *out_w = 1;
        return FHIP_OK;
    }
    if (p->stride_h < 1 || p->stride_w < 1) return fail(FHIP_E_BADARG, "stride < 1");
    // ceil, layers/pooling_layer.h:129-130
    *out_h = (int)ceilf((float)(p->input_h + p->pad_top + p->pad_bottom - p->kernel_h) / p->stride_h) + 1;
    *out_w = (int)ceilf((float)(p->input_w + p->pad_left + p->pad_right - p->kernel_w) / p->stride_w) + 1;
    return FHIP_OK;
}

int fhip_pooling(const fhip_pool_param* p, int batch, float* y, const float* x, void* stream)
{
    if (!p || !y || !x || batch < 1 || p->channels < 1) return fail(FHIP_E_BADARG, "bad argument");
    PoolParams q;
    int oh = 0, ow = 0;
    int rc = fhip_pooling_output_dim(p, &oh, &ow);
    if (rc) return rc;
    if (oh < 1 || ow < 1) return fail(FHIP_E_BADARG, "empty pooling output");
    q.planes = batch * p->channels;
    q.H = p->input_h;
    q.W = p->input_w;
    q.OH = oh;
    q.OW = ow;
    q.KH = p->global_pooling ? p->input_h : p->kernel_h;
    q.KW = p->global_pooling ? p->input_w : p->kernel_w;
    q.SH = p->global_pooling ? 1 : p->stride_h;
    q.SW = p->global_pooling ? 1 : p->stride_w;
    q.off_y = p->pad_top + p->pad_bottom;
    q.off_x = p->pad_left + p->pad_right;
    q.average = p->pooling_type != 0;
    const long long total = (long long)q.planes * oh * ow;
    // whole-plane windows (global pooling, or a kernel covering the unpadded image): one wave per plane, coalesced
    if (oh == 1 && ow == 1 && q.off_y == 0 && q.off_x == 0 && q.KH >= q.H && q.KW >= q.W)
    {
        const int planes = q.planes;
        const int hw = q.H * q.W;
        if (hw < 128 && ((uintptr_t)x & 15) == 0) // 128 planes x (hw | 1) floats stay within the 64 KB a launch gets without asking
        {
            const dim3 grid(ceil_div(planes, kPlaneReducePB));
            const size_t lds = (size_t)kPlaneReducePB * (hw | 1) * sizeof(float);
            if (q.average) hipLaunchKernelGGL(plane_reduce_small_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, y, x, planes, hw);
            else hipLaunchKernelGGL(plane_reduce_small_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, y, x, planes, hw);
            FHIP_CHECK_HIP(hipGetLastError());
            return FHIP_OK;
        }
        if (q.average)
            hipLaunchKernelGGL(plane_reduce_kernel<true>, dim3(ceil_div(planes, 4)), dim3(256), 0, (hipStream_t)stream, y, x, planes, q.H * q.W);
        else
            hipLaunchKernelGGL(plane_reduce_kernel<false>, dim3(ceil_div(planes, 4)), dim3(256), 0, (hipStream_t)stream, y, x, planes, q.H * q.W);
        FHIP_CHECK_HIP(hipGetLastError());
        return FHIP_OK;
    }
    if (!q.average && q.KH == 3 && q.KW == 3 && q.SH == 2 && q.SW == 2 && q.off_y == 0 && q.off_x == 0 && (q.W % 4) == 0 &&
        (((uintptr_t)x | (uintptr_t)y) & 15) == 0)
    {
        const int ow4 = ceil_div(ow, 4);
        const long long items = (long long)q.planes * oh * ow4;
        hipLaunchKernelGGL(maxpool3s2_kernel, dim3((unsigned)std::min<long long>(256 * 32, (items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y,
                           x, q.planes, q.H, q.W, oh, ow, ow4, items);
        FHIP_CHECK_HIP(hipGetLastError());
        return FHIP_OK;
    }
    const int grid = (int)std::min<long long>(256 * 16, (total + 255) / 256); // looped: measured faster than one output per lane here
    hipLaunchKernelGGL(pooling_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, x, q, total);
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

int fhip_softmax(float* y, const float* x, int batch, int count_per_image, void* stream)
{
    if (!y || !x || batch < 1 || count_per_image < 1) return fail(FHIP_E_BADARG, "bad argument");
    hipLaunchKernelGGL(softmax_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, y, x, count_per_image);
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

} // extern "C"
