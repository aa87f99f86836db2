// tools/g4_bench.hip -- timing of the prototype fused tile GEMM + output transform (tools/experiments/wino_gemm_out.h) on VGG-16 conv1_2 b32
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wino_gemm_out.h"
using namespace fhip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const int batch = argc > 1 ? atoi(argv[1]) : 32;
    const int NT = argc > 2 ? atoi(argv[2]) : 2;
    const int TX = 38, T = TX * TX, P = T * batch, blocks = (P + 16 * NT - 1) / (16 * NT);
    WinoGemmOutParams p;
    float *V4, *U4, *bias, *out;
    const size_t vn = (size_t)blocks * 64 * 1024 * NT;
    CK(hipMalloc(&V4, vn * 4));
    CK(hipMalloc(&U4, (size_t)64 * 4096 * 4));
    CK(hipMalloc(&bias, 64 * 4));
    CK(hipMalloc(&out, (size_t)batch * 64 * 112 * 112 * 4));
    std::vector<float> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    for (size_t off = 0; off < vn; off += h.size()) CK(hipMemcpy(V4 + off, h.data(), std::min(h.size(), vn - off) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(U4, h.data(), 64 * 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data(), 64 * 4, hipMemcpyHostToDevice));
    p.V4 = V4; p.U4 = U4; p.bias = bias; p.out = out; p.TX = TX; p.T = T; p.P = P; p.OH = 224; p.OW = 224; p.relu = 1; p.has_bias = 1;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<double> ts;
    for (int r = 0; r < 8; ++r)
    {
        CK(hipEventRecord(a, 0));
        if (NT == 1) hipLaunchKernelGGL((wino_gemm_out_kernel<true, 1>), dim3(blocks), dim3(256), 0, 0, p);
        else hipLaunchKernelGGL((wino_gemm_out_kernel<true, 2>), dim3(blocks), dim3(256), 0, 0, p);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (r) ts.push_back(ms * 1e3);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    const double us = ts[ts.size() / 2], gf = 2.0 * 64 * 64 * 64 * (double)P / 1e9;
    printf("batch %d: %d blocks, fused tile GEMM + output transform + pooling: median %.1f us (min %.1f)  %.1f TF  V read %.2f TB/s\n", batch, blocks, us, ts[0], gf / us / 1e3,
           vn * 4 / us / 1e6);
    std::vector<float> o(1000);
    CK(hipMemcpy(o.data(), out, 4000, hipMemcpyDeviceToHost));
    double s = 0;
    for (float v : o) s += v;
    printf("checksum of the first 1000 outputs %.4f\n", s);
    return 0;
}
