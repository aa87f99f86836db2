// tests/cpp/host_forward_test.cpp -- the C++ host side end to end on a GPU: ConvParam / ConvBooster from
// include/booster/booster.h used the way feather::ConvLayer does (reference src/layers/conv_layer.h:92-172): Reshape
// (AssignOutputDim + SelectAlgo + GetBufferSize), Init once, Forward per batch -- with device buffers.  Checked against a
// plain fp64 direct convolution computed right here.
#include <booster/booster.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                    \
    do                                                                           \
    {                                                                            \
        hipError_t e = (x);                                                      \
        if (e != hipSuccess)                                                     \
        {                                                                        \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
            return 2;                                                            \
        }                                                                        \
    } while (0)

static float frand() { return (rand() / (float)RAND_MAX) * 2.f - 1.f; }

static int run_case(int C, int K, int H, int k, int s, int pad, int group, int batch, booster::ConvAlgo expect)
{
    booster::ConvParam p;
    memset(&p, 0, sizeof(p));
    p.input_channels = C;
    p.output_channels = K;
    p.input_h = p.input_w = H;
    p.kernel_h = p.kernel_w = k;
    p.stride_h = p.stride_w = s;
    p.pad_left = p.pad_right = p.pad_top = p.pad_bottom = pad;
    p.group = group;
    p.bias_term = true;
    p.activation = booster::ReLU;
    p.batch = batch;
    p.AssignOutputDim();
    booster::ConvBooster bst;
    if (bst.SelectAlgo(&p) != 0 || bst.Algo() != expect) return printf("selection failed\n"), 1;
    size_t buf_bytes = 0, pk_bytes = 0;
    if (bst.GetBufferSizeBytes(&p, &buf_bytes, &pk_bytes) != 0) return printf("GetBufferSizeBytes failed\n"), 1;

    const int cpg = C / group, OH = p.output_h, OW = p.output_w;
    std::vector<float> x((size_t)batch * C * H * H), w((size_t)p.output_channels * cpg * k * k), b(p.output_channels);
    for (auto& v : x) v = frand();
    for (auto& v : w) v = frand() / std::sqrt((float)(cpg * k * k));
    for (auto& v : b) v = 0.1f * frand();
    std::vector<float> y((size_t)batch * p.output_channels * OH * OW);

    float *dx, *dw, *db, *dy, *dpk, *dbuf;
    CK(hipMalloc(&dx, x.size() * 4));
    CK(hipMalloc(&dw, w.size() * 4));
    CK(hipMalloc(&db, b.size() * 4));
    CK(hipMalloc(&dy, y.size() * 4));
    CK(hipMalloc(&dpk, pk_bytes ? pk_bytes : 4));
    CK(hipMalloc(&dbuf, buf_bytes ? buf_bytes : 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    booster::SetStream(st);
    if (bst.Init(&p, dpk, dw) != 0) return printf("Init failed\n"), 1;
    if (bst.Forward(&p, dy, dx, dpk, dbuf, db, 1) != 0) return printf("Forward failed\n"), 1;
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));

    double maxerr = 0, maxref = 0;
    for (int n = 0; n < batch; ++n)
        for (int ko = 0; ko < p.output_channels; ++ko)
            for (int oy = 0; oy < OH; ++oy)
                for (int ox = 0; ox < OW; ++ox)
                {
                    double acc = b[ko];
                    for (int cc = 0; cc < cpg; ++cc)
                    {
                        const int c = group > 1 ? ko : cc;
                        for (int u = 0; u < k; ++u)
                            for (int v = 0; v < k; ++v)
                            {
                                const int yy = oy * s - pad + u, xx = ox * s - pad + v;
                                if (yy < 0 || yy >= H || xx < 0 || xx >= H) continue;
                                acc += (double)x[(((size_t)n * C + c) * H + yy) * H + xx] * w[(((size_t)ko * cpg + cc) * k + u) * k + v];
                            }
                    }
                    if (acc < 0) acc = 0;
                    const double got = y[(((size_t)n * p.output_channels + ko) * OH + oy) * OW + ox];
                    maxerr = std::fmax(maxerr, std::fabs(got - acc));
                    maxref = std::fmax(maxref, std::fabs(acc));
                }
    printf("C%d K%d H%d k%d s%d g%d batch %d algo %d: normalised max error %.3e\n", C, K, H, k, s, group, batch, (int)bst.Algo(), maxerr / maxref);
    hipFree(dx), hipFree(dw), hipFree(db), hipFree(dy), hipFree(dpk), hipFree(dbuf);
    return (maxerr / maxref <= 1e-4) ? 0 : 1;
}

int main()
{
    int bad = 0;
    bad += run_case(16, 32, 20, 3, 1, 1, 1, 3, booster::WINOGRADF63);
    bad += run_case(24, 40, 14, 1, 1, 0, 1, 2, booster::IM2COL);
    bad += run_case(3, 16, 33, 7, 2, 3, 1, 2, booster::IM2COL);
    bad += run_case(16, 16, 28, 3, 2, 1, 16, 2, booster::DEPTHWISE);
    if (bad == 0) printf("host forward ok\n");
    return bad;
}
