// tools/clock_probe.hip -- what does the fp32 MFMA pipe sustain on this chip, and at which clock?
// Pure-register MFMA loops (no LDS, no memory) with (a) zero operands, (b) random operands; reports TF and
// the effective shader clock (s_memtime ticks / wall_clock64 100 MHz ticks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool M16>
__global__ __launch_bounds__(256) void mfma_loop(const float* in, float* out, long long* clk, int iters)
{
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x16 acc[NACC];
    f32x4 acc4[NACC];
    for (int i = 0; i < NACC; ++i)
    {
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    }
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
        {
            if (M16)
                acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
            else
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i)
    {
        for (int r = 0; r < 16; ++r) s += acc[i][r];
        for (int r = 0; r < 4; ++r) s += acc4[i][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0)
    {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}

template <int NACC, bool M16>
void run(const char* name, const float* in, float* out, long long* clk, int blocks, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((mfma_loop<NACC, M16>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((mfma_loop<NACC, M16>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < blocks; ++i)
    {
        cyc += h[2 * i];
        wall += h[2 * i + 1];
    }
    const double flop = (double)blocks * 4 * iters * NACC * (M16 ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2);
    printf("%-28s blocks %5d  %8.3f ms  %7.2f TF  shader clock %.0f MHz (memtime/wall_clock64@100MHz)  cyc/mfma/SIMD-wave %.1f\n", name,
           blocks, ms, flop / ms / 1e9, cyc / wall * 100.0, cyc / blocks / ((double)iters * NACC));
}

int main()
{
    float *in, *out;
    long long* clk;
    hipMalloc(&in, 512 * 4);
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&clk, 4096 * 16);
    std::vector<float> h(512, 0.f);
    for (int pass = 0; pass < 2; ++pass)
    {
        if (pass == 1)
            for (auto& x : h) x = (rand() / (float)RAND_MAX) * 2 - 1;
        hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
        printf("== operands %s\n", pass ? "uniform(-1,1)" : "zero");
        run<4, false>("32x32x2 4acc 1 wave/SIMD", in, out, clk, 256, 20000);
        run<4, false>("32x32x2 4acc 2 waves/SIMD", in, out, clk, 512, 20000);
        run<4, false>("32x32x2 4acc 4 waves/SIMD", in, out, clk, 1024, 20000);
        run<1, false>("32x32x2 1acc 4 waves/SIMD", in, out, clk, 1024, 40000);
        run<4, true>("16x16x4 4acc 2 waves/SIMD", in, out, clk, 512, 40000);
    }
    return 0;
}
