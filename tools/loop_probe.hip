// tools/loop_probe.hip -- which ingredient of the GEMM k loop costs MFMA issue slots?  Synthetic loop with the same
// instruction mix as gemm_core.h (2x2 32x32x2 MFMAs per k-pair, operands via ds_read from an LDS tile, optional
// barrier and LDS writes per k-tile), no global memory.  Reports TF per variant and waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// FLAGS: 1 = operands from LDS every k-pair, 2 = barrier per k-tile, 4 = ds_write_b128 x4 per k-tile,
//        8 = prefetch the next k-pair's operands before the MFMAs (forced with sched_barrier)
template <int FLAGS, int BLOCKS_PER_CU>
__global__ __launch_bounds__(256, BLOCKS_PER_CU) void loop_kernel(const float* in, float* out, int ktiles)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * 16 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 16 * 256; i += 256) lds[i] = in[i & 511];
    __syncthreads();
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
    const float* as0 = lds + half * 128 + wm * 64 + l31;
    const float* bs0 = lds + 2 * 16 * 128 + half * 128 + wn * 64 + l31;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float a0 = in[tid], a1 = in[tid + 1], b0 = in[tid + 2], b1 = in[tid + 3];
    float4 w = make_float4(a0, a1, b0, b1);
    int cur = 0;
    for (int kt = 0; kt < ktiles; ++kt)
    {
        if (FLAGS & 4)
        {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&lds[(cur ^ 1) * 16 * 128 + (i & 1) * 2 * 16 * 128 + (tid >> 5) * 128 + (tid & 31) * 4 + (i >> 1) * 1024]) = w;
        }
        const float* as = as0 + cur * 16 * 128;
        const float* bs = bs0 + cur * 16 * 128;
        float na0, na1, nb0, nb1;
        if (FLAGS & 8)
        {
            a0 = as[0]; a1 = as[32]; b0 = bs[0]; b1 = bs[32];
        }
#pragma unroll
        for (int kp = 0; kp < 8; ++kp)
        {
            if (FLAGS & 8)
            {
                if (kp < 7)
                {
                    na0 = as[(2 * kp + 2) * 128]; na1 = as[(2 * kp + 2) * 128 + 32];
                    nb0 = bs[(2 * kp + 2) * 128]; nb1 = bs[(2 * kp + 2) * 128 + 32];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            else if (FLAGS & 1)
            {
                a0 = as[2 * kp * 128]; a1 = as[2 * kp * 128 + 32];
                b0 = bs[2 * kp * 128]; b1 = bs[2 * kp * 128 + 32];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            if (FLAGS & 8)
            {
                __builtin_amdgcn_sched_barrier(0);
                if (kp < 7) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
            }
        }
        if (FLAGS & 2) __syncthreads();
        if (FLAGS & 6) cur ^= 1;
    }
    float s = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int FLAGS, int BPC>
void run(const char* name, const float* in, float* out, int ktiles)
{
    const int blocks = 256 * BPC;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL((loop_kernel<FLAGS, BPC>), dim3(blocks), dim3(256), 0, 0, in, out, ktiles);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((loop_kernel<FLAGS, BPC>), dim3(blocks), dim3(256), 0, 0, in, out, ktiles);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double flop = (double)blocks * 4 * ktiles * 32 * 4096.0;
    printf("%-44s waves/SIMD %d  %8.3f ms  %7.2f TF (%.1f%%)\n", name, BPC, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
}

template <int BPC>
void all(const float* in, float* out, int kt)
{
    run<0, BPC>("regs only", in, out, kt);
    run<1, BPC>("lds operands", in, out, kt);
    run<9, BPC>("lds operands, prefetched (sched_barrier)", in, out, kt);
    run<3, BPC>("lds operands + barrier", in, out, kt);
    run<7, BPC>("lds operands + barrier + ds_write", in, out, kt);
    run<15, BPC>("prefetched + barrier + ds_write", in, out, kt);
    run<2, BPC>("regs + barrier", in, out, kt);
}

int main()
{
    float *in, *out;
    (void)hipMalloc(&in, 1024 * 4);
    (void)hipMalloc(&out, 1024 * 256 * 4);
    std::vector<float> h(1024);
    for (auto& x : h) x = (rand() / (float)RAND_MAX) * 2 - 1;
    (void)hipMemcpy(in, h.data(), 1024 * 4, hipMemcpyHostToDevice);
    all<1>(in, out, 2000);
    all<2>(in, out, 2000);
    all<4>(in, out, 1000);
    return 0;
}
