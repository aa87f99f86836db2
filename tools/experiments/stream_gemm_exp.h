// stream_gemm.h -- EXPERIMENT (tools only): a 1x1 / stride-1 convolution as a register-streamed GEMM without LDS and without
// barriers.  A wave owns 32 output channels x 128 consecutive pixels: per pair of input channels every lane loads ONE float4 of the
// activation (lanes 0-31: channel 2j, pixels 4*l .. 4*l+3; lanes 32-63: channel 2j+1) -- component t of that float4 is the B operand
// of MFMA t (v_mfma_f32_32x32x2_f32: B[k = lane / 32][n = lane % 32]), so the four MFMAs of a step compute the four interleaved
// pixel sets {4n + t}.  The accumulators then hold, per register, four CONSECUTIVE pixels across (acc0..acc3): the result leaves as
// 16 dwordx4 stores (512 contiguous bytes per half-wave) with no transpose.  Weights are pre-packed as the A operand image
// ([m-group][channel pair][lane]) and stream through a register ring as one coalesced dword load per step.
#pragma once
#include <hip/hip_runtime.h>

namespace fhip
{
typedef float f32x16s __attribute__((ext_vector_type(16)));

struct StreamParams
{
    const float* in;   // [N][C][HW]
    const float* wp;   // [K/32][C/2][64]: wp[mg][j][lane] = W[32 mg + lane % 32][2 j + lane / 32]
    const float* bias; // [K] or null
    float* out;        // [N][K][HW]
    int C, K, HW, N, relu;
    long long total_px; // N * HW
    int mgroups;        // K / 32
    int px_tiles;       // ceil(total_px / 128)
    int mg_per_block;   // waves of a block take consecutive m-groups of one pixel tile (they share the activation through L1)
};

__global__ void stream_pack_weights(float* wp, const float* w, int K, int C)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)K * C;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    const long long rest = i >> 6;
    const int J = C / 2;
    const int j = (int)(rest % J), mg = (int)(rest / J);
    wp[i] = w[(size_t)(32 * mg + (lane & 31)) * C + 2 * j + (lane >> 5)];
}

typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef f32x16s st_acc_t;
// measurement: every 64th block adds its shader-clock and 100 MHz wall-clock ticks (first wave): effective shader clock of the kernel
static __device__ unsigned long long g_stream_clock_probe[2];
// hipcc sinks ordinary loads towards their uses (it kept two of the eight float4 of the ring in flight): the ring is written with
// inline-asm loads and counted waits.  vmcnt retires in order, so with P requests outstanding "s_waitcnt vmcnt(P - 2)" says the oldest
// two (one B float4, one A dword) have landed; the registers are operands of the wait so that their uses cannot be hoisted above it.
#define STREAM_LD4(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr))
#define STREAM_LD1(dst, ptr) asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(ptr))
#define STREAM_WAIT(n, b_, a_) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b_), "+v"(a_) : "n"(n))

#define STREAM_WAIT2(n, b_, a0_, a1_) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(b_), "+v"(a0_), "+v"(a1_) : "n"(n))

// MG = 32-row groups of output channels per wave (1: 32 x 128 wave tile, 2: 64 x 128 -- the activation float4 feeds 8 MFMAs)
template <int D, int WAVES, int MG, bool XCD = true, int ABL = 0>
__global__ __launch_bounds__(64 * WAVES) void stream_pw_kernel(const StreamParams q)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long probe_c0 = clock64(), probe_w0 = wall_clock64();
    const int wgroups = q.mgroups / MG; // K % (32 * MG) == 0
    const int mg_blocks = (wgroups + WAVES - 1) / WAVES;
    // XCD-aware: workgroup i runs on XCD i % 8; consecutive virtual ids (= the m-groups of one pixel tile, then the next pixel tile)
    // land on ONE XCD, so an activation tile is fetched into one L2 instead of eight
    int vid = blockIdx.x;
    if (XCD)
    {
        const int nwg = gridDim.x, qx = nwg / 8, rx = nwg % 8, xcd = vid % 8, local = vid / 8;
        vid = ((xcd < rx) ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + local;
    }
    const int pt = vid / mg_blocks, wg = (vid - pt * mg_blocks) * WAVES + wave;
    if (wg >= wgroups) return;
    const int mg = wg * MG;
    const int half = lane >> 5, l31 = lane & 31;
    const long long g = (long long)pt * 128 + 4 * l31;
    const bool ok = g < q.total_px;
    const long long gc = ok ? g : 0;
    const int n = (int)(gc / q.HW), p = (int)(gc - (long long)n * q.HW);
    const float* bp = q.in + ((size_t)n * q.C + half) * q.HW + p;
    const float* ap = q.wp + (size_t)mg * (q.C / 2) * 64 + lane;
    const size_t astride = (size_t)(q.C / 2) * 64; // next m-group
    const size_t bstep = (size_t)2 * q.HW;
    const int J = q.C / 2;

    f32x16s acc[MG][4];
#pragma unroll
    for (int m = 0; m < MG; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

    f32x4s b[D];
    float a[D][2];
#pragma unroll
    for (int u = 0; u < D; ++u)
    {
        STREAM_LD4(b[u], bp + (size_t)u * bstep);
#pragma unroll
        for (int m = 0; m < MG; ++m) STREAM_LD1(a[u][m], ap + m * astride + (size_t)u * 64);
    }
    constexpr int PER = 1 + MG; // requests per step
    const float* bnext = bp + (size_t)D * bstep;
    const float* anext = ap + (size_t)D * 64;
    if (ABL & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int j0 = 0; j0 < J - D; j0 += D)
    {
#pragma unroll
        for (int u = 0; u < D; ++u)
        {
            if (ABL & 1)
                asm volatile("" : "+v"(b[u]), "+v"(a[u][0])); // ablation: no requests in the loop, the ring's first contents are reused
            else if (MG == 1)
                STREAM_WAIT(PER * D - PER, b[u], a[u][0]);
            else
                STREAM_WAIT2(PER * D - PER, b[u], a[u][0], a[u][1]);
#pragma unroll
            for (int m = 0; m < MG; ++m)
            {
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], b[u].x, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], b[u].y, acc[m][1], 0, 0, 0);
                acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], b[u].z, acc[m][2], 0, 0, 0);
                acc[m][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], b[u].w, acc[m][3], 0, 0, 0);
            }
            if (!(ABL & 1))
            {
                STREAM_LD4(b[u], bnext + (size_t)u * bstep);
#pragma unroll
                for (int m = 0; m < MG; ++m) STREAM_LD1(a[u][m], anext + m * astride + (size_t)u * 64);
            }
        }
        bnext += (size_t)D * bstep;
        anext += (size_t)D * 64;
    }
#pragma unroll
    for (int u = 0; u < D; ++u)
    {
        if (ABL & 1)
            asm volatile("" : "+v"(b[u]), "+v"(a[u][0]));
        else if (MG == 1)
            STREAM_WAIT(PER * (D - u) - PER, b[u], a[u][0]);
        else
            STREAM_WAIT2(PER * (D - u) - PER, b[u], a[u][0], a[u][1]);
#pragma unroll
        for (int m = 0; m < MG; ++m)
        {
            acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], b[u].x, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], b[u].y, acc[m][1], 0, 0, 0);
            acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], b[u].z, acc[m][2], 0, 0, 0);
            acc[m][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], b[u].w, acc[m][3], 0, 0, 0);
        }
    }
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0)
    {
        atomicAdd(&g_stream_clock_probe[0], (unsigned long long)(clock64() - probe_c0));
        atomicAdd(&g_stream_clock_probe[1], (unsigned long long)(wall_clock64() - probe_w0));
    }
    if (!ok) return;
    if (ABL & 2)
    {
        // ablation: one store per lane keeps the accumulators alive
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < MG; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[m][t][r];
        q.out[((size_t)n * q.K + 32 * mg + 4 * half) * q.HW + p] = sum;
        return;
    }
#pragma unroll
    for (int m = 0; m < MG; ++m)
    {
        float* op = q.out + ((size_t)n * q.K + 32 * (mg + m) + 4 * half) * q.HW + p;
        const float* bsp = q.bias + 32 * (mg + m) + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
            const int row = (r & 3) + 8 * (r >> 2);
            const float bs = bsp[row];
            float4 v = make_float4(acc[m][0][r] + bs, acc[m][1][r] + bs, acc[m][2][r] + bs, acc[m][3][r] + bs);
            if (q.relu)
            {
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
            }
            *reinterpret_cast<float4*>(op + (size_t)row * q.HW) = v;
        }
    }
}

// ---- split-C variant: the WAVES = 4 waves of a block split the input channels of (4 / SPLIT) m-groups of one pixel tile between them
// (SPLIT = 2 or 4) and add their accumulators through LDS in a fixed order.  Four (two) times as many, four (two) times shorter waves:
// the last partial round of the grid -- the waves that run while most SIMDs are already idle -- shrinks accordingly.
template <int D, int SPLIT, int ABL = 0>
__global__ __launch_bounds__(256) void stream_pw_split_kernel(const StreamParams q)
{
    extern __shared__ __attribute__((aligned(16))) float red[]; // [slot][16 regs x 4][64 lanes] float4-interleaved
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int MGB = 4 / SPLIT; // m-groups per block
    const int mg_blocks = q.mgroups / MGB;
    int vid = blockIdx.x;
    {
        const int nwg = gridDim.x, qx = nwg / 8, rx = nwg % 8, xcd = vid % 8, local = vid / 8;
        vid = ((xcd < rx) ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + local;
    }
    const int pt = vid / mg_blocks, mg = (vid - pt * mg_blocks) * MGB + wave / SPLIT, cp = wave % SPLIT;
    const int half = lane >> 5, l31 = lane & 31;
    const long long g = (long long)pt * 128 + 4 * l31;
    const bool ok = g < q.total_px;
    const long long gc = ok ? g : 0;
    const int n = (int)(gc / q.HW), p = (int)(gc - (long long)n * q.HW);
    const int J = q.C / 2 / SPLIT; // steps of this wave: channels [cp * C / SPLIT, (cp + 1) * C / SPLIT); a multiple of D
    const float* bp = q.in + ((size_t)n * q.C + half + (size_t)cp * 2 * J) * q.HW + p;
    const float* ap = q.wp + ((size_t)mg * (q.C / 2) + (size_t)cp * J) * 64 + lane;
    const size_t bstep = (size_t)2 * q.HW;

    f32x16s acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4s b[D];
    float a[D];
#pragma unroll
    for (int u = 0; u < D; ++u)
    {
        STREAM_LD4(b[u], bp + (size_t)u * bstep);
        STREAM_LD1(a[u], ap + (size_t)u * 64);
    }
    const float* bnext = bp + (size_t)D * bstep;
    const float* anext = ap + (size_t)D * 64;
    for (int j0 = 0; j0 < J - D; j0 += D)
    {
#pragma unroll
        for (int u = 0; u < D; ++u)
        {
            STREAM_WAIT(2 * D - 2, b[u], a[u]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].w, acc[3], 0, 0, 0);
            STREAM_LD4(b[u], bnext + (size_t)u * bstep);
            STREAM_LD1(a[u], anext + (size_t)u * 64);
        }
        bnext += (size_t)D * bstep;
        anext += (size_t)D * 64;
    }
#pragma unroll
    for (int u = 0; u < D; ++u)
    {
        STREAM_WAIT(2 * (D - u) - 2, b[u], a[u]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].w, acc[3], 0, 0, 0);
    }
    // ---- fixed-order reduction: (c0 + c2) + (c1 + c3) for SPLIT = 4, c0 + c1 for SPLIT = 2.  A slot = one wave's 64 accumulator
    // registers as 16 float4 per lane ([reg][lane] float4: conflict-free 16-byte accesses)
    constexpr int SLOT = 16 * 64 * 4;
    float4* const slots = reinterpret_cast<float4*>(red);
    auto put = [&](int slot) {
#pragma unroll
        for (int r = 0; r < 16; ++r) slots[(slot * 16 + r) * 64 + lane] = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
    };
    auto add = [&](int slot) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
            const float4 v = slots[(slot * 16 + r) * 64 + lane];
            acc[0][r] += v.x;
            acc[1][r] += v.y;
            acc[2][r] += v.z;
            acc[3][r] += v.w;
        }
    };
    (void)SLOT;
    const int grp = wave / SPLIT; // m-group of the block this wave works for
    if (SPLIT == 4)
    {
        if (cp >= 2) put(cp - 2);
        __syncthreads();
        if (cp < 2) add(cp);
        if (cp == 1) put(2);
        __syncthreads();
        if (cp == 0) add(2);
    }
    else
    {
        if (cp == 1) put(grp);
        __syncthreads();
        if (cp == 0) add(grp);
    }
    if (cp != 0 || !ok) return;
    float* op = q.out + ((size_t)n * q.K + 32 * mg + 4 * half) * q.HW + p;
    const float* bsp = q.bias + 32 * mg + 4 * half;
#pragma unroll
    for (int r = 0; r < 16; ++r)
    {
        const int row = (r & 3) + 8 * (r >> 2);
        const float bs = bsp[row];
        float4 v = make_float4(acc[0][r] + bs, acc[1][r] + bs, acc[2][r] + bs, acc[3][r] + bs);
        if (q.relu)
        {
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<float4*>(op + (size_t)row * q.HW) = v;
    }
}

// ---- persistent variant: a wave walks TPW pixel tiles of its m-group (tiles pt, pt + tile_stride, ...) with the request ring running
// across tile boundaries: the first D steps of the next tile are requested during the last D steps of this one, so neither the
// prologue latency nor the 16 stores of the epilogue leave the matrix pipe idle.  Stores count in vmcnt like loads and retire in
// order with them: the first D waits of a tile allow 16 more outstanding operations (the stores issued just before).
struct StreamTiles
{
    int tiles_per_wave, tile_stride; // wave w of m-group g handles pixel tiles first, first + tile_stride, ...
};

#define STREAM_STEP(WAITN, LDB, LDA)                                                \
    STREAM_WAIT(WAITN, b[u], a[u]);                                                  \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].x, acc[0], 0, 0, 0);    \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].y, acc[1], 0, 0, 0);    \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].z, acc[2], 0, 0, 0);    \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u].w, acc[3], 0, 0, 0);    \
    STREAM_LD4(b[u], LDB);                                                           \
    STREAM_LD1(a[u], LDA)

template <int D>
__global__ __launch_bounds__(256) void stream_pw_persistent_kernel(const StreamParams q, const StreamTiles ts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mg_blocks = (q.mgroups + 3) / 4;
    int vid = blockIdx.x;
    {
        const int nwg = gridDim.x, qx = nwg / 8, rx = nwg % 8, xcd = vid % 8, local = vid / 8;
        vid = ((xcd < rx) ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + local;
    }
    const int pt0 = vid / mg_blocks, mg = (vid - pt0 * mg_blocks) * 4 + wave;
    if (mg >= q.mgroups || pt0 >= q.px_tiles) return;
    const int half = lane >> 5, l31 = lane & 31;
    const float* ap = q.wp + (size_t)mg * (q.C / 2) * 64 + lane;
    const size_t bstep = (size_t)2 * q.HW;
    const int J = q.C / 2; // >= 2 * D, a multiple of D
    float bs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bs[r] = q.bias[32 * mg + 4 * half + (r & 3) + 8 * (r >> 2)];

    // per-lane addresses of a pixel tile
    auto locate = [&](int pt, const float*& bp, float*& op, bool& ok) {
        const long long g = (long long)pt * 128 + 4 * l31;
        ok = g < q.total_px;
        const long long gc = ok ? g : 0;
        const int n = (int)(gc / q.HW), p = (int)(gc - (long long)n * q.HW);
        bp = q.in + ((size_t)n * q.C + half) * q.HW + p;
        op = q.out + ((size_t)n * q.K + 32 * mg + 4 * half) * q.HW + p;
    };
    const float* bp;
    float* op;
    bool ok;
    locate(pt0, bp, op, ok);

    st_acc_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4s b[D];
    float a[D];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < D; ++u)
    {
        STREAM_LD4(b[u], bp + (size_t)u * bstep);
        STREAM_LD1(a[u], ap + (size_t)u * 64);
    }
    int pt = pt0;
    for (int i = 0; i < ts.tiles_per_wave; ++i)
    {
        const int ptn = pt + ts.tile_stride;
        const bool more = (i + 1 < ts.tiles_per_wave) && ptn < q.px_tiles;
        const float* bpn;
        float* opn;
        bool okn;
        locate(more ? ptn : pt, bpn, opn, okn);
        // first D steps: behind the previous tile's 16 stores (none before the first tile)
        {
            const float* bnext = bp + (size_t)D * bstep;
            const float* anext = ap + (size_t)D * 64;
            if (i == 0)
            {
#pragma unroll
                for (int u = 0; u < D; ++u) { STREAM_STEP(2 * D - 2, bnext + (size_t)u * bstep, anext + (size_t)u * 64); }
            }
            else
            {
#pragma unroll
                for (int u = 0; u < D; ++u) { STREAM_STEP(2 * D - 2 + 16, bnext + (size_t)u * bstep, anext + (size_t)u * 64); }
            }
        }
        // middle
        {
            const float* bnext = bp + (size_t)2 * D * bstep;
            const float* anext = ap + (size_t)2 * D * 64;
            for (int j0 = D; j0 < J - D; j0 += D)
            {
#pragma unroll
                for (int u = 0; u < D; ++u) { STREAM_STEP(2 * D - 2, bnext + (size_t)u * bstep, anext + (size_t)u * 64); }
                bnext += (size_t)D * bstep;
                anext += (size_t)D * 64;
            }
        }
        // last D steps: request the first D steps of the next tile (of this one again when there is none: drained below)
#pragma unroll
        for (int u = 0; u < D; ++u) { STREAM_STEP(2 * D - 2, bpn + (size_t)u * bstep, ap + (size_t)u * 64); }
        // epilogue: 16 stores (lane 0 of a tile is always inside the tensor, so the stores are always issued)
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
            const int row = (r & 3) + 8 * (r >> 2);
            float4 v = make_float4(acc[0][r] + bs[r], acc[1][r] + bs[r], acc[2][r] + bs[r], acc[3][r] + bs[r]);
            if (q.relu)
            {
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
            }
            if (ok) *reinterpret_cast<float4*>(op + (size_t)row * q.HW) = v;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        if (!more) break;
        pt = ptn;
        bp = bpn;
        op = opn;
        ok = okn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the requests made for a tile that does not exist
}
} // namespace fhip
