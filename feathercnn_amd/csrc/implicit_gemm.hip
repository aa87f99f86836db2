// implicit_gemm.hip -- the IM2COL route of ConvBooster on gfx950 as an implicit GEMM:
//   out[K x (N*Ho*Wo)] = W[K x C*kh*kw] * col,   col[(c,u,v)][(n,oy,ox)] = in[n][c][oy*sh+u-pt][ox*sw+v-pl] or 0
// Replaces IM2COL_GetBufferSize/Init/Forward (reference src/booster/avx/booster.cpp:64-102): booster::im2col
// (avx/generic_kernels.cpp:50-85) is never materialised -- the B-operand loader of gemm_core.h gathers the
// column matrix straight from the NCHW input into LDS (zero for out-of-bounds taps, same rule as :66-69) --
// and packed_sgemm_init / packed_sgemm_activation<bias,relu> (avx/sgemm.cpp:312-433) become a transposed,
// zero-padded weight panel Wt[Cp*kh*kw][Kp] plus the shared fp32-MFMA main loop with the bias / ReLU
// epilogue fused into the accumulator store.  It serves exactly what AVX SelectAlgo routes to IM2COL
// (avx/booster.cpp:294-303): 1x1 s1/s2, 7x7 s2, 3x3 with H <= 8 or C % 4 != 0, 3x3 s2; and NAIVE
// (avx/booster.cpp:28-61, which ignores activation).
#include <string.h>

#include <algorithm>
#include <mutex>

#include "gemm_core.h"
#include "conv_gemm_policy.h"
#include "dwpw_band.h"
#include "stream_gemm.h"
#include "ip_stream.h"

namespace fhip
{

size_t depthwise_packed_floats(const fhip_conv_param& p, size_t* w12_offset); // depthwise.hip

using ConvShapeBig = GemmShape<128, 64, 16, 2, 2>;
using ConvShapeSmallM = GemmShape<64, 128, 16, 1, 4>;
// (Bounding the register allocation for 6 blocks per CU instead of the 5 that 86 - 96 registers give -- 80 registers, 44 - 132 bytes of scratch per
// lane -- was measured in round 5: ResNet-50 b64 14 847 -> 13 804 img/s.  EXPERIMENTS.md A8.)
// A 64x64 tile (GemmShape<64, 64, 16, 2, 2, 8>: 4x the blocks, 8 blocks per CU) was measured against split-K on ResNet-50's
// under-filled layers and made no difference (C1024->K256 @14 b64: 0.093 vs 0.092 ms; C256->K64 @56: 0.077 vs 0.075): not kept.

static bool conv_small_m(int K) { return K <= 64; }
// InnerProduct at small batch: with N <= 32 columns a 64-column tile spends half its MFMAs on padding, and fp32 MFMA is slow
// enough that this bounds the rate the weight matrix can be streamed at (128x64 tile: 8 B of weights per clk per CU = 4.5 TB/s
// at best).  A 64x32 tile of two waves doubles that bound.
using ConvShapeNarrow = GemmShape<64, 32, 16, 2, 1, 8>;
static bool conv_narrow_n(long long ntot) { return ntot <= 32; }

// ---- the register-streamed 1x1 route (stream_gemm.h) ----------------------------------------------------------------------------
// eligible: a pure function of the geometry (decides the packed-weight size); profitable: where it was measured faster than the
// LDS-tiled kernel (deep reduction, narrow-to-medium output) -- both independent of everything but the param and the batch, so that
// GetBufferSize, Init and Forward agree.
static bool stream_eligible(const fhip_conv_param& p)
{
    return p.group == 1 && p.kernel_h == 1 && p.kernel_w == 1 && p.stride_h <= 1 && p.stride_w <= 1 && p.pad_left == 0 && p.pad_right == 0 &&
           p.pad_top == 0 && p.pad_bottom == 0 && p.input_channels % 16 == 0 && p.output_channels % 32 == 0 && p.output_h * p.output_w >= 4 &&
           p.output_h == p.input_h && p.output_w == p.input_w; // Ho*Wo % 4 != 0 runs the RAGGED form of the kernel
}
static bool stream_profitable(const fhip_conv_param& p, int batch)
{
    return stream_eligible(p) && p.input_channels >= 256 && p.output_channels >= 128 && p.output_channels <= 512 &&
           (long long)batch * p.output_h * p.output_w >= 4096;
}
// ---- the weight-streaming InnerProduct route (ip_stream.h) -----------------------------------------------------------------------
// eligible: a 1x1 convolution over a 1x1 image (what feather::InnerProductLayer is on the device) -- decides the packed size;
// profitable: at most one MFMA column tile of images and a weight matrix worth streaming (VGG-16's fc6 / fc7 / fc8 at batch <= 32).
// (the channel thresholds are geometry, not batch: they sit in ip_eligible so that small InnerProduct / 1x1 layers, which never take the
// route at any batch, do not carry a second copy of their weights -- ADVICE r03)
static bool ip_eligible(const fhip_conv_param& p)
{
    return p.group == 1 && p.kernel_h == 1 && p.kernel_w == 1 && p.input_h == 1 && p.input_w == 1 && p.output_h == 1 && p.output_w == 1 &&
           p.pad_left == 0 && p.pad_right == 0 && p.pad_top == 0 && p.pad_bottom == 0 && p.input_channels >= 1024 && p.output_channels >= 256;
}
static bool ip_profitable(const fhip_conv_param& p, int batch) { return ip_eligible(p) && batch >= 1 && batch <= 32; }
static int ip_kg(const fhip_conv_param& p) { return ceil_div(p.output_channels, 32); }
static int ip_kq(const fhip_conv_param& p) { return ceil_div(p.input_channels, 8); }
static size_t ip_packed_floats(const fhip_conv_param& p) { return (size_t)ip_kg(p) * ip_kq(p) * 256; }
// pieces of the reduction: about three 4-wave blocks per CU, at least 8 octets (32 MFMAs) per piece
static int ip_pieces(const fhip_conv_param& p)
{
    const int want = device_compute_units() * 3 / std::max(1, ceil_div(ip_kg(p), 4)); // ~3 blocks (of 4 m-groups) per CU: tools/ip_stream_bench.hip
    return std::max(1, std::min(want, ip_kq(p) / 8));
}
static size_t ip_xq_floats(const fhip_conv_param& p) { return round_up_sz((size_t)ip_kq(p) * 256, 64); }

// ---- 1x1 / stride-1 / unpadded layers on planes whose size is not a multiple of 4 (ResNet-50's 7 x 7 stage): ConvGemmPolicy<5> works on pixel
// SLOTS -- ceil(Ho*Wo / 4) groups of 4 per image, the last one shifted back inside the image -- so the GEMM has batch * 4 * ceil(Ho*Wo / 4)
// columns instead of batch * Ho*Wo (conv_gemm_policy.h).  Pure functions of geometry and batch: GetBufferSize, the split-K decision and Forward agree.
static bool ragged_1x1(const fhip_conv_param& p, int batch)
{
    const int ohw = p.output_h * p.output_w;
    return p.group == 1 && p.kernel_h == 1 && p.kernel_w == 1 && p.stride_h <= 1 && p.stride_w <= 1 && p.pad_left == 0 && p.pad_right == 0 &&
           p.pad_top == 0 && p.pad_bottom == 0 && p.output_h == p.input_h && p.output_w == p.input_w && (ohw % 4) != 0 && ohw >= 4 &&
           !conv_narrow_n((long long)batch * ohw);
}
static long long igemm_columns(const fhip_conv_param& p, int batch)
{
    const long long ohw = (long long)p.output_h * p.output_w;
    return ragged_1x1(p, batch) ? (long long)batch * 4 * ((ohw + 3) / 4) : (long long)batch * ohw;
}

void igemm_packed_dims(const fhip_conv_param& p, int* kd_padded, int* k_padded);
size_t igemm_packed_floats(const fhip_conv_param& p)
{
    int kdp, kp;
    igemm_packed_dims(p, &kdp, &kp);
    // [ Wt: the LDS-tiled kernel's panels | wp: the streamed 1x1 kernel's A-operand image (eligible 1x1 layers only) | the streamed
    //   InnerProduct kernel's A-operand image (1x1 convolutions over a 1x1 image only) ]
    return (size_t)kdp * kp + (stream_eligible(p) ? (size_t)p.output_channels * p.input_channels : 0) + (ip_eligible(p) ? ip_packed_floats(p) : 0);
}

void igemm_packed_dims(const fhip_conv_param& p, int* kd_padded, int* k_padded)
{
    const int Kd = p.input_channels * p.kernel_h * p.kernel_w;
    *kd_padded = round_up(Kd, kConvKTile);
    *k_padded = round_up(p.output_channels, conv_small_m(p.output_channels) ? 64 : 128);
}

// Split-K decision (pure function of the geometry and the batch: GetBufferSize and Forward must agree).
// Grids below ~2 blocks per CU leave the matrix pipes idle (ResNet-50's 3x3 and 1x1 layers at 7x7: 196 tiles of 288
// k-tiles each); the reduction is cut into S equal parts, S a divisor of the k-tile count.
static int igemm_split(const fhip_conv_param& p, int batch)
{
    if (stream_profitable(p, batch)) return 1; // the streamed kernel never splits
    if (ip_profitable(p, batch)) return 1;     // the streamed InnerProduct has its own pieces (ip_pieces)
    int kdp, kp;
    igemm_packed_dims(p, &kdp, &kp);
    const long long ntot = igemm_columns(p, batch);
    const bool narrow = conv_narrow_n((long long)batch * p.output_h * p.output_w);
    const int bm = (narrow || conv_small_m(p.output_channels)) ? 64 : 128, bn = narrow ? 32 : (conv_small_m(p.output_channels) ? 128 : 64);
    const long long tiles = (long long)(kp / bm) * ((ntot + bn - 1) / bn);
    const int kt = kdp / kConvKTile;
    if (tiles >= 512 || kt < 16) return 1;
    int want;
    if (kt >= 512)
        // InnerProduct-like shapes (a handful of tiles, thousands of k-tiles: VGG fc6 is 64 narrow tiles x 1568) stream the weight
        // matrix once and are HBM bound: they want many blocks in flight (fc6: 16 pieces measured best of 8..98)
        want = (int)std::min<long long>(32, (1024 + tiles - 1) / tiles);
    else
        // convolutions: as many pieces as keep ALL blocks resident at once (256 CUs x 5 blocks): equal blocks that start
        // together finish together, whereas a grid just above the resident count pays a second, nearly empty round
        want = (int)std::min<long long>(8, (long long)device_compute_units() * 5 / tiles);
    want = std::min(want, kt / 8); // at least 8 k-tiles per piece
    return std::max(want, 1);
}

// Tail split (conv_gemm_policy.h): for an UNSPLIT launch of the 128 x 64 tile whose tile count leaves a remainder of whole column tiles over the
// CU count -- those last column tiles are cut `pieces` ways along the reduction so that the remainder lands as one short block per CU (or on a
// part of the CUs) instead of a last round of lone full blocks.  Pure function of the geometry and the batch (GetBufferSize and Forward agree).
// -> first tail column tile (0 = no tail split), *pieces.
static int igemm_tail(const fhip_conv_param& p, int batch, int* pieces)
{
    *pieces = 1;
#ifdef FHIP_NO_TAIL_SPLIT
    return 0;
#endif
    if (p.group != 1 || p.kernel_h != 1 || p.kernel_w != 1 || p.pad_left || p.pad_right || p.pad_top || p.pad_bottom) return 0; // 1x1 routes only
    if (stream_profitable(p, batch) || ip_profitable(p, batch) || igemm_split(p, batch) > 1) return 0; // (the small-C kernel takes K <= 64 only: excluded below)
    if (conv_small_m(p.output_channels)) return 0;
    const long long ntot = igemm_columns(p, batch);
    if (conv_narrow_n((long long)batch * p.output_h * p.output_w)) return 0;
    int kdp, kp;
    igemm_packed_dims(p, &kdp, &kp);
    const int m_tiles = kp / 128, n_tiles = (int)((ntot + 63) / 64), kt = kdp / kConvKTile, cus = device_compute_units();
    const long long tiles = (long long)m_tiles * n_tiles;
    // measured per layer (ResNet-50 b64 / MobileNet-V1 b256, tools/tail_layers.sh): 3.25 rounds 82 -> 76 us, 6.1 rounds 76 -> 73, 6.5 rounds 261 ->
    // 248; at 12.25 rounds the pieces and their reduce launch cost what the shorter tail saves (+2 ... +3 us on 235 us): 8 rounds at most
    if (tiles < cus || tiles > 8LL * cus || (cus & 7)) return 0;
    const int rem = (int)(tiles % cus);
    if (rem == 0 || rem % m_tiles || (m_tiles * (n_tiles - rem / m_tiles)) % 8) return 0;
    int s = std::min(8, cus / rem);
    while (s > 1 && (kt / s < 4 || (rem * s) % 8)) --s;
    if (s < 2) return 0;
    *pieces = s;
    return n_tiles - rem / m_tiles;
}

bool igemm_streams(const fhip_conv_param& p, int batch) { return stream_profitable(p, batch); }

size_t igemm_buffer_bytes(const fhip_conv_param& p, int batch)
{
    if (ip_profitable(p, batch)) // [ xq: the re-packed activations | partial sums of the pieces ]
        return (ip_xq_floats(p) + (size_t)ip_pieces(p) * p.output_channels * batch) * sizeof(float);
    const int s = igemm_split(p, batch);
    if (s <= 1)
    {
        int pieces;
        const int nt0 = igemm_tail(p, batch, &pieces);
        if (!nt0) return 0;
        const long long tail_cols = ((igemm_columns(p, batch) + 63) / 64 - nt0) * 64; // partial[pieces][K][tail columns]
        return (size_t)pieces * p.output_channels * (size_t)tail_cols * sizeof(float);
    }
    return (size_t)s * p.output_channels * (size_t)igemm_columns(p, batch) * sizeof(float);
}

// finishes a split-K convolution: out[img][m][rem] = act(sum_s partial[s][m][n] + bias[m])
__global__ __launch_bounds__(256) void igemm_splitk_reduce_kernel(float* __restrict__ out, const float* __restrict__ partial,
                                                                 const float* __restrict__ bias, const float* __restrict__ residual, int K,
                                                                 int Ntot, int OHW, int S, int has_bias, int relu, int rag_gpi, int col0, int pnt)
{
    // columns col0 .. Ntot - 1; partial is [S][K][pnt] with column n at n - col0 (plain split-K: col0 = 0, pnt = Ntot)
    const int n = col0 + blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.y;
    if (n >= Ntot) return;
    int img, rem;
    if (rag_gpi)
    {
        // columns are pixel slots (ConvGemmPolicy<5>): slot e of group g4 / 4 is pixel min(g4, OHW - 4) + e; the components of an image's last
        // group that the group before already owns are dropped
        const int spi = 4 * rag_gpi, sl = n - (n / spi) * spi, g4 = sl & ~3, e = sl & 3, off = min(g4, OHW - 4);
        if (e < g4 - off) return;
        img = n / spi;
        rem = off + e;
    }
    else
    {
        img = n / OHW;
        rem = n - img * OHW;
    }
    const size_t stride = (size_t)K * pnt;
    const float* src = partial + (size_t)m * pnt + (n - col0);
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += src[(size_t)s * stride];
    if (has_bias) v += bias[m];
    const size_t o = ((size_t)img * K + m) * OHW + rem;
    if (residual) v += residual[o];
    if (relu) v = fmaxf(v, 0.f);
    out[o] = v;
}

// K7: Wt[k / bm][q][k % bm] = W[k][q], zero padded: panels of bm output channels, reduction index next, channel fastest
// (the GPU analogue of packed_sgemm_init's row panels, avx/sgemm.cpp:312-346).
__global__ __launch_bounds__(256) void igemm_pack_weights_kernel(float* __restrict__ Wt, const float* __restrict__ w, int K,
                                                                int Kd, int Kdp, int bm)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    for (int q = blockIdx.y; q < Kd; q += gridDim.y) Wt[((size_t)(k / bm) * Kdp + q) * bm + (k % bm)] = w[(size_t)k * Kd + q];
}

int igemm_init(const fhip_conv_param& p, float* packed, const float* kernel, hipStream_t s)
{
    if (p.group > 1) return fail(FHIP_E_UNSUPPORTED, "implicit GEMM handles group == 1 only");
    int kdp, kp;
    igemm_packed_dims(p, &kdp, &kp);
    const int Kd = p.input_channels * p.kernel_h * p.kernel_w;
    StageTimer tm(FHIP_STAGE_INIT, s);
    FHIP_CHECK_HIP(hipMemsetAsync(packed, 0, (size_t)kdp * kp * sizeof(float), s));
    hipLaunchKernelGGL(igemm_pack_weights_kernel, dim3(ceil_div(p.output_channels, 256), std::min(Kd, 65535)), dim3(256), 0, s, packed, kernel,
                       p.output_channels, Kd, kdp, conv_small_m(p.output_channels) ? 64 : 128);
    if (stream_eligible(p))
    {
        const size_t kc = (size_t)p.output_channels * p.input_channels;
        hipLaunchKernelGGL(stream_pack_weights_kernel, dim3((unsigned)((kc + 255) / 256)), dim3(256), 0, s, packed + (size_t)kdp * kp, kernel,
                           p.output_channels, p.input_channels);
    }
    if (ip_eligible(p))
    {
        const size_t off = (size_t)kdp * kp + (stream_eligible(p) ? (size_t)p.output_channels * p.input_channels : 0);
        const long long n4 = (long long)ip_kg(p) * ip_kq(p) * 64;
        hipLaunchKernelGGL(ip_pack_weights_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, packed + off, kernel, p.output_channels,
                           p.input_channels, ip_kg(p), ip_kq(p));
    }
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

// ---- small-C convolutions (the first layer of every benchmark net: 3 input channels) ---------------------------------------
// Measured against the generic gather (standalone, MI355X): VGG conv1_1 b32 0.158 vs 0.183 ms, ResNet conv1 b64 0.198 vs 0.271,
// MobileNet conv1 b256 0.192 vs 0.276.
// The generic gather above spends 4 scalar global loads + bounds checks per float4 of the column matrix and re-reads every
// input pixel kh*kw times through L1.  Here a persistent block stages the whole weight matrix (C*kh*kw <= 160 rows) and, per
// tile of 8 x 16 output pixels, the input patch ((8-1)*S+kh rows x (16-1)*S+kw columns x C) in LDS; the im2col gather then
// happens on LDS addresses (one ds_read_b32 per MFMA B operand, offset table per reduction row).  This is the "im2col in LDS"
// form SURVEY.md 8(a) names for K5.  Output tile = BM (32 or 64) channels x 128 pixels, 4 waves of BM x 32.
struct SmallCParams
{
    const float* Wt; // packed weights: K <= 64 means ONE panel [Kd16][64] (igemm_pack_weights_kernel)
    const float* in;
    float* out;
    const float* bias;
    int C, K, H, W, OH, OW, S, PL, PT, KH, KW, Kd, Kp, N;
    int kd2;           // MFMA k-steps: ceil(Kd / 2) rounded up to a multiple of 4 (the extra rows meet zero weights)
    int PH, PW, patch; // patch rows, columns, floats
    int patch_alloc;   // LDS floats reserved for the patch (patch rounded up to 64)
    int epi_alias;     // the epilogue's transpose buffer [4][16][36] lives in the patch area (which is at least that large)
    int tiles_x, tiles_y;
    long long tiles;
    int has_bias, relu;
    int tw_shift; // tile = (1 << tw_shift) columns x (128 >> tw_shift) rows: 32 x 4 (a wave stores whole 128-byte lines) or 16 x 8
};

constexpr int kSmallCMaxPatch = 2816; // 11 passes of 256 lanes

template <int TM, int PASSES> // PASSES * 256 >= patch floats
__global__ __launch_bounds__(256) void conv_smallc_kernel(const SmallCParams q)
{
    constexpr int BM = 32 * TM, EPI_LD = 36;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                                        // [2*kd2][BM]
    float* const patch = As + (size_t)2 * q.kd2 * BM;              // [patch_alloc]
    int* const koff = reinterpret_cast<int*>(patch + q.patch_alloc); // [2][kd2]: koff[h*kd2 + kp] = patch offset of reduction row 2*kp + h
    // epilogue transpose buffer [4 waves][16 rows][EPI_LD] (two half passes per 32 x 32 accumulator).  Where the patch area is large enough it IS
    // the patch area (q.epi_alias: ResNet-50's 7 x 7 conv1 -- 38.9 KB of weights + 9.5 KB of patch + 18.4 KB of transpose buffer were 67 KB = two
    // blocks per CU; without the separate buffer three fit) at the price of one more barrier per tile: no wave may still be reading the patch
    float* const scr = q.epi_alias ? patch : reinterpret_cast<float*>(koff + 2 * q.kd2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    // ---- once per block: weights, reduction-row table
    for (int i = tid; i < 2 * q.kd2 * (BM / 4); i += 256)
    {
        const int row = i / (BM / 4), c4 = i - row * (BM / 4);
        *reinterpret_cast<float4*>(As + (size_t)row * BM + c4 * 4) = *reinterpret_cast<const float4*>(q.Wt + (size_t)row * q.Kp + c4 * 4);
    }
    const int KHW = q.KH * q.KW, PHW = q.PH * q.PW;
    for (int k = tid; k < 2 * q.kd2; k += 256)
    {
        const int c = k / KHW, rem = k - c * KHW, u = rem / q.KW, w = rem - u * q.KW;
        koff[(k & 1) * q.kd2 + (k >> 1)] = k < q.Kd ? c * PHW + u * q.PW + w : 0; // padded rows multiply zero weights
    }
    const int pix = wave * 32 + l31;
    const int TW = 1 << q.tw_shift, TH = 128 >> q.tw_shift;
    const int poff = (pix >> q.tw_shift) * q.S * q.PW + (pix & (TW - 1)) * q.S;
    const size_t HW = (size_t)q.H * q.W;

    // patch of tile `tt` -> registers: every load is issued unconditionally (clamped address), zeros are selected later.  TWO tiles are kept
    // in flight per block where the registers allow (sets A and B, TM == 1: up to 32 output channels; round 4): with one, a block's tile took
    // as long as an HBM round trip under load (MobileNet-V1 conv1 b256: 4 blocks per CU x 1 tile each = 1.25 us per tile and CU, 140 of the
    // kernel's 174 us, the matrix pipe 23 % busy): MobileNet-V1 b256 69 820 -> 70 503 img/s.  With 64 output channels (TM == 2, ResNet-50's 7 x 7
    // conv1: 144 -> 184 registers, 3 -> 2 blocks per CU) the second set loses 0.6 % of ResNet-50 b64: one set there.
    constexpr int DEPTH = TM == 1 ? 2 : 1;
    struct Pre
    {
        float pv[PASSES];
        unsigned pok;
        int n, ty, tx; // tile index -> (image, tile row, tile column), decoded once, in 32-bit arithmetic (host: tiles < 2^31)
    };
    int pck[PASSES]; // the lane's patch elements, decoded once: c << 16 | row << 8 | column
#pragma unroll
    for (int j = 0; j < PASSES; ++j)
    {
        const int ee = min(j * 256 + tid, q.patch - 1);
        const int c = ee / PHW, rem = ee - c * PHW, r = rem / q.PW, x = rem - r * q.PW;
        pck[j] = (c << 16) | (r << 8) | x;
    }
    __syncthreads(); // weights and the reduction-row table are in place
    const int ntiles = (int)q.tiles, G = (int)gridDim.x;
    auto fetch_patch = [&](Pre& P, int tt) __attribute__((always_inline)) {
        const int t2 = tt / q.tiles_x;
        P.tx = tt - t2 * q.tiles_x;
        P.n = t2 / q.tiles_y;
        P.ty = t2 - P.n * q.tiles_y;
        const int iy0 = P.ty * TH * q.S - q.PT, ix0 = P.tx * TW * q.S - q.PL;
        const float* img = q.in + (size_t)P.n * q.C * HW;
        P.pok = 0;
#pragma unroll
        for (int j = 0; j < PASSES; ++j)
        {
            const int c = pck[j] >> 16, r = (pck[j] >> 8) & 0xff, x = pck[j] & 0xff;
            const int iy = iy0 + r, ix = ix0 + x;
            P.pok |= ((unsigned)iy < (unsigned)q.H && (unsigned)ix < (unsigned)q.W) ? (1u << j) : 0u;
            const int cy = min(max(iy, 0), q.H - 1), cx = min(max(ix, 0), q.W - 1);
            P.pv[j] = img[(size_t)c * HW + (size_t)cy * q.W + cx];
        }
    };

    auto do_tile = [&](Pre& P, int t) __attribute__((always_inline)) {
        const int tx_t = P.tx, ty_t = P.ty, n = P.n;
        const int oy0 = ty_t * TH, ox0 = tx_t * TW;
        __syncthreads(); // every wave is done reading the previous patch
#pragma unroll
        for (int j = 0; j < PASSES; ++j)
            if (j * 256 + tid < q.patch) patch[j * 256 + tid] = (P.pok & (1u << j)) ? P.pv[j] : 0.f;
        __syncthreads();
        // this register set's next tile (two ahead) is requested now, ahead of this tile's MFMAs and output stores: vmcnt retires in order and
        // counts stores, so loads issued behind the stores would wait for the stores to drain
        fetch_patch(P, min(t + DEPTH * G, ntiles - 1));

        f32x16 acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // 4 k-steps per trip: the offsets of the NEXT trip are fetched (one 16-byte LDS read) while this trip's 4 B operands,
        // 4*TM A operands and MFMAs are in flight -- otherwise every k-step is a chain of two dependent LDS round trips
        const float* ap = As + half * BM + l31;
        const int4* kq = reinterpret_cast<const int4*>(koff + half * q.kd2);
        const int groups = q.kd2 >> 2;
        int4 ko = kq[0];
        for (int g = 0; g < groups; ++g)
        {
            const int4 kn = kq[min(g + 1, groups - 1)];
            const float b0 = patch[ko.x + poff], b1 = patch[ko.y + poff], b2 = patch[ko.z + poff], b3 = patch[ko.w + poff];
            const float* a = ap + (size_t)(8 * g) * BM;
            float fa[4][TM];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[u][i] = a[(size_t)(2 * u) * BM + i * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][i], b0, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][i], b1, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2][i], b2, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3][i], b3, acc[i], 0, 0, 0);
            ko = kn;
        }

        // ---- epilogue: per-wave LDS transpose (16 rows at a time: registers 8 h .. 8 h + 7 of the C/D layout are rows 16 h .. 16 h + 15), bias +
        // ReLU, 16-byte stores of 4 consecutive pixels of one output row
        if (q.epi_alias) __syncthreads(); // the transpose buffer is the patch: every wave has left its k-loop
        float* const ws = scr + wave * (16 * EPI_LD);
        const int e_row = lane >> 3, e_c4 = (lane & 7) * 4;
        const int p4 = wave * 32 + e_c4, oy = oy0 + (p4 >> q.tw_shift), ox = ox0 + (p4 & (TW - 1));
#pragma unroll
        for (int ih = 0; ih < 2 * TM; ++ih)
        {
            const int i = ih >> 1, h = ih & 1;
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) ws[((r8 & 3) + 8 * (r8 >> 2) + 4 * half) * EPI_LD + l31] = acc[i][8 * h + r8];
#pragma unroll
            for (int qd = 0; qd < 2; ++qd)
            {
                const int m = i * 32 + h * 16 + qd * 8 + e_row;
                float4 v = *reinterpret_cast<const float4*>(&ws[(qd * 8 + e_row) * EPI_LD + e_c4]);
                if (m < q.K && oy < q.OH && ox < q.OW)
                {
                    if (q.has_bias)
                    {
                        const float bb = q.bias[m];
                        v.x += bb;
                        v.y += bb;
                        v.z += bb;
                        v.w += bb;
                    }
                    if (q.relu)
                    {
                        v.x = fmaxf(v.x, 0.f);
                        v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f);
                        v.w = fmaxf(v.w, 0.f);
                    }
                    float* o = q.out + (((size_t)n * q.K + m) * q.OH + oy) * q.OW + ox;
                    if (ox + 3 < q.OW && (q.OW & 3) == 0)
                        stg4_act<1>(o, v);
                    else
                    {
                        o[0] = v.x;
                        if (ox + 1 < q.OW) o[1] = v.y;
                        if (ox + 2 < q.OW) o[2] = v.z;
                        if (ox + 3 < q.OW) o[3] = v.w;
                    }
                }
            }
        }
    };

    Pre A, B;
    const int t0 = (int)blockIdx.x;
    if (t0 < ntiles)
    {
        fetch_patch(A, t0);
        if (DEPTH == 2) fetch_patch(B, min(t0 + G, ntiles - 1));
    }
    for (int t = t0; t < ntiles; t += DEPTH * G)
    {
        do_tile(A, t);
        if (DEPTH == 2 && t + G < ntiles) do_tile(B, t + G);
    }
}

// tile width of the small-C kernel: 32 x 4 tiles when they divide the row (a wave then stores whole 128-byte lines: VGG conv1_1
// 0.158 vs 0.180 ms), else 16 x 8 (no wasted columns on 112-pixel rows: ResNet conv1 0.198 vs 0.223 ms)
// (round 4, tools/variant_ab.sh: 32 x 4 tiles on 112-pixel rows too -- 128-byte instead of 64-byte output pieces, the fourth tile of a row half
// empty -- measured: MobileNet-V1 b256 68 987 vs 69 010 images/s, ResNet-50 b64 14 334 vs 14 416: not taken)
static int smallc_tile_w(int ow) { return (ow % 32) == 0 ? 32 : 16; }

// does the small-C kernel take this geometry?  (pure function of the param: forward and the tests agree)
static bool smallc_applicable(const fhip_conv_param& p)
{
    const int s = p.stride_h > 0 ? p.stride_h : 1;
    if (p.group != 1 || p.stride_h != p.stride_w || s > 2 || p.output_channels > 64) return false;
    const int kd = p.input_channels * p.kernel_h * p.kernel_w;
    // the patch of the tile smallc_forward will really use (32 x 4 when 32 divides the row, else 16 x 8: the 16 x 8 patch is the
    // larger one for s > k or kw > kh)
    const int tw = smallc_tile_w(p.output_w), th = 128 / tw;
    const int ph = (th - 1) * s + p.kernel_h, pw = (tw - 1) * s + p.kernel_w;
    // output rows must take 16-byte stores: with scalar stores (SqueezeNet's 111-pixel rows) the generic kernel is faster
    return kd <= 160 && p.input_channels <= 8 && ph < 256 && pw < 256 && p.input_channels * ph * pw <= kSmallCMaxPatch &&
           (p.output_w % 4) == 0;
}

static int smallc_forward(const fhip_conv_param& p, int batch, float* out, const float* in, const float* packed, const float* bias, bool relu,
                          hipStream_t s)
{
    SmallCParams q;
    q.Wt = packed;
    q.in = in;
    q.out = out;
    q.bias = bias;
    q.C = p.input_channels;
    q.K = p.output_channels;
    q.H = p.input_h;
    q.W = p.input_w;
    q.OH = p.output_h;
    q.OW = p.output_w;
    q.S = p.stride_h > 0 ? p.stride_h : 1;
    q.PL = p.pad_left;
    q.PT = p.pad_top;
    q.KH = p.kernel_h;
    q.KW = p.kernel_w;
    q.Kd = q.C * q.KH * q.KW;
    int kdp;
    igemm_packed_dims(p, &kdp, &q.Kp);
    q.N = batch;
    q.kd2 = round_up((q.Kd + 1) / 2, 4); // 2*kd2 <= Kd rounded up to 8 <= the packed matrix's Kd16 rows (zero padded)
    q.tw_shift = smallc_tile_w(q.OW) == 32 ? 5 : 4;
    const int tw = 1 << q.tw_shift, th = 128 >> q.tw_shift;
    q.PH = (th - 1) * q.S + q.KH;
    q.PW = (tw - 1) * q.S + q.KW;
    q.patch = q.C * q.PH * q.PW;
    q.tiles_x = ceil_div(q.OW, tw);
    q.tiles_y = ceil_div(q.OH, th);
    q.tiles = (long long)batch * q.tiles_x * q.tiles_y;
    if (q.tiles > 0x7fffffffLL) return fail(FHIP_E_BADARG, "small-C convolution: more than 2^31 output tiles");
    q.has_bias = p.bias_term != 0;
    q.relu = relu;
    const int tm = q.K <= 32 ? 1 : 2, bm = 32 * tm;
    q.patch_alloc = round_up(q.patch, 64);
    // the transpose buffer of the epilogue (4 waves x 16 rows x 36 floats = 9 KB): a region of its own, or -- where that costs a resident block and
    // the patch area is large enough (ResNet-50's 7 x 7 conv1: 38.9 KB of weights, 9.5 KB of patch) -- the patch area itself
    const size_t lds_base = ((size_t)2 * q.kd2 * bm + q.patch_alloc) * sizeof(float) + ((size_t)2 * q.kd2) * sizeof(int);
    const size_t scr_bytes = (size_t)4 * 16 * 36 * sizeof(float);
    const int cap = q.S == 1 ? 6 : 4;
    auto resident = [&](size_t bytes) { return (int)std::min<size_t>(cap, device_lds_bytes() / (bytes + 512)); };
#ifdef FHIP_SMALLC_NO_ALIAS
    q.epi_alias = 0;
#else
    q.epi_alias = (size_t)q.patch_alloc * sizeof(float) >= scr_bytes && resident(lds_base) > resident(lds_base + scr_bytes);
#endif
    const size_t lds = lds_base + (q.epi_alias ? 0 : scr_bytes);
    // resident blocks per CU (the grid is persistent): as many as the LDS takes, up to 6 at stride 1 and 4 at stride 2 (tools/conv1_bench.py,
    // same box: VGG conv1_1 b32 137 / 126 / 119 / 118 us with 3 / 4 / 5 / 6, MobileNet conv1 b256 191 / 183 / 191 / 191; an XCD-contiguous
    // tile order changed nothing)
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(q.S == 1 ? 6 : 4, device_lds_bytes() / (lds + 512)));
    const int grid = (int)std::min<long long>(q.tiles, (long long)device_compute_units() * per_cu);
    const int passes = ceil_div(q.patch, 256);
#define FHIP_SMALLC_LAUNCH(TM_, P_)                                                                                               \
    do                                                                                                                            \
    {                                                                                                                             \
        /* dynamic LDS above 64 KB must be allowed once per kernel AND device: one std::once_flag per device (threads of one process  */ \
        /* drive different devices, tests/cpp/multi_gpu_main.cpp); devices beyond the table set the attribute on every launch         */ \
        static std::once_flag attr_once[64];                                                                                      \
        static hipError_t attr_err[64];                                                                                           \
        int dev_ = 0;                                                                                                             \
        FHIP_CHECK_HIP(hipGetDevice(&dev_));                                                                                      \
        auto set_ = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_smallc_kernel<TM_, P_>),                  \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); };                   \
        if (dev_ < 0 || dev_ >= 64)                                                                                               \
            FHIP_CHECK_HIP(set_());                                                                                               \
        else                                                                                                                      \
        {                                                                                                                         \
            std::call_once(attr_once[dev_], [&] { attr_err[dev_] = set_(); });                                                    \
            FHIP_CHECK_HIP(attr_err[dev_]);                                                                                       \
        }                                                                                                                         \
        hipLaunchKernelGGL((conv_smallc_kernel<TM_, P_>), dim3(grid), dim3(256), lds, s, q);                                      \
    } while (0)
    if (tm == 1)
    {
        if (passes <= 3) FHIP_SMALLC_LAUNCH(1, 3);
        else if (passes <= 7) FHIP_SMALLC_LAUNCH(1, 7);
        else FHIP_SMALLC_LAUNCH(1, 11);
    }
    else
    {
        if (passes <= 3) FHIP_SMALLC_LAUNCH(2, 3);
        else if (passes <= 7) FHIP_SMALLC_LAUNCH(2, 7);
        else FHIP_SMALLC_LAUNCH(2, 11);
    }
#undef FHIP_SMALLC_LAUNCH
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

template <class Shape, int MODE, bool TWIN = false>
static void launch(const ConvGemmParams& g0, hipStream_t s)
{
    ConvGemmParams g = g0;
    g.m_tiles = g.Kp / Shape::BM;
    g.n_tiles = ceil_div(g.Ntot, Shape::BN);
    g.batches = g.split_k;
    unsigned grid = (unsigned)(g.batches * g.m_tiles * g.n_tiles);
    if (g.tail_nt0 > 0)
    {
        // tail split (igemm_tail): column tiles [0, tail_nt0) unsplit, the rest in split_k pieces each
        g.tail_main = g.m_tiles * g.tail_nt0;
        g.part_col0 = g.tail_nt0 * Shape::BN;
        g.part_ntot = (g.n_tiles - g.tail_nt0) * Shape::BN;
        grid = (unsigned)(g.tail_main + g.m_tiles * (g.n_tiles - g.tail_nt0) * g.split_k);
    }
    else
    {
        g.tail_main = 0;
        g.part_col0 = 0;
        g.part_ntot = g.Ntot;
    }
    hipLaunchKernelGGL((gemm_mfma_kernel<Shape, ConvGemmPolicy<MODE, TWIN>>), dim3(grid), dim3(Shape::THREADS), 0, s, g);
}

// force_no_act: the NAIVE algo ignores activation (avx/booster.cpp:41-61).
int igemm_forward(const fhip_conv_param& p, int batch, float* out, const float* in, const float* packed, const float* bias,
                  float* buffer, bool force_no_act, hipStream_t s, const float* residual)
{
    if (p.group > 1) return fail(FHIP_E_UNSUPPORTED, "implicit GEMM handles group == 1 only");
    if (batch < 1) return fail(FHIP_E_BADARG, "batch < 1");
    if (p.bias_term && !bias) return fail(FHIP_E_BADARG, "bias_term set but bias_arr is NULL");
    ConvGemmParams g;
    g.batches = 1;
    g.Wt = packed;
    g.in = in;
    g.out = out;
    g.out2 = nullptr;
    g.twin_rows = g.relu2 = 0;
    g.bias = bias;
    g.has_residual = residual != nullptr;
    g.residual_delta = residual ? reinterpret_cast<const char*>(residual) - reinterpret_cast<const char*>(out) : 0;
    g.C = p.input_channels;
    g.K = p.output_channels;
    g.H = p.input_h;
    g.W = p.input_w;
    g.OH = p.output_h;
    g.OW = p.output_w;
    g.SH = p.stride_h > 0 ? p.stride_h : 1;
    g.SW = p.stride_w > 0 ? p.stride_w : 1;
    g.PL = p.pad_left;
    g.PT = p.pad_top;
    g.KH = p.kernel_h;
    g.KW = p.kernel_w;
    g.Kd = g.C * g.KH * g.KW;
    int kdp;
    igemm_packed_dims(p, &kdp, &g.Kp);
    g.Kdp = kdp;
    g.bm = conv_small_m(g.K) ? 64 : 128; // panel height the weights were packed with (igemm_init)
    g.OHW = g.OH * g.OW;
    g.HW = g.H * g.W;
    g.KHW = g.KH * g.KW;
    const long long ntot = (long long)batch * g.OHW;
    if (ntot > 0x7fffff00LL) return fail(FHIP_E_BADARG, "N*Ho*Wo exceeds 32-bit column indices");
    g.Ntot = (int)ntot;
    g.rag_gpi = 0;
    g.has_bias = p.bias_term != 0;
    g.relu = (p.activation == FHIP_ACT_RELU) && !force_no_act;
    if (!residual && smallc_applicable(p))
    {
        StageTimer tm(FHIP_STAGE_IGEMM, s);
        return smallc_forward(p, batch, out, in, packed, bias, g.relu, s);
    }
    if (!residual && stream_profitable(p, batch))
    {
        StreamGemmParams q;
        q.in = in;
        q.wp = packed + (size_t)kdp * g.Kp;
        q.out = out;
        q.bias = bias;
        q.C = g.C;
        q.K = g.K;
        q.HW = g.OHW;
        const bool ragged = (g.OHW % 4) != 0;
        q.gpi = (g.OHW + 3) / 4;
        q.total_px = ragged ? (long long)batch * q.gpi * 4 : ntot;
        q.mgroups = g.K / 32;
        q.px_tiles = (int)((q.total_px + 127) / 128);
        const unsigned blocks = (unsigned)q.px_tiles * (unsigned)((q.mgroups + 3) / 4);
        StageTimer tm(FHIP_STAGE_IGEMM, s);
#define FHIP_STREAM(D_, B_, R_)                                                                                     \
    do                                                                                                              \
    {                                                                                                               \
        if (ragged) hipLaunchKernelGGL((stream_gemm_kernel<D_, B_, R_, true>), dim3(blocks), dim3(256), 0, s, q);   \
        else hipLaunchKernelGGL((stream_gemm_kernel<D_, B_, R_, false>), dim3(blocks), dim3(256), 0, s, q);         \
    } while (0)
        const int pick = ((g.C % 32) == 0 ? 4 : 0) | (g.has_bias ? 2 : 0) | (g.relu ? 1 : 0);
        switch (pick)
        {
            case 7: FHIP_STREAM(16, true, true); break;
            case 6: FHIP_STREAM(16, true, false); break;
            case 5: FHIP_STREAM(16, false, true); break;
            case 4: FHIP_STREAM(16, false, false); break;
            case 3: FHIP_STREAM(8, true, true); break;
            case 2: FHIP_STREAM(8, true, false); break;
            case 1: FHIP_STREAM(8, false, true); break;
            default: FHIP_STREAM(8, false, false); break;
        }
#undef FHIP_STREAM
        FHIP_CHECK_HIP(hipGetLastError());
        return FHIP_OK;
    }
    if (!residual && ip_profitable(p, batch))
    {
        if (!buffer) return fail(FHIP_E_BADARG, "this geometry streams its weight matrix in pieces and needs the scratch buffer GetBufferSize asked for");
        IpStreamParams q;
        q.wp = packed + (size_t)kdp * g.Kp + (stream_eligible(p) ? (size_t)g.K * g.C : 0);
        q.xq = buffer;
        q.partial = buffer + ip_xq_floats(p);
        q.K = g.K;
        q.Kg = ip_kg(p);
        q.KQ = ip_kq(p);
        q.S = ip_pieces(p);
        q.batch = batch;
        StageTimer tm(FHIP_STAGE_IGEMM, s);
        hipLaunchKernelGGL(ip_pack_input_kernel, dim3(ceil_div(q.KQ, kIpPackOctets)), dim3(256), 0, s, buffer, in, batch, g.C, q.KQ);
        hipLaunchKernelGGL(ip_stream_kernel<4>, dim3((unsigned)q.S * (unsigned)ceil_div(q.Kg, 4)), dim3(256), 0, s, q);
        hipLaunchKernelGGL(ip_reduce_kernel, dim3(ceil_div(g.K * batch, 256)), dim3(256), 0, s, out, q.partial, bias, g.K, batch, q.S, g.has_bias, g.relu);
        FHIP_CHECK_HIP(hipGetLastError());
        return FHIP_OK;
    }
    g.split_k = igemm_split(p, batch);
    g.tail_main = g.tail_nt0 = 0;
    if (g.split_k <= 1)
    {
        int pieces;
        g.tail_nt0 = igemm_tail(p, batch, &pieces);
        if (g.tail_nt0) g.split_k = pieces;
    }
    if (g.split_k > 1 && !buffer) return fail(FHIP_E_BADARG, "this geometry runs split-K and needs the scratch buffer GetBufferSize asked for");
    g.partial = buffer;
    g.k_tiles = kdp / kConvKTile; // in total; a split reduces its share (ConvGemmPolicy::k_first / k_count)
    g.m_tiles = 0;

    const bool one = g.KH == 1 && g.KW == 1 && p.pad_left == 0 && p.pad_right == 0 && p.pad_top == 0 && p.pad_bottom == 0;
    int mode = 0;
    if (one) mode = (g.SH == 1 && g.SW == 1 && (g.OHW % 4) == 0) ? 2 : 1;
    if (ragged_1x1(p, batch))
    {
        // planes that are not a multiple of 4 pixels: the columns become pixel slots (ConvGemmPolicy<5>)
        mode = 5;
        g.rag_gpi = (g.OHW + 3) / 4;
        const long long slots = igemm_columns(p, batch);
        if (slots > 0x7fffff00LL) return fail(FHIP_E_BADARG, "N*Ho*Wo exceeds 32-bit column indices");
        g.Ntot = (int)slots;
    }
    const bool small = conv_small_m(g.K);
    StageTimer tm(FHIP_STAGE_IGEMM, s);
    if (conv_narrow_n(ntot))
    {
        if (mode == 2) launch<ConvShapeNarrow, 2>(g, s);
        else if (mode == 1) launch<ConvShapeNarrow, 1>(g, s);
        else launch<ConvShapeNarrow, 0>(g, s);
    }
    else if (small)
    {
        if (mode == 5) launch<ConvShapeSmallM, 5>(g, s);
        else if (mode == 2) launch<ConvShapeSmallM, 2>(g, s);
        else if (mode == 1) launch<ConvShapeSmallM, 1>(g, s);
        else launch<ConvShapeSmallM, 0>(g, s);
    }
    else
    {
        if (mode == 5) launch<ConvShapeBig, 5>(g, s);
        else if (mode == 2) launch<ConvShapeBig, 2>(g, s);
        else if (mode == 1) launch<ConvShapeBig, 1>(g, s);
        else launch<ConvShapeBig, 0>(g, s);
    }
    FHIP_CHECK_HIP(hipGetLastError());
    if (g.split_k > 1)
    {
        // (tail split: the columns from the first tail tile on; `partial` is [pieces][K][those columns, padded to whole tiles])
        const int col0 = g.tail_nt0 * 64, pnt = g.tail_nt0 ? (ceil_div(g.Ntot, 64) - g.tail_nt0) * 64 : g.Ntot;
        hipLaunchKernelGGL(igemm_splitk_reduce_kernel, dim3(ceil_div(g.Ntot - col0, 256), g.K), dim3(256), 0, s, out, g.partial, bias, residual, g.K,
                           g.Ntot, g.OHW, g.split_k, g.has_bias, g.relu, g.rag_gpi, col0, pnt);
        FHIP_CHECK_HIP(hipGetLastError());
    }
    return FHIP_OK;
}

// ---- two 1x1 convolutions of the same input as ONE GEMM (ResNet's projection shortcut + the first layer of the main branch) ----------
// Both read the same (strided) pixels of the same blob: their filter matrices are stacked ([Ka + Kb][C], packed by igemm_init of the
// concatenated geometry) and gemm_mfma_kernel<ConvShapeBig, ConvGemmPolicy<MODE, TWIN>> writes rows < Ka to the first output and the
// rest to the second, each with its own activation.  One launch instead of two, and the short grid of the main-branch layer (K = 64 ...
// 512: 1 - 4 row tiles) rides in the long one of the shortcut instead of running the chip at 3 blocks per CU.
fhip_conv_param igemm_twin_geometry(const fhip_conv_param& a, const fhip_conv_param& b)
{
    fhip_conv_param c = a;
    c.output_channels = a.output_channels + b.output_channels;
    c.bias_term = (a.bias_term || b.bias_term) ? 1 : 0;
    c.activation = FHIP_ACT_NONE;
    return c;
}

bool igemm_twin_applicable(const fhip_conv_param& a, const fhip_conv_param& b, int batch)
{
    auto plain1x1 = [](const fhip_conv_param& c) {
        return c.group == 1 && c.kernel_h == 1 && c.kernel_w == 1 && !c.pad_left && !c.pad_right && !c.pad_top && !c.pad_bottom &&
               (c.activation == FHIP_ACT_NONE || c.activation == FHIP_ACT_RELU);
    };
    if (!plain1x1(a) || !plain1x1(b) || batch < 1) return false;
    if (a.input_channels != b.input_channels || a.input_h != b.input_h || a.input_w != b.input_w || a.stride_h != b.stride_h || a.stride_w != b.stride_w ||
        a.output_h != b.output_h || a.output_w != b.output_w)
        return false;
    if (a.output_channels % ConvShapeBig::BM) return false; // a block writes ONE of the two tensors
    const fhip_conv_param c = igemm_twin_geometry(a, b);
    const long long ntot = (long long)batch * c.output_h * c.output_w;
    if (ntot > 0x7fffff00LL || conv_narrow_n(ntot) || conv_small_m(c.output_channels) || smallc_applicable(c)) return false;
    return !stream_profitable(c, batch) && !ip_profitable(c, batch) && igemm_split(c, batch) == 1;
}

int igemm_twin_forward(const fhip_conv_param& a, const fhip_conv_param& b, int batch, float* out_a, float* out_b, const float* in, const float* packed,
                       const float* bias, hipStream_t s)
{
    if (!igemm_twin_applicable(a, b, batch)) return fail(FHIP_E_UNSUPPORTED, "these two layers cannot run as one GEMM (fhip_conv_can_fuse_siblings)");
    const fhip_conv_param p = igemm_twin_geometry(a, b);
    if (p.bias_term && !bias) return fail(FHIP_E_BADARG, "one of the layers has a bias but the concatenated bias is NULL");
    ConvGemmParams g;
    g.batches = 1;
    g.Wt = packed;
    g.in = in;
    g.out = out_a;
    g.out2 = out_b;
    g.twin_rows = a.output_channels;
    g.bias = bias;
    g.has_residual = 0;
    g.residual_delta = 0;
    g.C = p.input_channels;
    g.K = p.output_channels;
    g.H = p.input_h;
    g.W = p.input_w;
    g.OH = p.output_h;
    g.OW = p.output_w;
    g.SH = p.stride_h > 0 ? p.stride_h : 1;
    g.SW = p.stride_w > 0 ? p.stride_w : 1;
    g.PL = g.PT = 0;
    g.KH = g.KW = 1;
    g.Kd = g.C;
    int kdp;
    igemm_packed_dims(p, &kdp, &g.Kp);
    g.Kdp = kdp;
    g.bm = 128;
    g.OHW = g.OH * g.OW;
    g.HW = g.H * g.W;
    g.KHW = 1;
    g.Ntot = batch * g.OHW;
    g.has_bias = p.bias_term != 0;
    g.relu = a.activation == FHIP_ACT_RELU;
    g.relu2 = b.activation == FHIP_ACT_RELU;
    g.split_k = 1;
    g.partial = nullptr;
    g.tail_main = g.tail_nt0 = 0;
    g.k_tiles = kdp / kConvKTile;
    g.m_tiles = 0;
    g.dw_w12 = g.dw_bias = nullptr;
    g.dw_stride = g.dw_relu = 0;
    g.rag_gpi = 0;
    StageTimer tm(FHIP_STAGE_IGEMM, s);
    if (g.SH == 1 && g.SW == 1 && (g.OHW % 4) == 0) launch<ConvShapeBig, 2, true>(g, s);
    else launch<ConvShapeBig, 1, true>(g, s);
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

// ---- depthwise 3x3 fused into the 1x1 convolution that consumes it (MobileNet's dw -> pw pairs) -------------------------------
// out = act_pw(W_pw * act_dw(dw3x3(in) + b_dw) + b_pw): ConvGemmPolicy<3> computes the pointwise GEMM's B operand from the depthwise
// layer's INPUT, so the depthwise output (a tensor as large as the pair's input) is never written nor read.
// ---- the band-staged, wave-specialised form (dwpw_band.h) for MobileNet's first pair: a 3x3 / stride-1 depthwise layer with 32 channels on
// 112-pixel rows + a 1x1 layer with a multiple of 64 output channels.  The row width, the stride and the channel count are template parameters
// (constant divisions, fully unrolled chunk pipeline); the other pair geometries stay on ConvGemmPolicy<3|4>, which measured faster there.
constexpr int kDwPwBandMaxK = 128; // measured at 64 (MobileNet-V1's first pair); beyond two channel blocks the redundant depthwise work outgrows the saving
using Band112s1c32 = DwPwBandShape<112, 1, 2, 32, 8, 2, 4, 2>; // W, S, R, C, CH, consumer waves per pixel group, producer waves, blocks per CU
static bool dwpw_band_applicable(const fhip_conv_param& dw, const fhip_conv_param& pw)
{
    const int s = dw.stride_h > 0 ? dw.stride_h : 1;
    return dw.group == dw.input_channels && dw.kernel_h == 3 && dw.kernel_w == 3 && s == 1 && dw.stride_w == dw.stride_h && dw.pad_left == 1 &&
           dw.pad_top == 1 && dw.input_w == 112 && dw.output_w == 112 && dw.input_channels == 32 && pw.output_channels % 64 == 0 &&
           pw.output_channels <= kDwPwBandMaxK && // a work item covers 64 output channels and redoes the band's depthwise pass per channel block

           (dw.activation == FHIP_ACT_NONE || dw.activation == FHIP_ACT_RELU) && (pw.activation == FHIP_ACT_NONE || pw.activation == FHIP_ACT_RELU);
}

template <class SH>
static int dwpw_band_launch(const DwPwBandParams& q, int batch, hipStream_t s)
{
    constexpr size_t lds = (size_t)SH::LDS_FLOATS * sizeof(float);
    static_assert(lds <= 64 * 1024, "within the default dynamic-LDS limit: no hipFuncSetAttribute (no per-device state, nothing on the launch path a graph capture could trip over)");
    DwPwBandParams qq = q;
    qq.m_tiles = q.K / (32 * SH::CW);
    const long long items = (long long)batch * q.groups * qq.m_tiles;
    if (items > 0x7fffffffLL) return fail(FHIP_E_BADARG, "N * row groups too large");
    qq.bands = (int)items;
    // persistent: SH::BPC blocks per CU, each with a contiguous share of the (image, row group, channel block) items
    const int grid = (int)std::min<long long>(items, (long long)device_compute_units() * SH::BPC);
    hipLaunchKernelGGL((dwpw_band_kernel<SH>), dim3((unsigned)grid), dim3(SH::THREADS), lds, s, qq);
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

bool dwpw_applicable(const fhip_conv_param& dw, const fhip_conv_param& pw, int batch)
{
    const int s = dw.stride_h > 0 ? dw.stride_h : 1;
    if (dw.group != dw.input_channels || dw.group < 1 || dw.kernel_h != 3 || dw.kernel_w != 3 || dw.stride_w != dw.stride_h || (s != 1 && s != 2)) return false;
    if (dw.pad_left != 1 || dw.pad_top != 1 || dw.input_channels > kDwFusedMaxC) return false;
    if ((dw.input_w % 4) || (dw.output_w % 4) || dw.output_w * s != dw.input_w) return false; // aligned patch loads, no ragged right edge
    if (pw.group != 1 || pw.kernel_h != 1 || pw.kernel_w != 1 || (pw.stride_h > 1) || (pw.stride_w > 1) || pw.pad_left || pw.pad_right || pw.pad_top ||
        pw.pad_bottom)
        return false;
    if (pw.input_channels != dw.input_channels || pw.input_h != dw.output_h || pw.input_w != dw.output_w) return false;
    if (pw.output_h != dw.output_h || pw.output_w != dw.output_w) return false;
    // Worth it?  The fused kernel costs ~1.4x the pointwise GEMM alone (the 36 FMAs per operand vector issue next to the MFMAs), the
    // two-kernel form costs the GEMM plus an HBM-bound depthwise pass of 4*C*(HWin + HWout)*N bytes: with the GEMM at ~60 % of the
    // fp32 MFMA peak and HBM at ~5 TB/s the fused form wins while (HWin / HWout + 1) * 38 / K > 0.45, i.e. K < 160 behind a stride-1
    // and K < 400 behind a stride-2 depthwise layer.  Measured on MobileNet-V1 b256: C64->K128 s2 0.374 -> 0.310 ms,
    // C128->K128 s1 0.455 -> 0.369, C128->K256 s2 0.250 -> 0.238; C256->K256 s1 0.358 -> 0.337 and C32->K64 (64-row tile, two operand
    // requests per thread) 0.414 -> 0.404 are inside the noise and stay two kernels.
    if (!dwpw_band_applicable(dw, pw) && (pw.output_channels <= 64 || pw.output_channels >= (s == 1 ? 160 : 400))) return false;
    const long long ntot = (long long)batch * pw.output_h * pw.output_w;
    return batch >= 1 && ntot <= 0x7fffff00LL && !conv_narrow_n(ntot) && igemm_split(pw, batch) == 1;
}

int dwpw_forward(const fhip_conv_param& dw, const fhip_conv_param& pw, int batch, float* out, const float* in, const float* dw_packed,
                 const float* dw_bias, const float* pw_packed, const float* pw_bias, hipStream_t s)
{
    if (!dwpw_applicable(dw, pw, batch)) return fail(FHIP_E_UNSUPPORTED, "this depthwise + pointwise pair cannot be fused (fhip_conv_can_fuse_dw_pw)");
    if (dw.bias_term && !dw_bias) return fail(FHIP_E_BADARG, "depthwise bias_term set but its bias is NULL");
    if (pw.bias_term && !pw_bias) return fail(FHIP_E_BADARG, "pointwise bias_term set but its bias is NULL");
    ConvGemmParams g;
    memset(&g, 0, sizeof g);
    g.batches = 1;
    g.Wt = pw_packed;
    g.in = in;
    g.out = out;
    g.out2 = nullptr;
    g.rag_gpi = 0;
    g.twin_rows = g.relu2 = 0;
    g.bias = pw_bias;
    g.C = pw.input_channels;
    g.K = pw.output_channels;
    g.H = dw.input_h;
    g.W = dw.input_w;
    g.OH = pw.output_h;
    g.OW = pw.output_w;
    g.SH = g.SW = 1;
    g.KH = g.KW = 1;
    g.Kd = g.C;
    int kdp;
    igemm_packed_dims(pw, &kdp, &g.Kp);
    g.Kdp = kdp;
    g.bm = conv_small_m(g.K) ? 64 : 128;
    g.OHW = g.OH * g.OW;
    g.HW = g.H * g.W;
    g.KHW = 1;
    g.Ntot = (int)((long long)batch * g.OHW);
    g.has_bias = pw.bias_term != 0;
    g.relu = pw.activation == FHIP_ACT_RELU;
    g.split_k = 1;
    g.tail_main = g.tail_nt0 = 0;
    g.k_tiles = kdp / kConvKTile;
    size_t w12 = 0;
    depthwise_packed_floats(dw, &w12);
    g.dw_w12 = dw_packed + w12;
    g.dw_bias = dw.bias_term ? dw_bias : nullptr;
    g.dw_stride = dw.stride_h > 0 ? dw.stride_h : 1;
    g.dw_relu = dw.activation == FHIP_ACT_RELU;
    StageTimer tm(FHIP_STAGE_IGEMM, s);
    if (dwpw_band_applicable(dw, pw))
    {
        DwPwBandParams q;
        q.in = in;
        q.dw_w12 = g.dw_w12;
        q.dw_bias = g.dw_bias;
        q.wp = pw_packed + (size_t)kdp * g.Kp; // the streamed kernel's A-operand image behind the tiled kernel's panels (igemm_init)
        q.pw_bias = pw.bias_term ? pw_bias : nullptr;
        q.out = out;
        q.N = batch;
        q.K = g.K;
        q.H = g.H;
        q.OH = g.OH;
        q.dw_relu = g.dw_relu;
        q.pw_relu = g.relu;
        q.groups = ceil_div(g.OH, Band112s1c32::R);
        return dwpw_band_launch<Band112s1c32>(q, batch, s);
    }
    // three blocks per CU (168 VGPRs): the in-flight depthwise patches take 18 / 27 registers per operand request
    using FusedBig = GemmShape<128, 64, 16, 2, 2, 3>;
    if (g.dw_stride == 1) launch<FusedBig, 3>(g, s); // K > 64 (dwpw_applicable): always the 128-row tile
    else launch<FusedBig, 4>(g, s);
    FHIP_CHECK_HIP(hipGetLastError());
    return FHIP_OK;
}

} // namespace fhip
