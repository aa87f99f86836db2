/* feather_hip.h -- C-ABI of the MI355X (gfx950) convolution hot path behind FeatherCNN's booster API.
 *
 * This is the drop-in boundary: plain C, plain pointers and sizes, no torch / no C++
 * types.  All tensor pointers are DEVICE pointers (fp32, dense NCHW); `stream` is a hipStream_t passed
 * as void* (NULL = the default stream).  Every entry point returns 0 on success or a negative code
 * (FHIP_E_*); nothing throws across this ABI and nothing allocates device memory after Init -- the
 * caller owns every buffer, exactly as "ConvBooster doesn't allocate any memory"
 * (reference src/booster/include/booster/booster.h:155).
 *
 * Each declaration cites the reference interface it replaces (paths relative to /root/reference/src/booster).
 * Differences from the reference contract, all forced by the GPU batch path:
 *   - an explicit `batch` (the reference is N=1: src/layers/conv_layer.h:107);
 *   - sizes are BYTES in size_t (the reference reports float counts in int, booster.h:151);
 *   - `num_threads` is gone; a stream takes its place.
 *
 * Alignment.  Tensor pointers need 4-byte alignment, nothing more: the kernels that move whole planes as aligned 16-byte vectors check the
 * pointer and fall back to their dword forms (pooling fast paths, the LDS-staged Winograd input transform), and the ones that issue 16-byte
 * accesses at 4-byte-aligned addresses BY DESIGN (1x1 convolutions on planes that are not a multiple of 4 pixels, the staged Winograd output
 * stores) rely on the unaligned-access mode gfx9 devices run in under ROCm (SH_MEM_CONFIG.ALIGNMENT_MODE = unaligned, the driver default).
 * 16-byte-aligned tensors -- what hipMalloc, torch and the Net runtime's blobs give (256 bytes) -- are what every fast path is tuned for.
 */
#ifndef FEATHER_HIP_H_
#define FEATHER_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FHIP_API __attribute__((visibility("default")))

/* booster::ConvAlgo, include/booster/booster.h:42-51 (same numeric values). */
enum fhip_conv_algo
{
    FHIP_NAIVE = 0,            /* golden im2col + plain GEMM + bias, IGNORES activation (avx/booster.cpp:28-61) */
    FHIP_IM2COL = 1,           /* implicit-GEMM conv on fp32 MFMA (replaces im2col + packed SGEMM)              */
    FHIP_SGECONV = 2,          /* unsupported, as on AVX (avx/booster.cpp:104-118 are empty stubs)              */
    FHIP_DEPTHWISE = 3,        /* LDS-staged depthwise                                                           */
    FHIP_WINOGRADF63 = 4,      /* Winograd F(6x6,3x3): input xform -> 64 tile GEMMs on MFMA -> output xform     */
    FHIP_WINOGRADF63FUSED = 5, /* unsupported (never selected by the reference, avx/booster.cpp:291-292)        */
    FHIP_WINOGRADF23 = 6       /* unsupported, as on AVX (avx/booster.cpp:162-176)                              */
};

/* booster::ActivationType, include/booster/booster.h:53-57 */
enum fhip_activation
{
    FHIP_ACT_NONE = 0,
    FHIP_ACT_RELU = 1
};

enum fhip_error
{
    FHIP_OK = 0,
    FHIP_E_UNSUPPORTED = -1, /* same code the reference returns for an unsupported algo / partial group
                                (avx/booster.cpp:304-308,348-354) */
    FHIP_E_BADARG = -2,
    FHIP_E_HIP = -3,         /* a HIP runtime call failed; see fhip_last_error() */
    FHIP_E_NODEVICE = -4
};

/* booster::ConvParam, include/booster/booster.h:59-77, field for field and in the same order.
 * bias_term is `bool` in the reference; it is an int here so the struct has one C layout. */
typedef struct fhip_conv_param
{
    int output_channels;
    int input_channels;
    int input_h;
    int input_w;
    int kernel_h;
    int kernel_w;
    int output_h;
    int output_w;
    int stride_h;
    int stride_w;
    int pad_left;
    int pad_bottom;
    int pad_right;
    int pad_top;
    int group;
    int bias_term;
    int activation; /* enum fhip_activation */
} fhip_conv_param;

/* ConvParam::AssignOutputDim, include/booster/booster.h:113-125: defaults group/stride 0 -> 1, floor output
 * dims, depthwise forces output_channels = input_channels. */
FHIP_API int fhip_conv_assign_output_dim(fhip_conv_param* param);

/* ConvParam::GetFLOPS, include/booster/booster.h:145-148 (per image). */
FHIP_API double fhip_conv_flops(const fhip_conv_param* param);

/* ConvBooster::SelectAlgo, avx/booster.cpp:283-310 (the AVX rule, incl. input_channels % 4 == 0 for Winograd).
 * Returns FHIP_E_UNSUPPORTED for partial groups, like the reference's -1. */
FHIP_API int fhip_conv_select_algo(const fhip_conv_param* param, int* algo);
/* The same candidates under the MI355X cost model (GPU-side addition, never the default): identical to
 * fhip_conv_select_algo except that the reference's `input_h > 8 && input_w > 8` guard on Winograd
 * (avx/booster.cpp:289, tuned for the CPU's cache blocking) is relaxed to `>= 4`: on this chip F(6x6,3x3) is
 * 1.7-2.2x faster than the implicit GEMM on 3x3 stride-1 layers of 4..8 pixels (ResNet-50's 7x7 stage: 0.245 ->
 * 0.118 ms at batch 64).  Limited to 16 <= input_channels <= 1024: F(6,3)'s fp32 error grows with the reduction length
 * (4e-5 normalised at C = 1024 against the reference's IM2COL result; the bar is 1e-4). */
FHIP_API int fhip_conv_select_algo_tuned(const fhip_conv_param* param, int* algo);

/* GET_BUFFER_SIZE_FUNC, include/booster/booster.h:151; per-algo bodies avx/booster.cpp:28-33,64-71,121-128,178-197.
 * Pure and cheap (callers invoke it on every Net::Forward, src/layers/conv_layer.h:105-112).
 * buffer_bytes: scratch for ONE in-flight forward of `batch` images; processed_kernel_bytes: packed weights. */
FHIP_API int fhip_conv_get_buffer_size(const fhip_conv_param* param, int algo, int batch, size_t* buffer_bytes,
                                       size_t* processed_kernel_bytes);

/* INIT_FUNC, include/booster/booster.h:152; bodies avx/booster.cpp:35-39,73-81,130-134,199-203.
 * One-time weight pre-processing on the device: Winograd U = G g G^T (transformKernel_F6x6_3x3,
 * avx/winograd_kernels_F63.cpp:256-271), GEMM weight packing (packed_sgemm_init, avx/sgemm.cpp:312-346),
 * depthwise copy.  kernel layout [K][C/group][kh][kw] (src/layers/conv_layer.h:81). Idempotent. */
FHIP_API int fhip_conv_init(const fhip_conv_param* param, int algo, float* processed_kernel, const float* kernel,
                            void* stream);

/* FORWARD_FUNC, include/booster/booster.h:153; bodies avx/booster.cpp:41-61,83-102,136-160,205-230.
 * input [batch][C][H][W] -> output [batch][K][Ho][Wo]; bias_arr [K] (may be NULL when !bias_term);
 * fused bias / ReLU epilogue selected by param->bias_term / param->activation.  Asynchronous on `stream`;
 * never allocates.  One in-flight forward per `buffer`. */
FHIP_API int fhip_conv_forward(const fhip_conv_param* param, int algo, int batch, float* output, const float* input,
                               const float* processed_kernel, float* buffer, const float* bias_arr, void* stream);

/* ---- stage-level entry points (what the reference exposes as free functions) ----------------------
 * They exist so each kernel can be tested and roofline-timed on its own, like booster's
 * winograd_kernels.h / sgemm.h / depthwise.h free functions "facilitate unit testing" (booster.h:15-16). */

/* Geometry of the Winograd scratch: tiles per image, padded column count and byte offsets of V and M
 * inside `buffer` (the GPU analogue of the VT | WT carve in WINOGRADF63_Forward, avx/booster.cpp:214-217). */
typedef struct fhip_winograd_plan
{
    int tiles_x, tiles_y;   /* nRowBlocks, nColBlocks (avx/booster.cpp:209-210)          */
    int tiles_per_image;    /* T                                                        */
    int columns;            /* P = T * batch                                            */
    int columns_padded;     /* Pp: P rounded up to the GEMM column tile and to column_block */
    int column_block;       /* BP: V and M are stored in blocks of BP columns, [Pp / BP][64][rows][BP]; BP == Pp: whole rows */
    int frequency_points;   /* 64: F(6x6,3x3) on 8x8 input tiles; 36: F(4x4,3x3) on 6x6 input tiles -- planes of 7 or 8 output pixels per side,
                               where 2x2 tiles of 4x4 outputs waste far less than 2x2 tiles of 6x6 (round 4; "64" above then reads 36) */
    int tile_outputs;       /* 6 or 4: output pixels per tile side                      */
    int in_channels_padded; /* C rounded up to the GEMM reduction tile (U only)         */
    int out_channels_padded;/* K rounded up to the GEMM row tile (U only)               */
    size_t v_offset_bytes, v_bytes; /* V[Pp / BP][64][C][BP]                            */
    size_t m_offset_bytes, m_bytes; /* M[Pp / BP][64][K][BP]                            */
    size_t u_bytes;                 /* U[64][Cp][Kp] = processed kernel                 */
} fhip_winograd_plan;

FHIP_API int fhip_winograd_f63_plan(const fhip_conv_param* param, int batch, fhip_winograd_plan* plan);

/* transformKernel_F6x6_3x3, include/booster/winograd_kernels.h:33, avx/winograd_kernels_F63.cpp:256-271 */
FHIP_API int fhip_winograd_f63_transform_kernel(const fhip_conv_param* param, float* u, const float* kernel,
                                                void* stream);
/* pad_input + winogradInputFrameTransformSeq, avx/generic_kernels.cpp:31-48 + avx/winograd_kernels_F63.cpp:327-513 */
FHIP_API int fhip_winograd_f63_input_transform(const fhip_conv_param* param, int batch, float* v, const float* input,
                                               void* stream);
/* TensorGEMM, avx/winograd_kernels_F63.cpp:518-692: M_xi[K x P] = U_xi[K x C] * V_xi[C x P] for 64 xi */
FHIP_API int fhip_winograd_f63_tile_gemm(const fhip_conv_param* param, int batch, float* m, const float* u,
                                         const float* v, void* stream);
/* winogradOutputTransform<relu,bias>, avx/winograd_kernels_F63.cpp:1088-1269 */
FHIP_API int fhip_winograd_f63_output_transform(const fhip_conv_param* param, int batch, float* output, const float* m,
                                                const float* bias_arr, void* stream);

/* ---- introspection / measurement ------------------------------------------------------------------ */

/* Stage timing with HIP events recorded on the launch stream (off by default; adds two event records per
 * kernel).  Stages: 0 winograd input transform, 1 tile GEMM, 2 winograd output transform,
 * 3 implicit-GEMM conv, 4 depthwise, 5 weight transforms, 6 chained Winograd output -> input transform. */
enum fhip_stage
{
    FHIP_STAGE_WINO_INPUT = 0,
    FHIP_STAGE_WINO_GEMM = 1,
    FHIP_STAGE_WINO_OUTPUT = 2,
    FHIP_STAGE_IGEMM = 3,
    FHIP_STAGE_DEPTHWISE = 4,
    FHIP_STAGE_INIT = 5,
    FHIP_STAGE_WINO_CHAIN = 6, /* chained output -> next layer's input transform (feather_net.h) */
    FHIP_STAGE_COUNT = 7
};
FHIP_API int fhip_stage_timing_enable(int on);
/* Synchronises the recorded events, adds their durations to per-stage totals, returns totals (ms) and
 * launch counts, then clears them.  ms and launches must hold FHIP_STAGE_COUNT entries. */
FHIP_API int fhip_stage_timing_collect(double* ms, long long* launches);

FHIP_API const char* fhip_last_error(void);
FHIP_API const char* fhip_version(void);
/* Introspection: 1 when fhip_conv_forward runs this IM2COL / NAIVE layer at this batch through the register-streamed 1x1 GEMM
 * (stream_gemm.h: 1x1, stride 1, unpadded, C % 16 == 0, K % 32 == 0, Ho*Wo >= 4, C >= 256, 128 <= K <= 512, >= 4096 pixels in the
 * batch) instead of the LDS-tiled implicit GEMM.  Same result up to the summation order (tests/test_stream_gemm_gpu.py). */
FHIP_API int fhip_conv_streams_1x1(const fhip_conv_param* param, int algo, int batch);

/* name (>= 64 bytes), compute units, LDS bytes per CU; returns FHIP_E_NODEVICE if there is no GPU. */
FHIP_API int fhip_device_info(char* name, int name_len, int* compute_units, int* lds_bytes);
/* Calibration (measurement aid, ~10 ms): runs a kernel that is nothing but v_mfma_f32_32x32x2_f32 chains -- random U(-1,1) operands in
 * registers, four independent accumulators per wave, 3 waves per SIMD on every CU, i.e. the residency of the library's GEMM kernels --
 * and reports the best of 11 repetitions: *tflops = what the fp32 matrix pipe of THIS device sustains, *shader_mhz = shader-clock
 * ticks per 100 MHz wall-clock tick inside the kernel (s_memtime / s_memrealtime).  The nominal peak (157.3 TFLOP/s on MI355X) assumes
 * 2.4 GHz; under full-chip fp32 MFMA load the part clocks to its power budget.  Measured on MI355X: 126-147 TFLOP/s with random operands, depending on the box and its thermal state
 * (149-156 with constant ones -- dynamic power follows operand toggling), so that, not the nominal figure, is what a perfect
 * MFMA-bound kernel could reach on real data.  bench.py reports both. */
FHIP_API int fhip_calibrate_mfma_f32(double* tflops, double* shader_mhz, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FEATHER_HIP_H_ */
