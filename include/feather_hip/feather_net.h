/* feather_net.h -- C-ABI of the layers between the convolutions and of the feather::Net-compatible
 * runtime on device blobs (the callers and data formats either side of the
 * ConvBooster hot path).  Same conventions as feather_hip.h: plain C, DEVICE pointers (fp32 NCHW dense),
 * `stream` is a hipStream_t as void*, every call returns 0 or a negative fhip_error.
 *
 * Paths cited are relative to /root/reference/src.
 */
#ifndef FEATHER_NET_H_
#define FEATHER_NET_H_

#include "feather_hip/feather_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- layer kernels ---------------------------------------------------------------------------------- */

/* ReluLayer::Forward, layers/relu_layer.h:29-41: y = x > 0 ? x : 0 over `count` floats. */
FHIP_API int fhip_relu(float* y, const float* x, size_t count, void* stream);

/* EltwiseLayer::Forward (SUM only) -> booster::add_relu<fuse_relu>, layers/eltwise_layer.h:69-80,
 * booster/avx/generic_kernels.cpp:138: y = a + b, then max(0, .) when relu != 0. */
FHIP_API int fhip_add(float* y, const float* a, const float* b, size_t count, int relu, void* stream);

/* booster::scale<bias> (generic_kernels.cpp:203-233) and booster::batchnorm<bias,scale,relu>
 * (generic_kernels.cpp:237-279) are both per-channel affine maps: y[n][c][i] = x[n][c][i]*mul[c] + add[c],
 * optionally followed by ReLU.  mul/add are device vectors of `channels` floats; add may be NULL. */
FHIP_API int fhip_affine(float* y, const float* x, const float* mul, const float* add, int batch, int channels, int hw,
                         int relu, void* stream);

/* Convolution (+bias, +ReLU as the param says) followed by a 2x2 / stride-2 / unpadded MAX pooling, fused: the pooled
 * tensor [N][K][OH/2][OW/2] is written straight from the Winograd output transform and the full-resolution activation
 * never reaches HBM (VGG: every pooling layer follows a 3x3 convolution).  Same arguments as fhip_conv_forward.  Only the
 * WINOGRADF63 route with even output dims can do it: fhip_conv_can_fuse_maxpool2 returns 1 when it can, and
 * fhip_conv_forward_maxpool2 returns FHIP_E_UNSUPPORTED otherwise (run fhip_conv_forward + fhip_pooling instead).
 * Equal to ConvLayer::Forward + PoolingLayer::Forward of the reference (conv_layer.h:141-150, pooling_layer.h:37-88). */
FHIP_API int fhip_conv_can_fuse_maxpool2(const fhip_conv_param* param, int algo);
FHIP_API int fhip_conv_forward_maxpool2(const fhip_conv_param* param, int algo, int batch, float* pooled_output, const float* input,
                                        const float* packed, float* buffer, const float* bias, void* stream);

/* Convolution followed by an Eltwise SUM with `residual` (same shape as the output) and the activation the param names:
 * output = act(conv(input) + bias + residual), the add done in the GEMM epilogue (ResNet's last 1x1 of a bottleneck +
 * shortcut + ReLU; reference: ConvLayer::Forward, EltwiseLayer::Forward -> booster::add_relu, eltwise_layer.h:69-80).  Same
 * arguments as fhip_conv_forward plus `residual`.  IM2COL route only (fhip_conv_can_fuse_residual), else FHIP_E_UNSUPPORTED. */
FHIP_API int fhip_conv_can_fuse_residual(const fhip_conv_param* param, int algo);
FHIP_API int fhip_conv_forward_residual(const fhip_conv_param* param, int algo, int batch, float* output, const float* input,
                                        const float* packed, float* buffer, const float* bias, const float* residual, void* stream);

/* A 3x3 depthwise convolution (+bias, +ReLU as `dw` says) followed by the 1x1 convolution that is its only consumer (+bias, +ReLU as
 * `pw` says), fused: output = act_pw(W_pw * act_dw(dw(input) + b_dw) + b_pw).  The pointwise GEMM computes its column operand from
 * the depthwise layer's INPUT on the fly, so the depthwise output -- a tensor as large as the pair's input -- never reaches HBM
 * (MobileNet-V1's first dw/pw pairs are HBM bound).  Equal to two ConvLayer::Forward calls of the reference (conv_layer.h:141-150
 * with DEPTHWISE_Forward, avx/booster.cpp:136-160, then IM2COL_Forward, :83-102).
 *   dw_packed / pw_packed: what fhip_conv_init produced for FHIP_DEPTHWISE / FHIP_IM2COL; no scratch buffer is needed.
 * fhip_conv_can_fuse_dw_pw returns 1 when the pair qualifies: depthwise 3x3, stride 1 or 2, pad_left = pad_top = 1, input and
 * output widths multiples of 4 with output_w * stride == input_w, at most 256 channels; pointwise 1x1 stride 1 unpadded on exactly
 * the depthwise output, large enough not to run split-K at this batch, and with 64 < output_channels < 160 (stride 1) / 400 (stride 2)
 * -- the range in which one kernel beats two on this chip.  Otherwise fhip_conv_forward_dw_pw returns FHIP_E_UNSUPPORTED (run the two
 * layers one after the other). */
FHIP_API int fhip_conv_can_fuse_dw_pw(const fhip_conv_param* dw, const fhip_conv_param* pw, int batch);
FHIP_API int fhip_conv_forward_dw_pw(const fhip_conv_param* dw, const fhip_conv_param* pw, int batch, float* output, const float* input,
                                     const float* dw_packed, const float* dw_bias, const float* pw_packed, const float* pw_bias, void* stream);

/* PoolingLayer, layers/pooling_layer.h:90-131 (fields as its LoadParam reads them). */
typedef struct fhip_pool_param
{
    int channels, input_h, input_w;
    int kernel_h, kernel_w;
    int stride_h, stride_w;
    int pad_left, pad_right, pad_top, pad_bottom;
    int pooling_type;   /* 0 = max, anything else = average (pooling_layer.h:75) */
    int global_pooling; /* kernel = whole image, output 1x1 (pooling_layer.h:116-123) */
} fhip_pool_param;

/* PoolingLayer::Reshape, pooling_layer.h:116-131: ceil((in + pads - k) / stride) + 1, or 1x1 when global. */
FHIP_API int fhip_pooling_output_dim(const fhip_pool_param* p, int* out_h, int* out_w);
/* PoolingLayer::Forward, pooling_layer.h:37-88.  Reference semantics kept exactly: the window origin is
 * j*stride - pad_top - pad_bottom (both pads, :56,:67), windows are clipped to the image, the average
 * divides by the number of in-range taps, an empty window yields -FLT_MAX (max) or NaN (average). */
FHIP_API int fhip_pooling(const fhip_pool_param* p, int batch, float* y, const float* x, void* stream);

/* SoftmaxLayer::Forward, layers/softmax_layer.h:33-53: max-subtracted exp / sum over all
 * `count_per_image` values of each image. */
FHIP_API int fhip_softmax(float* y, const float* x, int batch, int count_per_image, void* stream);

/* Consecutive Winograd layers without the activation between them.  When layer `p` (WINOGRADF63: 3x3, stride 1) is followed -- directly,
 * or behind a 2x2 / stride-2 max pooling when pool = 1 -- by a 3x3 / stride-1 / pad-1 WINOGRADF63 layer `next` that is its only consumer,
 * p's output transform can write next's TRANSFORMED INPUT V' (fhip_winograd_plan(next): [64][C'][P'_pad]) instead of the activation:
 * one kernel does Y = A^T m A, + bias, ReLU [, pooling] into LDS (one whole (image, channel) plane per block, zero border = next's
 * padding) and V' = B^T d B out of it, with the butterflies and the fp32 operation order of the two separate transforms, so V' is
 * bit-identical.  The activation -- one HBM write and one read per layer boundary -- never exists (VGG-16: 11 of its 12 boundaries
 * between 3x3 layers).  Equal to ConvLayer::Forward [+ PoolingLayer::Forward] of the reference followed by the input-transform step of
 * the next ConvLayer::Forward (conv_layer.h:141-150; sgemm/winograd split: avx/booster.cpp:163-217, winograd_kernels_F63.cpp:64-515).
 *   fhip_conv_can_chain_winograd: 1 when the pair qualifies (both WINOGRADF63, next 3x3/s1/p1 on exactly p's [pooled] output, the
 *     padded plane fits a block's LDS: up to 112x112 activations).
 *   fhip_conv_forward_chained runs ONE layer of such a run:  input != NULL -> first layer: transform `input` into `v` first;
 *     input == NULL -> `v` already holds this layer's V (the previous call wrote it as its v_next).  next != NULL -> write next's V into
 *     `v_next` (a different buffer than `v`; `output` is ignored); next == NULL -> last layer: write `output` (pooled when pool = 1,
 *     as fhip_conv_forward_maxpool2).  `v` / `m` / `v_next` are caller-owned scratch of fhip_winograd_plan's v_bytes / m_bytes.
 *   fhip_winograd_f63_output_to_next_input is the chained transform alone (stage-level tests and profiling). */
FHIP_API int fhip_conv_can_chain_winograd(const fhip_conv_param* param, int algo, const fhip_conv_param* next, int next_algo, int pool);
FHIP_API int fhip_conv_forward_chained(const fhip_conv_param* param, int batch, float* output, const float* input, const float* packed, float* v,
                                       float* m, const float* bias, const fhip_conv_param* next, float* v_next, int pool, void* stream);
FHIP_API int fhip_winograd_f63_output_to_next_input(const fhip_conv_param* param, const fhip_conv_param* next, int batch, float* v_next,
                                                    const float* m, const float* bias, int pool, void* stream);

/* Two 1x1 convolutions of the SAME input as one GEMM: ResNet's projection shortcut (res3a_branch1, 256 -> 512, stride 2) and the first
 * layer of the main branch beside it (res3a_branch2a, 256 -> 128, stride 2) read the same pixels of the same blob; in the reference they are
 * two ConvLayer::Forward calls (conv_layer.h:141-150; packed SGEMM avx/sgemm.cpp:377-433).  Here their filter matrices are stacked
 * ([Ka + Kb][C]) and one launch writes rows < Ka to `output_a` and the rest to `output_b`, each with its own activation: the short grid of
 * the main-branch layer (1 - 4 row tiles) rides in the long one of the shortcut instead of running the chip at 3 blocks per CU.
 *   fhip_conv_can_fuse_siblings: 1 when the pair qualifies at this batch (both IM2COL 1x1 / pad 0 / group 1 on the same input geometry and
 *     stride, Ka a multiple of 128, the combined grid needs no split-K and takes the LDS-tiled route).
 *   fhip_conv_siblings_geometry fills the geometry of the stacked layer: run fhip_conv_get_buffer_size / fhip_conv_init (algo IM2COL) on
 *     it with the stacked filters to get `packed_both`; `bias_both` = the two biases one after the other (zeros for a layer without one).
 *   fhip_conv_forward_siblings: the launch.  Results equal the two layers run on the LDS-tiled route (same tile, same k order). */
FHIP_API int fhip_conv_can_fuse_siblings(const fhip_conv_param* a, int algo_a, const fhip_conv_param* b, int algo_b, int batch);
FHIP_API int fhip_conv_siblings_geometry(const fhip_conv_param* a, const fhip_conv_param* b, fhip_conv_param* both);
FHIP_API int fhip_conv_forward_siblings(const fhip_conv_param* a, const fhip_conv_param* b, int batch, float* output_a, float* output_b,
                                        const float* input, const float* packed_both, const float* bias_both, void* stream);

/* A net's first convolution inside the Winograd layer behind it.  `first` is a 3x3 / stride-1 / pad-1 convolution with 2 .. 4 input
 * channels (VGG-16's conv1_1: ConvLayer::Forward through the im2col + SGEMM route, conv_layer.h:141-150, avx/booster.cpp:108-160) whose
 * only consumer `next` is a 3x3 / stride-1 / pad-1 WINOGRADF63 layer.  Its output is as large as the largest tensor of the net and holds
 * 27 multiply-adds per value, so next's input transform can compute every activation value of its 8x8 windows from the image instead
 * of loading it: V(next) = B^T act(first(image)) B with the consumer's zero padding outside the image; the first layer's output never
 * exists in memory.  `first_kernel` are first's filters as loaded ([K][C][3][3], device), `first_bias` its bias or NULL; the image
 * must be under 1 GiB and its width even.
 *   fhip_conv_can_fuse_first_winograd: 1 when the pair qualifies at this batch.
 *   fhip_winograd_f63_input_from_first writes next's V (fhip_winograd_plan(next).v_bytes); run the rest of `next` with
 *     fhip_conv_forward_chained(next, ..., input = NULL, v = that buffer, ...). */
FHIP_API int fhip_conv_can_fuse_first_winograd(const fhip_conv_param* first, const fhip_conv_param* next, int next_algo, int batch);
FHIP_API int fhip_winograd_f63_input_from_first(const fhip_conv_param* first, const fhip_conv_param* next, int batch, float* v_next,
                                                const float* input, const float* first_kernel, const float* first_bias, void* stream);

/* ---- feather::Net on device blobs --------------------------------------------------------------------- */

/* Opaque handle to a feather::Net (include/feather/net.h; reference net.h:30-70). */
typedef struct fhip_net fhip_net;

FHIP_API int fhip_net_create(fhip_net** net);
FHIP_API int fhip_net_destroy(fhip_net* net);
/* All work of this net is enqueued on `stream` (default: the NULL stream).  Set before the first Forward. */
FHIP_API int fhip_net_set_stream(fhip_net* net, void* stream);
/* Fusion level.  Set before the first Forward.
 *   0: none -- every layer of the file runs and every blob can be extracted, like the reference as shipped.
 *   1 (default): the TryFuse pass the reference declares but never calls (layer.cpp:82-101): Conv+ReLU, InnerProduct+ReLU,
 *      BatchNorm+Scale(+ReLU), Scale+ReLU, Eltwise+ReLU.
 *   2: + BatchNorm / Scale folded into the convolution before them, Conv + 2x2 max pooling, Conv + Eltwise SUM (+ReLU), a 3x3
 *      depthwise layer + the 1x1 convolution behind it as one layer (fhip_conv_forward_dw_pw where the pair qualifies), and two 1x1
 *      convolutions of the same input that follow each other as one GEMM (fhip_conv_forward_siblings; both blobs keep their storage).
 *   3: + runs of Winograd layers chained (fhip_conv_forward_chained): the blob between two chained layers has a shape but no storage;
 *      a first layer (3x3 / stride 1 / pad 1, 2 .. 4 input channels) in front of a Winograd layer is computed inside that layer's input
 *      transform (fhip_winograd_f63_input_from_first): its top has no storage either.
 * A blob that a fusion removed cannot be extracted (fhip_net_extract fails and says which level to use). */
FHIP_API int fhip_net_set_fusion(fhip_net* net, int on);
/* 1: convolutions choose their route with fhip_conv_select_algo_tuned (MI355X cost model) instead of the reference's
 * SelectAlgo rule; 0 (default): the reference rule. */
FHIP_API int fhip_net_set_tuned_selection(fhip_net* net, int on);
/* 1: independent branches run concurrently: a convolution that needs no scratch arena and whose output is only consumed
 * further down the layer list (ResNet's projection shortcut, SqueezeNet's expand1x1) is enqueued on a second stream owned by
 * the net while the main stream continues; fork and join are events (hipGraph-capturable).  0 (default): one stream. */
FHIP_API int fhip_net_set_concurrency(fhip_net* net, int on);
/* 1: after the first Forward for a shape, record the layer sequence into a hipGraph and replay it.  Capture is not
 * allowed on the NULL stream: if no stream was set, the net creates and uses its own non-blocking stream (order device
 * inputs produced on other streams yourself). */
FHIP_API int fhip_net_set_graph(fhip_net* net, int on);

/* Sub-batch replicas (default 1 = off).  With R > 1 the handle owns R complete copies of the net (weights, blobs, arena, graph), each on a
 * stream of its own: FeedInput deals the images of a batch out in R contiguous shares (the first ones take the remainder), Forward
 * runs the replicas concurrently -- forked off and joined back into the net's stream with events, so the caller's stream order is
 * unchanged -- and Extract puts the shares back together ([N][C][H][W], images in feed order).  Images are independent, so results
 * equal the single-net ones up to the batch-dependent reduction order of split-K layers (<= 1e-6 normalised).  What it buys: kernels of
 * different character overlap (MobileNet-V1 b256: HBM-bound depthwise layers of one share under the MFMA-bound 1x1 layers of the
 * other, +8 % images/s with R = 2) and the tails of small launches are filled; nets made of long uniform launches gain nothing
 * (VGG-16, ResNet-50: +-1 %).  Set once, before LoadParam.  Introspection calls describe replica 0; fhip_net_forward_timed refuses a net
 * with replicas (kernels of concurrent replicas share the chip, their durations are not layer times). */
FHIP_API int fhip_net_set_sub_batches(fhip_net* net, int replicas);

/* Net::LoadParam, net.cpp:54-170 (ncnn text .param: magic 7767517, "layers blobs", one line per layer). */
FHIP_API int fhip_net_load_param(fhip_net* net, const char* path);
FHIP_API int fhip_net_load_param_mem(fhip_net* net, const char* text, size_t len);
/* Net::LoadWeights, net.cpp:172-233 (ncnn .bin read in layer order; ncnn/modelbin.cpp:47-197). */
FHIP_API int fhip_net_load_weights(fhip_net* net, const char* path);
FHIP_API int fhip_net_load_weights_mem(fhip_net* net, const void* data, size_t len);
/* The same image in DEVICE memory of the current device: what `ncclBroadcast` of the .bin from rank 0 leaves on every rank
 * (the one collective of the multi-GPU path; tests/cpp/multi_gpu_main.cpp, INTEGRATION.md "multi-GPU from C++").
 * Staged through host memory once (weights stay on the host until Init folds BatchNorm / Scale into them); `device_data` stays the
 * caller's and may be freed on return. */
FHIP_API int fhip_net_load_weights_device(fhip_net* net, const void* device_data, size_t len);

/* Net::FeedInput, net.cpp:235-246, extended with a batch.  `data` holds n*c*h*w floats; `on_device` says
 * whether it is a device pointer (copied device-to-device on the net's stream) or a host pointer. */
FHIP_API int fhip_net_feed_input(fhip_net* net, const char* blob_name, int n, int c, int h, int w, const float* data,
                                 int on_device);
/* Net::Forward, net.cpp:297-334: Reshape when the input shape changed, Init once, then every layer in file
 * order on the net's stream.  Asynchronous: returns after enqueueing. */
FHIP_API int fhip_net_forward(fhip_net* net);
/* Net::Extract(name, float**, n, c, h, w), net.cpp:263-279: DEVICE pointer into the net's blob + its shape.
 * The pointer stays valid until the next Reshape.  With sub-batch replicas the pointer is a per-name buffer the shares are gathered
 * into (on the net's stream); it is re-allocated only when that blob's total size grows, i.e. after a FeedInput with a larger shape. */
FHIP_API int fhip_net_extract(fhip_net* net, const char* blob_name, float** device_ptr, int* n, int* c, int* h, int* w);
/* Convenience: synchronise the stream and copy the blob to a host buffer of `capacity` floats. */
FHIP_API int fhip_net_extract_host(fhip_net* net, const char* blob_name, float* host, size_t capacity);

/* Introspection (after LoadParam / fusion). */
FHIP_API int fhip_net_layer_count(fhip_net* net);
/* type / name are copied (truncated) into caller buffers of `len` bytes; algo = fhip_conv_algo for
 * convolutions after the first Forward, else -1. */
FHIP_API int fhip_net_layer_info(fhip_net* net, int index, char* type, char* name, int len, int* algo);
/* Geometry of a Convolution / ConvolutionDepthWise layer as it runs (after Reshape: output dims assigned, BatchNorm folded, ...) and
 * the batch of its input blob; FHIP_E_BADARG for any other layer type.  (bench.py prices each layer's kernel against its roofline
 * with ConvParam::GetFLOPS, booster.h:145-148.) */
FHIP_API int fhip_net_layer_conv_param(fhip_net* net, int index, fhip_conv_param* param, int* batch);
/* Fusion level 2 makes a 3x3 depthwise layer (<= 256 channels, stride 1 / 2, pad 1) and the 1x1 convolution behind it ONE layer: for
 * such a layer, fhip_net_layer_conv_param describes the depthwise half and this returns the pointwise half; *one_kernel = 1 when the
 * pair runs as one kernel at the current shape (fhip_conv_forward_dw_pw), 0 when it runs the two kernels one after the other.
 * FHIP_E_BADARG for every other layer. */
FHIP_API int fhip_net_layer_fused_pointwise(fhip_net* net, int index, fhip_conv_param* param, int* one_kernel);
/* Fusion level 3: *v_from_previous = 1 when this convolution's transformed input is written by the layer before it (it runs no input
 * transform), *writes_next_v = 1 when its output leaves as the next layer's transformed input (fhip_conv_forward_chained).  The value
 * is 2 for the pair "first layer computed inside the next layer's input transform" (fhip_winograd_f63_input_from_first): writes_next_v =
 * 2 on the first layer (it launches nothing), v_from_previous = 2 on the Winograd layer behind it. */
FHIP_API int fhip_net_layer_chain(fhip_net* net, int index, int* v_from_previous, int* writes_next_v);
/* Fusion level 2: *state = 1 when this 1x1 convolution's launch also computes the NEXT layer of the list (a 1x1 convolution of the same
 * input: fhip_conv_forward_siblings), 2 when this layer is that next one (it launches nothing), 0 otherwise. */
FHIP_API int fhip_net_layer_sibling(fhip_net* net, int index, int* state);
/* Fusion level 2: *state = 1 when an Eltwise SUM behind this convolution was absorbed and its other operand is added in the GEMM epilogue
 * (fhip_conv_forward_residual: the operand is one more read of an output-sized tensor by this launch), 2 when it was absorbed but is added by a
 * launch of its own (fhip_add), 0 when the layer has no residual operand. */
FHIP_API int fhip_net_layer_residual(fhip_net* net, int index, int* state);
/* One eager forward with a pair of events around every layer; ms must hold layer_count entries. */
FHIP_API int fhip_net_forward_timed(fhip_net* net, float* ms);
/* Device bytes currently held: blobs, weights, scratch arena. */
FHIP_API int fhip_net_memory(fhip_net* net, size_t* blob_bytes, size_t* weight_bytes, size_t* arena_bytes);

#ifdef __cplusplus
}
#endif
#endif /* FEATHER_NET_H_ */
