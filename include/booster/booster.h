// booster/booster.h -- C++ host side of the MI355X conv hot path: the booster operator API
// (booster::ConvParam / booster::ConvBooster) with the same names, argument meaning and error behaviour
// as the reference header (reference src/booster/include/booster/booster.h:42-170), so that
// feather::ConvLayer (reference src/layers/conv_layer.h:92-172) compiles against it unchanged:
//
//     conv_param.AssignOutputDim();  conv_booster.SelectAlgo(&conv_param);
//     conv_booster.GetBufferSize(&conv_param, &buffer_size, &processed_kernel_size);
//     conv_booster.Init(&conv_param, processed_kernel, kernel);
//     conv_booster.Forward(&conv_param, output, input, processed_kernel, buffer, bias, num_threads);
//
// What changes is where the pointers live: output / input / kernel / processed_kernel / buffer / bias_arr are
// DEVICE pointers, and the three function pointers forward to the C-ABI in feather_hip/feather_hip.h.
// GPU-only additions (all optional, zero-initialised ConvParam behaves like the N=1 reference):
//   * ConvParam::batch       -- images per Forward (0 or 1 = the reference's single image);
//   * booster::SetStream()   -- thread-local HIP stream used by Init / Forward (default: the null stream);
//   * ConvBooster::GetBufferSizeBytes() -- size_t byte counts (the reference's int float counts overflow
//     for batched Winograd scratch; GetBufferSize returns -1 instead of a truncated number).
// `num_threads` is accepted and ignored.
#pragma once

#include <stddef.h>
#include <stdio.h>

namespace booster
{

// Same enumerators and numeric values as the reference (booster.h:42-57): they travel through the C-ABI as plain ints.
enum ConvAlgo { NAIVE, IM2COL, SGECONV, DEPTHWISE, WINOGRADF63, WINOGRADF63FUSED, WINOGRADF23 };
enum ActivationType { None, ReLU };

// Field for field and in the same order as the reference struct (booster.h:59-77), so host code that fills a ConvParam
// compiles and lays out unchanged; `batch` is appended.  The member functions keep the reference's names and meaning but
// live in libfeather_hip.so (booster_host.hip) on top of the C-ABI, so host and device can never disagree about a dimension.
struct ConvParam
{
    int output_channels, input_channels;
    int input_h, input_w;
    int kernel_h, kernel_w;
    int output_h, output_w;
    int stride_h, stride_w;
    int pad_left, pad_bottom, pad_right, pad_top;
    int group;
    bool bias_term;
    ActivationType activation;
    int batch; // GPU extension: images per Forward; 0 means 1

    void AssignOutputDim();                   // fhip_conv_assign_output_dim: defaults, floor dims, depthwise oc = ic
    void AssignPaddedDim();                   // fold the pads into input_h / input_w and clear them
    void LogParams(const char* layer_name);   // one block of text on stdout
    double GetFLOPS();                        // fhip_conv_flops, per image
};

typedef int (*GET_BUFFER_SIZE_FUNC)(ConvParam* param, int* buffer_size, int* processed_kernel_size);
typedef int (*INIT_FUNC)(ConvParam* param, float* processed_kernel, float* kernel);
typedef int (*FORWARD_FUNC)(ConvParam* param, float* output, float* input, float* kernel, float* buffer, float* bias_arr,
                            int num_threads);

// Thread-local HIP stream (a hipStream_t passed as void*) for Init / Forward.
void SetStream(void* hip_stream);
void* GetStream();

// ConvBooster doesn't allocate any memory.
class ConvBooster
{
public:
    ConvBooster();
    ~ConvBooster() {}
    int SelectAlgo(ConvParam* param);
    int ForceSelectAlgo(ConvAlgo algo);
    int SetFuncs();
    // GPU extension: byte counts as size_t for the currently selected algo.
    int GetBufferSizeBytes(ConvParam* param, size_t* buffer_bytes, size_t* processed_kernel_bytes);
    ConvAlgo Algo() const { return algo; }
    GET_BUFFER_SIZE_FUNC GetBufferSize;
    INIT_FUNC Init;
    FORWARD_FUNC Forward;

private:
    ConvAlgo algo;
};

} // namespace booster
