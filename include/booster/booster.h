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

enum ConvAlgo
{
    NAIVE,
    IM2COL,
    SGECONV,
    DEPTHWISE,
    WINOGRADF63,
    WINOGRADF63FUSED,
    WINOGRADF23,
};

enum ActivationType
{
    None,
    ReLU,
};

struct ConvParam
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
    bool bias_term;
    ActivationType activation;
    int batch; // GPU extension: images per Forward; 0 means 1

    void AssignOutputDim()
    {
        if (group == 0) group = 1;
        if (stride_h == 0) stride_h = 1;
        if (stride_w == 0) stride_w = 1;
        output_h = (input_h + pad_top + pad_bottom - kernel_h) / stride_h + 1;
        output_w = (input_w + pad_left + pad_right - kernel_w) / stride_w + 1;
        if (group == input_channels) output_channels = input_channels;
    }
    void AssignPaddedDim()
    {
        input_h = input_h + pad_top + pad_bottom;
        input_w = input_w + pad_left + pad_right;
        pad_left = pad_bottom = pad_right = pad_top = 0;
    }
    void LogParams(const char* layer_name)
    {
        printf("-----Layer %s ConvParam----\n", layer_name);
        printf("Input CxHxW=(%d, %d, %d)\n", input_channels, input_h, input_w);
        printf("Output CxHxW=(%d, %d, %d)\n", output_channels, output_h, output_w);
        printf("Group = %d\n", group);
        printf("Kernel HxW=(%d, %d)\n", kernel_h, kernel_w);
        printf("Stride HxW=(%d, %d)\n", stride_h, stride_w);
        printf("Paddings (%d %d %d %d)\n", pad_left, pad_bottom, pad_right, pad_top);
        printf("Batch = %d\n", batch > 0 ? batch : 1);
    }
    double GetFLOPS()
    {
        return 2.0 * this->output_channels * this->input_channels * this->output_h * this->output_w * this->kernel_h *
               this->kernel_w / this->group;
    }
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
