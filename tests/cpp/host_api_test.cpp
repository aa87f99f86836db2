// tests/cpp/host_api_test.cpp -- compiled with plain g++ against include/booster/booster.h and linked with
// libfeather_hip.so: the same call sequence feather::ConvLayer performs (reference src/layers/conv_layer.h:92-172),
// restricted to the host-only part (SelectAlgo / GetBufferSize) so it runs without a GPU.  With a GPU present and
// FEATHER_HOST_TEST_GPU=1 it also runs Init + Forward on device buffers obtained through the HIP runtime (dlopen, so this
// file needs no HIP headers) -- see tests/test_parity_gpu.py::test_cpp_host_forward.
#include <booster/booster.h>

#include <cstdio>
#include <cstring>

static int check(bool ok, const char* what)
{
    if (!ok) printf("FAILED: %s\n", what);
    return ok ? 0 : 1;
}

int main()
{
    int bad = 0;
    booster::ConvParam conv_param;
    memset(&conv_param, 0, sizeof(conv_param));
    conv_param.input_channels = 64;
    conv_param.output_channels = 128;
    conv_param.input_h = conv_param.input_w = 56;
    conv_param.kernel_h = conv_param.kernel_w = 3;
    conv_param.pad_left = conv_param.pad_right = conv_param.pad_top = conv_param.pad_bottom = 1;
    conv_param.bias_term = true;
    conv_param.activation = booster::ReLU;
    conv_param.AssignOutputDim(); // group / stride default to 1
    bad += check(conv_param.group == 1 && conv_param.stride_h == 1 && conv_param.output_h == 56 && conv_param.output_w == 56, "AssignOutputDim");
    bad += check(conv_param.GetFLOPS() == 2.0 * 128 * 64 * 56 * 56 * 9, "GetFLOPS");

    booster::ConvBooster conv_booster;
    bad += check(conv_booster.GetBufferSize == NULL && conv_booster.Init == NULL && conv_booster.Forward == NULL, "unbound pointers are NULL");
    bad += check(conv_booster.SelectAlgo(&conv_param) == 0, "SelectAlgo");
    bad += check(conv_booster.Algo() == booster::WINOGRADF63, "3x3 s1 64->128 @56 selects WINOGRADF63");
    int buffer_size = 0, processed_kernel_size = 0;
    bad += check(conv_booster.GetBufferSize(&conv_param, &buffer_size, &processed_kernel_size) == 0, "GetBufferSize");
    bad += check(processed_kernel_size == 64 * 64 * 128, "packed kernel = 64*C*K floats");
    bad += check(buffer_size >= 64 * 100 * (64 + 128), "scratch holds V and M of one image");

    conv_param.batch = 32; // GPU extension: images per Forward
    size_t bb = 0, pb = 0;
    bad += check(conv_booster.GetBufferSizeBytes(&conv_param, &bb, &pb) == 0 && bb > 20u * (size_t)buffer_size * 4, "batched scratch in bytes");

    bad += check(conv_booster.ForceSelectAlgo(booster::IM2COL) == 0 && conv_booster.Forward != NULL, "ForceSelectAlgo(IM2COL)");
    bad += check(conv_booster.ForceSelectAlgo(booster::SGECONV) == -1 && conv_booster.Forward == NULL, "SGECONV unsupported -> -1, NULL pointers");

    booster::ConvParam dw = conv_param;
    dw.group = dw.input_channels = 32;
    dw.output_channels = 7; // AssignOutputDim forces it to input_channels for depthwise
    dw.AssignOutputDim();
    bad += check(dw.output_channels == 32, "depthwise forces output_channels");
    bad += check(conv_booster.SelectAlgo(&dw) == 0 && conv_booster.Algo() == booster::DEPTHWISE, "depthwise selection");

    booster::ConvParam pg = conv_param;
    pg.group = 2;
    bad += check(conv_booster.SelectAlgo(&pg) == -1, "partial group -> -1");

    booster::SetStream(NULL);
    bad += check(booster::GetStream() == NULL, "stream accessor");
    if (bad == 0) printf("host api ok\n");
    return bad;
}
