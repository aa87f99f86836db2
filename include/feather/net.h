// feather/net.h -- C++ host class feather::Net over the MI355X runtime: the reference's public Net API
// (reference src/net.h:30-70, src/net.cpp) with the same method names, argument meaning and return codes, so code written
// against the reference links against libfeather_hip.so instead of libfeather.a:
//
//     feather::Net net;
//     net.LoadParam("model.param");  net.LoadWeights("model.bin");
//     ncnn::Mat in(w, h, c);  ...  net.FeedInput("data", in);   // net.h:44 (or the pointer forms below, with a batch)
//     net.Forward();
//     ncnn::Mat out;  net.Extract("prob", out);                  // net.h:50: host copy
//     float* dev; int n, c, h, w;  net.Extract("prob", &dev, &n, &c, &h, &w);   // net.h:48: pointer into the blob
//
// Differences, all forced by the GPU batch path:
//   * blobs live in HBM: Extract(name, float**, ...) returns a DEVICE pointer (use ExtractHost for a host copy);
//   * besides the reference's FeedInput(name, ncnn::Mat&) (N = 1, net.cpp:235-246) there are pointer forms with an explicit batch;
//   * Extract(name, ncnn::Mat&) copies channel by channel like the reference (net.cpp:281-296) -- but every channel, where the
//     reference copies channel 0 into all of them (its source pointer never advances); a batch > 1 comes back as n*c channels;
//   * public data members of the reference class are not mirrored: `blob_map` (std::map<std::string, Blob<float>*>, net.h:54)
//     exposes host Blob objects that do not exist here -- use Extract / LayerCount / the C-ABI introspection instead -- and the
//     feather::Layer / Blob classes (layer.h:29-88, blob.h) are internal to the device runtime;
//   * Forward only enqueues work on the net's HIP stream (SetStream); ExtractHost / Synchronize wait for it;
//   * SetFusion: 0 = none (what the reference actually does: TryFuse is never called),
//     1 = the reference's declared Conv-ReLU / BN-Scale-ReLU / InnerProduct-ReLU patterns (default), 2 = also fold
//     BatchNorm/Scale into the preceding convolution's weights.
// Header-only: every method forwards to the C-ABI in feather_hip/feather_net.h.
#pragma once

#include <stdio.h>
#include <string.h>

#include <string>

#include "feather_hip/feather_net.h"
#include "ncnn/mat.h"

namespace feather
{

class Net
{
  public:
    Net() : net_(NULL) { fhip_net_create(&net_); }
    ~Net()
    {
        if (net_) fhip_net_destroy(net_);
    }

    // Net::LoadParam / LoadWeights (net.cpp:54-233): 0 on success, negative on failure (-1 I/O, -200 unknown layer,
    // -300 topology error, -100 bad layer parameters), message in LastError().
    int LoadParam(const char* param_path) { return fhip_net_load_param(net_, param_path); }
    int LoadParam(FILE* fp) { return load_file(fp, true); }
    int LoadWeights(const char* weights_path) { return fhip_net_load_weights(net_, weights_path); }
    int LoadWeights(FILE* fp) { return load_file(fp, false); }
    int LoadParamMem(const char* text, size_t len) { return fhip_net_load_param_mem(net_, text, len); }
    int LoadWeightsMem(const void* data, size_t len) { return fhip_net_load_weights_mem(net_, data, len); }
    // the .bin image in device memory (the receive buffer of the RCCL weight broadcast, one rank per GPU)
    int LoadWeightsDevice(const void* device_data, size_t len) { return fhip_net_load_weights_device(net_, device_data, len); }

    // Net::FeedInput(const char*, ncnn::Mat&), net.h:44 / net.cpp:235-246 + Blob::CopyFromMat (blob.cpp:71-95): a host Mat of
    // shape (w, h, c), fp32, copied channel by channel (the Mat's channel stride may be padded to 16 bytes).  -1 for an unknown
    // blob name, -500 for a Mat that is not 3-D fp32.
    int FeedInput(const char* input_name, ncnn::Mat& in)
    {
        if (in.dims != 3 || in.elemsize != 4u || !in.data) return -500; // BAD DATA DIMENSION (blob.cpp:84)
        const size_t plane = (size_t)in.w * in.h;
        if (in.cstep == plane) return fhip_net_feed_input(net_, input_name, 1, in.c, in.h, in.w, (const float*)in.data, 0);
        std::string dense;
        dense.resize(plane * in.c * sizeof(float));
        for (int q = 0; q < in.c; ++q) memcpy(&dense[(size_t)q * plane * sizeof(float)], (const float*)in.data + in.cstep * q, plane * sizeof(float));
        return fhip_net_feed_input(net_, input_name, 1, in.c, in.h, in.w, (const float*)dense.data(), 0);
    }
    // The same with plain pointers and a batch.  `data` = n*c*h*w floats, NCHW dense, host memory.
    int FeedInput(const char* input_name, int c, int h, int w, const float* data) { return fhip_net_feed_input(net_, input_name, 1, c, h, w, data, 0); }
    int FeedInput(const char* input_name, int n, int c, int h, int w, const float* data) { return fhip_net_feed_input(net_, input_name, n, c, h, w, data, 0); }
    int FeedInputDevice(const char* input_name, int n, int c, int h, int w, const float* device_data)
    {
        return fhip_net_feed_input(net_, input_name, n, c, h, w, device_data, 1);
    }

    int Forward() { return fhip_net_forward(net_); } // net.cpp:297-334

    // Net::Extract(name, float**, n, c, h, w), net.cpp:263-279; *output_ptr is a DEVICE pointer.
    int Extract(std::string blob_name, float** output_ptr, int* n, int* c, int* h, int* w)
    {
        return fhip_net_extract(net_, blob_name.c_str(), output_ptr, n, c, h, w);
    }
    // Net::Extract(std::string, ncnn::Mat&), net.h:50 / net.cpp:281-296: a host copy shaped (w, h, c) -- (w, h, n*c) for a batch.
    int Extract(std::string blob_name, ncnn::Mat& out)
    {
        float* dev = NULL;
        int n = 0, c = 0, h = 0, w = 0;
        int rc = fhip_net_extract(net_, blob_name.c_str(), &dev, &n, &c, &h, &w);
        if (rc) return rc;
        const size_t plane = (size_t)w * h, count = plane * c * n;
        out.create(w, h, c * n, 4u);
        if (out.empty() && count) return -1;
        if (out.cstep == plane) return fhip_net_extract_host(net_, blob_name.c_str(), (float*)out.data, count);
        std::string dense;
        dense.resize(count * sizeof(float));
        rc = fhip_net_extract_host(net_, blob_name.c_str(), (float*)&dense[0], count);
        if (rc) return rc;
        for (int q = 0; q < c * n; ++q) memcpy((float*)out.data + out.cstep * q, &dense[(size_t)q * plane * sizeof(float)], plane * sizeof(float));
        return 0;
    }
    int ExtractHost(std::string blob_name, float* host, size_t capacity_floats) { return fhip_net_extract_host(net_, blob_name.c_str(), host, capacity_floats); }

    int SetStream(void* hip_stream) { return fhip_net_set_stream(net_, hip_stream); }
    int SetFusion(int level) { return fhip_net_set_fusion(net_, level); }
    int SetGraph(bool on) { return fhip_net_set_graph(net_, on ? 1 : 0); }
    int SetTunedSelection(bool on) { return fhip_net_set_tuned_selection(net_, on ? 1 : 0); }
    int SetConcurrency(bool on) { return fhip_net_set_concurrency(net_, on ? 1 : 0); }
    int SetSubBatches(int replicas) { return fhip_net_set_sub_batches(net_, replicas); } // before LoadParam (feather_net.h)
    int LayerCount() { return fhip_net_layer_count(net_); }
    static const char* LastError() { return fhip_last_error(); }
    fhip_net* handle() { return net_; }

  private:
    Net(const Net&);
    Net& operator=(const Net&);
    int load_file(FILE* fp, bool param)
    {
        if (!fp) return -1;
        std::string buf;
        char chunk[1 << 16];
        size_t got;
        if (param) fseek(fp, 0, SEEK_SET); // ChkParamHeader rewinds too (utils.cpp:29)
        while ((got = fread(chunk, 1, sizeof(chunk), fp)) > 0) buf.append(chunk, got);
        return param ? fhip_net_load_param_mem(net_, buf.data(), buf.size()) : fhip_net_load_weights_mem(net_, buf.data(), buf.size());
    }
    fhip_net* net_;
};

} // namespace feather
