// oracle/net_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// extern "C" shim (ours) over the *real* reference runtime feather::Net (reference src/net.h:30-70, src/net.cpp),
// compiled together with the reference's own sources where they lie under /root/reference (oracle/Makefile; nothing is
// copied).  Driving sequence is the documented one: LoadParam -> LoadWeights -> FeedInput -> Forward -> Extract(ptr)
// (net.cpp:54-334; the Mat overload of Extract is broken).
// The reference is N = 1: callers loop over the images of a batch.  The reference prints progress to
// stdout/stderr from LoadParam/Forward; the shim silences both file descriptors for the duration of a call.

#include <fcntl.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>

#include "net.h"

namespace
{
struct Quiet
{
    int out, err;
    Quiet()
    {
        fflush(stdout);
        fflush(stderr);
        out = dup(1);
        err = dup(2);
        const int nul = open("/dev/null", O_WRONLY);
        dup2(nul, 1);
        dup2(nul, 2);
        close(nul);
    }
    ~Quiet()
    {
        fflush(stdout);
        fflush(stderr);
        dup2(out, 1);
        dup2(err, 2);
        close(out);
        close(err);
    }
};
} // namespace

extern "C"
{

__attribute__((visibility("default"))) void* ref_net_open(const char* param_path, const char* bin_path)
{
    Quiet q;
    feather::Net* net = new feather::Net();
    if (net->LoadParam(param_path) != 0) return nullptr;
    if (net->LoadWeights(bin_path) != 0) return nullptr;
    return net;
}

// One image [c][h][w] in, the named blob out.  Returns the element count written (<= capacity) or a negative code.
__attribute__((visibility("default"))) long ref_net_run(void* handle, const char* input_name, const float* image, int c, int h, int w,
                                                        const char* output_name, float* out, long capacity, int* dims)
{
    Quiet q;
    feather::Net* net = static_cast<feather::Net*>(handle);
    ncnn::Mat in(w, h, c);
    for (int ch = 0; ch < c; ++ch) memcpy(in.channel(ch), image + (size_t)ch * h * w, sizeof(float) * h * w);
    if (net->FeedInput(input_name, in) != 0) return -1;
    if (net->Forward() != 0) return -2;
    float* ptr = nullptr;
    int n = 0, oc = 0, oh = 0, ow = 0;
    if (net->Extract(std::string(output_name), &ptr, &n, &oc, &oh, &ow) != 0) return -3;
    const long count = (long)n * oc * oh * ow;
    if (count > capacity) return -4;
    memcpy(out, ptr, sizeof(float) * count);
    if (dims)
    {
        dims[0] = n;
        dims[1] = oc;
        dims[2] = oh;
        dims[3] = ow;
    }
    return count;
}

// Seconds per Forward (best of `reps` after one warm-up) on the image already fed.
__attribute__((visibility("default"))) double ref_net_time(void* handle, int reps)
{
    Quiet q;
    feather::Net* net = static_cast<feather::Net*>(handle);
    net->Forward();
    double best = 1e30;
    for (int r = 0; r < reps; ++r)
    {
        timespec a, b;
        clock_gettime(CLOCK_MONOTONIC, &a);
        net->Forward();
        clock_gettime(CLOCK_MONOTONIC, &b);
        const double s = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
        if (s < best) best = s;
    }
    return best;
}

// The named blob of the LAST forward (ref_net_run / ref_net_time*), no new forward: several blobs of one reference run.
__attribute__((visibility("default"))) long ref_net_extract(void* handle, const char* output_name, float* out, long capacity, int* dims)
{
    Quiet q;
    feather::Net* net = static_cast<feather::Net*>(handle);
    float* ptr = nullptr;
    int n = 0, oc = 0, oh = 0, ow = 0;
    if (net->Extract(std::string(output_name), &ptr, &n, &oc, &oh, &ow) != 0) return -3;
    const long count = (long)n * oc * oh * ow;
    if (count > capacity) return -4;
    memcpy(out, ptr, sizeof(float) * count);
    if (dims)
    {
        dims[0] = n;
        dims[1] = oc;
        dims[2] = oh;
        dims[3] = ow;
    }
    return count;
}

// `warmup` untimed forwards, then `reps` forwards timed one by one (seconds each -> secs[0 .. reps)) on the image already fed
// (warm-up 1 + >= 3 timed reps, clock_gettime like the reference's helper.cpp:89-98).
__attribute__((visibility("default"))) void ref_net_time_each(void* handle, int warmup, int reps, double* secs)
{
    Quiet q;
    feather::Net* net = static_cast<feather::Net*>(handle);
    for (int r = 0; r < warmup; ++r) net->Forward();
    for (int r = 0; r < reps; ++r)
    {
        timespec a, b;
        clock_gettime(CLOCK_MONOTONIC, &a);
        net->Forward();
        clock_gettime(CLOCK_MONOTONIC, &b);
        secs[r] = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
    }
}

__attribute__((visibility("default"))) void ref_net_close(void* handle)
{
    Quiet q;
    delete static_cast<feather::Net*>(handle);
}

} // extern "C"
