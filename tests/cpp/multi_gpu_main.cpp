// tests/cpp/multi_gpu_main.cpp -- BASELINE config 5 (batch sharding over the GPUs of one node) from C++, no Python:
//
//   rank 0 reads the ncnn .bin, every rank receives it in DEVICE memory by ONE ncclBroadcast (RCCL over xGMI) and hands the buffer to
//   its own feather::Net (LoadWeightsDevice); every rank then feeds its contiguous share of the batch, runs Forward and extracts its
//   share of the output.  There is no steady-state collective.  The shares put back together must equal a single-device run of the
//   whole batch.
//
//   usage: multi_gpu_main <model.param> <model.bin> <input blob> <output blob> <batch> <c> <h> <w> [ranks_per_device]
//
// One host thread per rank.  RCCL places one rank on each visible device (a communicator cannot hold two ranks of one device); with
// ranks_per_device > 1 the further ranks of a device take the image from that device's first rank by hipMemcpyPeerAsync -- the
// documented fallback hand-over, and the way to rehearse the N > 1 logic on a one-GPU box (rank count = devices x ranks_per_device).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

#include "feather/net.h"

#define CK(x)                                                                            \
    do                                                                                   \
    {                                                                                    \
        hipError_t e__ = (x);                                                            \
        if (e__ != hipSuccess)                                                           \
        {                                                                                \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
            exit(2);                                                                     \
        }                                                                                \
    } while (0)
#define NK(x)                                                                               \
    do                                                                                      \
    {                                                                                       \
        ncclResult_t r__ = (x);                                                             \
        if (r__ != ncclSuccess)                                                             \
        {                                                                                   \
            printf("RCCL error %s at %s:%d\n", ncclGetErrorString(r__), __FILE__, __LINE__); \
            exit(3);                                                                        \
        }                                                                                   \
    } while (0)

// images [lo, hi) of rank r: sizes differ by at most one (feathercnn_amd/shard.py::shard_range, the rule bench.py uses)
static void shard_range(int batch, int rank, int world, int* lo, int* hi)
{
    const int base = batch / world, rem = batch % world;
    *lo = rank * base + std::min(rank, rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
}

static int run_net(feather::Net& net, const char* in_name, const char* out_name, int n, int c, int h, int w, const float* images, std::vector<float>& out)
{
    if (net.FeedInput(in_name, n, c, h, w, images)) return 1;
    if (net.Forward()) return 2;
    float* dev = NULL;
    int on = 0, oc = 0, oh = 0, ow = 0;
    if (net.Extract(out_name, &dev, &on, &oc, &oh, &ow)) return 3;
    out.resize((size_t)on * oc * oh * ow);
    return net.ExtractHost(out_name, out.data(), out.size()) ? 4 : 0;
}

int main(int argc, char** argv)
{
    if (argc < 9)
    {
        printf("usage: %s model.param model.bin input output batch c h w [ranks_per_device]\n", argv[0]);
        return 1;
    }
    const char *param = argv[1], *bin = argv[2], *in_name = argv[3], *out_name = argv[4];
    const int batch = atoi(argv[5]), c = atoi(argv[6]), h = atoi(argv[7]), w = atoi(argv[8]);
    const int rpd = argc > 9 ? std::max(1, atoi(argv[9])) : 1;
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    if (ndev < 1)
    {
        printf("no device\n");
        return 1;
    }
    const int world = ndev * rpd;

    // ---- rank 0: the model image from disk
    std::vector<char> image;
    {
        FILE* fp = fopen(bin, "rb");
        if (!fp)
        {
            printf("cannot open %s\n", bin);
            return 1;
        }
        fseek(fp, 0, SEEK_END);
        image.resize((size_t)ftell(fp));
        fseek(fp, 0, SEEK_SET);
        if (fread(image.data(), 1, image.size(), fp) != image.size()) return 1;
        fclose(fp);
    }
    const size_t len = image.size(); // (separate processes would broadcast this 8-byte length first)

    // ---- synthetic batch (host), the same for the sharded and the single-device run
    const size_t per_image = (size_t)c * h * w;
    std::vector<float> x((size_t)batch * per_image);
    unsigned s = 20260924u;
    for (auto& v : x)
    {
        s = s * 1664525u + 1013904223u;
        v = (s >> 8) * (2.f / 16777216.f) - 1.f;
    }

    // ---- the one collective: ncclBroadcast of the .bin from rank 0, one RCCL rank per device
    std::vector<void*> recv(world, (void*)NULL);
    std::vector<hipStream_t> streams(ndev);
    std::vector<ncclComm_t> comms(ndev);
    std::vector<int> devs(ndev);
    for (int d = 0; d < ndev; ++d) devs[d] = d;
    NK(ncclCommInitAll(comms.data(), ndev, devs.data()));
    for (int d = 0; d < ndev; ++d)
    {
        CK(hipSetDevice(d));
        CK(hipStreamCreate(&streams[d]));
        for (int k = 0; k < rpd; ++k) CK(hipMalloc(&recv[d * rpd + k], len));
    }
    CK(hipSetDevice(0));
    CK(hipMemcpy(recv[0], image.data(), len, hipMemcpyHostToDevice)); // rank 0's copy; nobody else ever sees the host image
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    CK(hipEventRecord(t0, streams[0]));
    NK(ncclGroupStart());
    for (int d = 0; d < ndev; ++d) NK(ncclBroadcast(recv[d * rpd], recv[d * rpd], len, ncclChar, 0, comms[d], streams[d]));
    NK(ncclGroupEnd());
    for (int d = 0; d < ndev; ++d) // further ranks of a device: peer copy from the device's first rank
    {
        CK(hipSetDevice(d));
        for (int k = 1; k < rpd; ++k) CK(hipMemcpyPeerAsync(recv[d * rpd + k], d, recv[d * rpd], d, len, streams[d]));
    }
    CK(hipSetDevice(0));
    CK(hipEventRecord(t1, streams[0]));
    for (int d = 0; d < ndev; ++d)
    {
        CK(hipSetDevice(d));
        CK(hipStreamSynchronize(streams[d]));
    }
    float bcast_ms = 0;
    CK(hipEventElapsedTime(&bcast_ms, t0, t1));

    // ---- every rank: its own Net on its device, weights from the broadcast buffer, its share of the batch
    std::vector<std::vector<float>> shares(world);
    std::vector<int> rcs(world, 0);
    std::vector<std::string> errs(world);
    std::vector<std::thread> threads;
    for (int r = 0; r < world; ++r)
        threads.emplace_back([&, r] {
            if (hipSetDevice(r / rpd) != hipSuccess)
            {
                rcs[r] = 90;
                return;
            }
            int lo, hi;
            shard_range(batch, r, world, &lo, &hi);
            if (hi <= lo) return; // more ranks than images
            feather::Net net;
            net.SetFusion(3);
            net.SetTunedSelection(true);
            net.SetConcurrency(true);
            int rc = net.LoadParam(param);
            if (!rc) rc = net.LoadWeightsDevice(recv[r], len);
            if (!rc) rc = run_net(net, in_name, out_name, hi - lo, c, h, w, x.data() + (size_t)lo * per_image, shares[r]);
            if (rc) errs[r] = feather::Net::LastError();
            rcs[r] = rc;
        });
    for (auto& t : threads) t.join();
    for (int r = 0; r < world; ++r)
        if (rcs[r])
        {
            printf("rank %d failed (%d): %s\n", r, rcs[r], errs[r].c_str());
            return 4;
        }

    // ---- the same batch on one device, weights from the host image
    CK(hipSetDevice(0));
    std::vector<float> whole;
    {
        feather::Net net;
        net.SetFusion(3);
        net.SetTunedSelection(true);
        net.SetConcurrency(true);
        if (net.LoadParam(param) || net.LoadWeightsMem(image.data(), len) || run_net(net, in_name, out_name, batch, c, h, w, x.data(), whole))
        {
            printf("single-device run failed: %s\n", feather::Net::LastError());
            return 5;
        }
    }
    const size_t per_out = whole.size() / batch;
    double worst = 0, scale = 0;
    for (float v : whole) scale = std::max(scale, (double)std::fabs(v));
    for (int r = 0; r < world; ++r)
    {
        int lo, hi;
        shard_range(batch, r, world, &lo, &hi);
        if ((size_t)(hi - lo) * per_out != shares[r].size())
        {
            printf("rank %d returned %zu floats for %d images\n", r, shares[r].size(), hi - lo);
            return 6;
        }
        for (size_t i = 0; i < shares[r].size(); ++i) worst = std::max(worst, (double)std::fabs(shares[r][i] - whole[(size_t)lo * per_out + i]));
    }
    const double nerr = worst / std::max(scale, 1e-30);
    printf("multi_gpu: %d device(s) x %d rank(s) per device = %d ranks, batch %d, .bin %.1f MB broadcast in %.3f ms (ncclBroadcast over %d RCCL rank(s)%s), "
           "sharded vs single-device normalised max error %.3e\n",
           ndev, rpd, world, batch, len / 1e6, bcast_ms, ndev, rpd > 1 ? " + hipMemcpyPeerAsync to the co-located ranks" : "", nerr);
    for (int d = 0; d < ndev; ++d)
    {
        CK(hipSetDevice(d));
        for (int k = 0; k < rpd; ++k) CK(hipFree(recv[d * rpd + k]));
        NK(ncclCommDestroy(comms[d]));
        CK(hipStreamDestroy(streams[d]));
    }
    if (!(nerr <= 1e-5))
    {
        printf("multi_gpu FAILED: shares differ from the single-device run\n");
        return 7;
    }
    printf("multi_gpu OK\n");
    return 0;
}
