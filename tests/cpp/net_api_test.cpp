// tests/cpp/net_api_test.cpp -- plain g++ against include/feather/net.h + libfeather_hip.so: the reference's Net call
// sequence (reference src/net.cpp:54-334).  usage: net_api_test model.param model.bin [input.f32 n c h w output_blob out.f32]
// Without the optional arguments only the host-side part runs (LoadParam / LoadWeights / error codes): no GPU needed.
#include <feather/net.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static int check(bool ok, const char* what)
{
    if (!ok) printf("FAILED: %s (%s)\n", what, feather::Net::LastError());
    return ok ? 0 : 1;
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    int bad = 0;
    {
        feather::Net broken;
        bad += check(broken.LoadParam("/nonexistent.param") == -1, "missing param file -> -1");
        const char* old_magic = "123\n1 1\nInput data 0 1 data\n";
        bad += check(broken.LoadParamMem(old_magic, strlen(old_magic)) == -1, "bad magic -> -1");
        const char* unknown = "7767517\n1 1\nFoo a 0 1 a\n";
        bad += check(broken.LoadParamMem(unknown, strlen(unknown)) == -200, "unknown layer -> -200");
        bad += check(broken.LoadWeights(argv[2]) != 0, "LoadWeights before LoadParam fails");
    }
    feather::Net net;
    FILE* fp = fopen(argv[1], "r");
    bad += check(fp != NULL && net.LoadParam(fp) == 0, "LoadParam(FILE*)");
    if (fp) fclose(fp);
    bad += check(net.LoadWeights(argv[2]) == 0, "LoadWeights(path)");
    bad += check(net.LayerCount() > 0, "LayerCount");
    if (argc >= 10)
    {
        const int n = atoi(argv[4]), c = atoi(argv[5]), h = atoi(argv[6]), w = atoi(argv[7]);
        std::vector<float> in((size_t)n * c * h * w);
        fp = fopen(argv[3], "rb");
        bad += check(fp && fread(in.data(), sizeof(float), in.size(), fp) == in.size(), "read input");
        if (fp) fclose(fp);
        bad += check(net.FeedInput("data", n, c, h, w, in.data()) == 0, "FeedInput");
        bad += check(net.Forward() == 0, "Forward");
        float* dev = NULL;
        int on, oc, oh, ow;
        bad += check(net.Extract(argv[8], &dev, &on, &oc, &oh, &ow) == 0 && dev != NULL && on == n, "Extract (device pointer)");
        std::vector<float> out((size_t)on * oc * oh * ow);
        bad += check(net.ExtractHost(argv[8], out.data(), out.size()) == 0, "ExtractHost");
        bad += check(net.Extract("no_such_blob", &dev, &on, &oc, &oh, &ow) == -1, "unknown blob -> -1");
        fp = fopen(argv[9], "wb");
        fwrite(out.data(), sizeof(float), out.size(), fp);
        fclose(fp);
        printf("net forward ok %d %d %d %d\n", on, oc, oh, ow);
    }
    if (bad == 0) printf("net api ok\n");
    return bad;
}
