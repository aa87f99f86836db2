// tests/cpp/reference_style_main.cpp -- an application written against the REFERENCE's public API only (reference src/net.h:30-70:
// LoadParam / LoadWeights / FeedInput(const char*, ncnn::Mat&) / Forward / Extract(std::string, ncnn::Mat&) / Extract(name, float**,
// n, c, h, w)), the way FeatherCNN's own test programs drive a Net (reference test/test_bin.cpp).  It must compile against include/
// and link against libfeather_hip.so without a single source change.
// usage: reference_style_main model.param model.bin input.f32 c h w output_blob out.f32   (one image)
#include <net.h>

#include <stdio.h>
#include <stdlib.h>

using namespace feather;

int main(int argc, char* argv[])
{
    if (argc < 9) return 2;
    const int c = atoi(argv[4]), h = atoi(argv[5]), w = atoi(argv[6]);
    Net forward_net;
    if (forward_net.LoadParam(argv[1]) != 0) return 3;
    if (forward_net.LoadWeights(argv[2]) != 0) return 4;
    ncnn::Mat in(w, h, c);
    FILE* fp = fopen(argv[3], "rb");
    if (!fp) return 5;
    for (int q = 0; q < c; ++q)
    {
        float* plane = in.channel(q); // the Mat's channel stride may be padded: fill channel by channel, like ncnn users do
        if (fread(plane, sizeof(float), (size_t)w * h, fp) != (size_t)w * h) return 6;
    }
    fclose(fp);
    if (forward_net.FeedInput("data", in) != 0) return 7;
    if (forward_net.Forward() != 0) return 8;
    ncnn::Mat out;
    if (forward_net.Extract(std::string(argv[7]), out) != 0) return 9;
    float* blob_ptr = NULL;
    int on, oc, oh, ow;
    if (forward_net.Extract(std::string(argv[7]), &blob_ptr, &on, &oc, &oh, &ow) != 0) return 10;
    if (on != 1 || oc != out.c || oh != out.h || ow != out.w) return 11;
    fp = fopen(argv[8], "wb");
    for (int q = 0; q < out.c; ++q)
    {
        const float* plane = out.channel(q);
        fwrite(plane, sizeof(float), (size_t)out.w * out.h, fp);
    }
    fclose(fp);
    printf("reference-style main ok %d %d %d\n", out.c, out.h, out.w);
    return 0;
}
