// net.hip -- feather::Net on device blobs.
//
// The host runtime either side of the ConvBooster hot path: the ncnn .param/.bin readers, the layer objects
// with the reference's LoadParam / LoadWeights / Reshape / Init / Forward life cycle, one shared device scratch
// arena in place of CommonMemPool, and the layer-fusion pass the reference declares but never runs.
// Reference (paths relative to /root/reference/src): net.cpp:31-349, layer.cpp:24-142, layers/*.h,
// ncnn/paramdict.cpp:92-174, ncnn/modelbin.cpp:47-197, mempool.cpp:88-92.
//
// What differs from the reference on purpose:
//   * blobs carry a batch dimension (the reference hard-codes num = 1 in every Reshape);
//   * blob data lives in HBM, weights are uploaded once at Init and the packed conv weights stay resident;
//   * everything is enqueued on one HIP stream; Forward returns after enqueueing (Extract-to-host synchronises);
//   * Reshape runs only when the fed input shape changed (the reference re-runs it every Forward, net.cpp:299);
//   * Split tops and Dropout(scale = 1) tops alias their bottom instead of copying it -- no layer writes in place,
//     so an aliased blob is never modified;
//   * optional hipGraph replay of the whole forward.
#include <ctype.h>
#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "feather_hip/feather_net.h"

namespace fhip
{
namespace net
{

static int failf(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return fail(code, buf);
}

// Error codes of the Net level follow the reference where it has one (net.cpp:111,134,276; layers return -100).
enum
{
    NET_E_IO = -1,
    NET_E_SHAPE = -100,
    NET_E_UNKNOWN_LAYER = -200,
    NET_E_TOPOLOGY = -300,
    NET_E_BASE_RESHAPE = -400
};

// ---- ncnn ParamDict (ncnn/paramdict.cpp:92-174): "id=value" pairs, value is float iff it contains '.' or 'e' -----
struct ParamDict
{
    static const int kMax = 32; // NCNN_MAX_PARAM_COUNT is 20
    struct Entry
    {
        bool loaded = false;
        uint32_t bits = 0; // union { int i; float f; } exactly like the reference
        std::vector<float> array;
    } e[kMax];

    void clear()
    {
        for (auto& x : e) x = Entry();
    }
    int get(int id, int def) const
    {
        if (id < 0 || id >= kMax || !e[id].loaded) return def;
        int v;
        memcpy(&v, &e[id].bits, 4);
        return v;
    }
    float get(int id, float def) const
    {
        if (id < 0 || id >= kMax || !e[id].loaded) return def;
        float v;
        memcpy(&v, &e[id].bits, 4);
        return v;
    }
    bool has_array(int id) const { return id >= 0 && id < kMax && e[id].loaded && !e[id].array.empty(); }

    static bool is_float(const std::string& s)
    {
        for (char ch : s)
            if (ch == '.' || ch == 'e' || ch == 'E') return true;
        return false;
    }
    // `tok` is one whitespace-delimited token "id=value" or "-233xx=len,v0,v1,..."; returns false if it is not a pair.
    static bool looks_like_pair(const std::string& tok)
    {
        size_t i = 0;
        if (i < tok.size() && tok[i] == '-') ++i;
        const size_t d0 = i;
        while (i < tok.size() && isdigit((unsigned char)tok[i])) ++i;
        return i > d0 && i < tok.size() && tok[i] == '=';
    }
    int parse(const std::string& tok)
    {
        const size_t eq = tok.find('=');
        int id = atoi(tok.substr(0, eq).c_str());
        const std::string val = tok.substr(eq + 1);
        const bool is_array = id <= -23300;
        if (is_array) id = -id - 23300;
        if (id < 0 || id >= kMax) return failf(FHIP_E_BADARG, "param id %d out of range", id);
        if (is_array)
        {
            std::vector<std::string> parts;
            size_t a = 0;
            while (a <= val.size())
            {
                const size_t b = val.find(',', a);
                parts.push_back(val.substr(a, b == std::string::npos ? std::string::npos : b - a));
                if (b == std::string::npos) break;
                a = b + 1;
            }
            const int len = atoi(parts[0].c_str());
            if (len < 0 || (int)parts.size() != len + 1) return failf(FHIP_E_BADARG, "ParamDict read array element fail");
            e[id].array.resize(len);
            for (int j = 0; j < len; ++j) e[id].array[j] = (float)atof(parts[j + 1].c_str());
        }
        else
        {
            if (val.empty()) return failf(FHIP_E_BADARG, "ParamDict read value fail");
            if (is_float(val))
            {
                const float f = (float)atof(val.c_str());
                memcpy(&e[id].bits, &f, 4);
            }
            else
            {
                const int i = atoi(val.c_str());
                memcpy(&e[id].bits, &i, 4);
            }
        }
        e[id].loaded = true;
        return 0;
    }
};

// ---- ncnn ModelBin (ncnn/modelbin.cpp:47-197) over a memory image of the .bin ---------------------------------
struct ModelBin
{
    const unsigned char* p;
    const unsigned char* end;

    static float half_to_float(uint16_t h)
    {
        const uint32_t sign = (uint32_t)(h >> 15) << 31;
        int exp = (h >> 10) & 0x1f;
        uint32_t man = h & 0x3ff;
        uint32_t bits;
        if (exp == 0)
        {
            if (man == 0)
                bits = sign;
            else
            {
                exp = 1;
                while (!(man & 0x400))
                {
                    man <<= 1;
                    --exp;
                }
                man &= 0x3ff;
                bits = sign | ((uint32_t)(exp + 112) << 23) | (man << 13);
            }
        }
        else if (exp == 31)
            bits = sign | 0x7f800000u | (man << 13);
        else
            bits = sign | ((uint32_t)(exp + 112) << 23) | (man << 13);
        float f;
        memcpy(&f, &bits, 4);
        return f;
    }
    bool take(void* dst, size_t bytes)
    {
        if ((size_t)(end - p) < bytes) return false;
        if (dst) memcpy(dst, p, bytes);
        p += bytes;
        return true;
    }
    // type 0: 4-byte tag then payload (raw fp32 / fp16 / 256-entry table); type 1: raw fp32 (modelbin.cpp:52-189)
    int load(size_t w, int type, std::vector<float>& out)
    {
        out.assign(w, 0.f);
        if (type == 1)
        {
            if (!take(out.data(), w * 4)) return failf(NET_E_SHAPE, "ModelBin read weight_data failed (file too short)");
            return 0;
        }
        if (type != 0) return failf(NET_E_SHAPE, "ModelBin load type %d not implemented", type);
        unsigned char f[4];
        if (!take(f, 4)) return failf(NET_E_SHAPE, "ModelBin read flag_struct failed");
        uint32_t tag;
        memcpy(&tag, f, 4);
        const unsigned flag = f[0] + f[1] + f[2] + f[3];
        if (tag == 0x01306B47u) // fp16 payload, padded to 4 bytes
        {
            const size_t bytes = (w * 2 + 3) / 4 * 4;
            if ((size_t)(end - p) < bytes) return failf(NET_E_SHAPE, "ModelBin read float16_weights failed");
            for (size_t i = 0; i < w; ++i)
            {
                uint16_t h;
                memcpy(&h, p + 2 * i, 2);
                out[i] = half_to_float(h);
            }
            p += bytes;
            return 0;
        }
        if (tag == 0x000D4B38u) return failf(FHIP_E_UNSUPPORTED, "int8 weights are not supported (conv_layer.h:50-55 rejects them too)");
        if (tag == 0x0002C056u)
        {
            if (!take(out.data(), w * 4)) return failf(NET_E_SHAPE, "ModelBin read weight_data failed");
            return 0;
        }
        if (flag != 0) // 256-entry codebook + one byte per weight, padded to 4 bytes
        {
            float table[256];
            if (!take(table, sizeof(table))) return failf(NET_E_SHAPE, "ModelBin read quantization_value failed");
            const size_t bytes = (w + 3) / 4 * 4;
            if ((size_t)(end - p) < bytes) return failf(NET_E_SHAPE, "ModelBin read index_array failed");
            for (size_t i = 0; i < w; ++i) out[i] = table[p[i]];
            p += bytes;
            return 0;
        }
        if (!take(out.data(), w * 4)) return failf(NET_E_SHAPE, "ModelBin read weight_data failed (file too short)");
        return 0;
    }
};

// ---- blobs ---------------------------------------------------------------------------------------------------
struct Blob
{
    std::string name;
    int n = 0, c = 0, h = 0, w = 0;
    float* data = nullptr;
    size_t capacity = 0;   // floats owned
    Blob* alias = nullptr; // data is another blob's
    bool fused_away = false;
    bool chained = false; // fusion level 3: lives only as the next layer's Winograd-domain input (plan_chains); shape kept, no storage

    size_t count() const { return (size_t)n * c * h * w; }
    int reshape(int n_, int c_, int h_, int w_)
    {
        n = n_;
        c = c_;
        h = h_;
        w = w_;
        alias = nullptr;
        // a blob the previous plan chained away has no storage and, as a rule, will have none after this Reshape either: plan_chains
        // allocates it afterwards if the new shapes un-chain it (no device malloc + free per chained blob and shape change)
        if (chained && !data) return 0;
        if (count() > capacity)
        {
            if (data && capacity) (void)hipFree(data);
            data = nullptr;
            capacity = 0;
            FHIP_CHECK_HIP(hipMalloc((void**)&data, count() * sizeof(float)));
            capacity = count();
        }
        return 0;
    }
    void drop_storage()
    {
        if (data && capacity) (void)hipFree(data);
        data = nullptr;
        capacity = 0;
    }
    void share(Blob* src)
    {
        n = src->n;
        c = src->c;
        h = src->h;
        w = src->w;
        if (data && capacity) (void)hipFree(data);
        capacity = 0;
        alias = src;
        data = src->data;
    }
    ~Blob()
    {
        if (data && capacity) (void)hipFree(data);
    }
};

struct DeviceVec
{
    float* d = nullptr;
    size_t bytes = 0;
    int upload(const float* h, size_t count, hipStream_t s)
    {
        if (bytes != count * 4)
        {
            release();
            FHIP_CHECK_HIP(hipMalloc((void**)&d, std::max<size_t>(count, 1) * 4));
            bytes = count * 4;
        }
        if (count) FHIP_CHECK_HIP(hipMemcpyAsync(d, h, count * 4, hipMemcpyHostToDevice, s));
        return 0;
    }
    int resize(size_t nbytes)
    {
        if (bytes != nbytes)
        {
            release();
            FHIP_CHECK_HIP(hipMalloc((void**)&d, std::max<size_t>(nbytes, 4)));
            bytes = nbytes;
        }
        return 0;
    }
    void release()
    {
        if (d) (void)hipFree(d);
        d = nullptr;
        bytes = 0;
    }
    ~DeviceVec() { release(); }
};

struct Net;

// ---- layers (reference layer.h:28-86) -------------------------------------------------------------------------
struct Layer
{
    std::string type, name;
    std::vector<Blob*> bottoms, tops;
    Net* net = nullptr;

    virtual ~Layer() {}
    virtual int LoadParam(const ParamDict&) { return 0; }
    virtual int LoadWeights(ModelBin&) { return 0; }
    // default (layer.cpp:103-114): one bottom, one top of the same shape
    virtual int Reshape()
    {
        if (tops.size() != 1 || bottoms.size() != 1) return failf(NET_E_BASE_RESHAPE, "layer %s: base Reshape needs 1 bottom and 1 top", name.c_str());
        return tops[0]->reshape(bottoms[0]->n, bottoms[0]->c, bottoms[0]->h, bottoms[0]->w);
    }
    virtual int Init(hipStream_t) { return 0; }
    virtual int Forward(hipStream_t) = 0;
    virtual int Fuse(Layer*, int /*level*/) { return 0; }
    virtual size_t weight_bytes() const { return 0; }
    virtual size_t arena_bytes() const { return 0; }
    virtual int algo() const { return -1; }
    virtual const fhip_conv_param* conv_param() const { return nullptr; }
    virtual const fhip_conv_param* fused_pointwise(int*) const { return nullptr; } // the 1x1 convolution a depthwise layer absorbed
    virtual void chain_state(int* v_from_previous, int* writes_next_v) const { *v_from_previous = *writes_next_v = 0; }
    virtual int sibling_state() const { return 0; } // 1: launches the GEMM that also computes the NEXT layer; 2: computed by the layer before
    virtual int residual_state() const { return 0; } // 1: an Eltwise SUM operand is added in this layer's GEMM epilogue; 2: absorbed, added by a separate launch
};

struct Net
{
    std::vector<std::unique_ptr<Layer>> layers;
    std::map<std::string, std::unique_ptr<Blob>> blobs;
    // Blobs whose NAME was taken over by a later top (in-place style files: `ReLU r 1 1 conv1 conv1`, or two layers with the
    // same top).  The reference leaks the old Blob and replaces the map entry (net.cpp:135-139), so earlier layers keep
    // using it; here it stays owned, the name resolves to the newest blob exactly like the reference's blob_map.
    std::vector<std::unique_ptr<Blob>> shadowed;
    hipStream_t stream = nullptr;
    int fusion = 1;
    bool use_graph = false;
    bool tuned_selection = false; // fhip_conv_select_algo_tuned instead of the reference's SelectAlgo rule
    bool param_loaded = false, weights_loaded = false, fused = false, initialized = false, shapes_dirty = true;
    DeviceVec arena;
    // fusion level 3 (plan_chains): runs of Winograd layers hand their transformed input from one to the next; the arena then holds two V
    // slots (layers at even / odd positions of a run) and M
    size_t chain_slot[2] = {0, 0}, chain_m = 0;
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t owned_stream = nullptr; // graph capture is not allowed on the NULL stream
    // Branch concurrency (fhip_net_set_concurrency): a convolution that needs no scratch arena and whose output is consumed
    // only further down the layer list (ResNet's projection shortcut, SqueezeNet's expand1x1) runs on a second stream while the
    // main stream continues with the layers in between; fork / join are events, so the pattern is hipGraph-capturable.
    bool concurrency = false;
    hipStream_t side_stream = nullptr;
    std::vector<hipEvent_t> events;        // [2 * pair]: fork, join
    std::vector<int> side;                 // per layer: its event pair, or -1 (runs on the main stream)
    std::vector<std::vector<int>> waits;   // per layer: pairs whose join event the main stream waits for first

    ~Net()
    {
        drop_graph();
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        if (owned_stream) (void)hipStreamDestroy(owned_stream);
    }
    void drop_graph()
    {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        graph_exec = nullptr;
    }
    Blob* find(const std::string& n)
    {
        auto it = blobs.find(n);
        return it == blobs.end() ? nullptr : it->second.get();
    }
};

struct InputLayer : Layer
{
    int Reshape() override { return 0; } // shape comes from FeedInput (input_layer.h:33-46)
    int Forward(hipStream_t) override { return 0; }
};

// feather::ConvLayer, layers/conv_layer.h:26-193 (also registered as ConvolutionDepthWise, layer_factory.cpp)
struct ConvLayer : Layer
{
    fhip_conv_param p;
    int algo_ = -1, inited_algo = -2;
    std::vector<float> w_host, b_host;
    std::vector<float> post_mul, post_add; // folded BatchNorm/Scale (fusion level 2)
    DeviceVec packed, bias;
    size_t buffer_bytes = 0, packed_bytes = 0;
    // fusion level 2: a following 2x2 / stride-2 max pooling is absorbed (fhip_conv_forward_maxpool2 when the route can,
    // else this layer runs conv -> pre_pool -> fhip_pooling itself)
    bool fuse_pool = false, pool_fast = false;
    fhip_pool_param poolq;
    DeviceVec pre_pool;
    // fusion level 2: a following Eltwise SUM (+ReLU) with an earlier blob is absorbed: top = act(conv + bias + residual)
    Blob* residual = nullptr;
    bool res_fast = false;
    // fusion level 2: this is a 3x3 depthwise layer and the 1x1 convolution that is its only consumer was absorbed (`pw` owns its
    // parameters, weights and further fusions); when the pair qualifies at the current shape the two run as ONE kernel
    // (fhip_conv_forward_dw_pw) and the depthwise output never exists, else dw -> `mid` -> pw one after the other
    std::unique_ptr<ConvLayer> pw;
    bool pair_fast = false;
    DeviceVec mid;
    // fusion level 3 (plan_chains, after Reshape): this Winograd layer's input arrives already transformed (chain_in) and / or its
    // output leaves as the next layer's transformed input (chain_next); chain_pos = position in the run (picks the V slot)
    ConvLayer* chain_next = nullptr;
    bool chain_in = false;
    int chain_pos = 0;
    size_t chain_bytes = 0; // arena the run needs (reported instead of buffer_bytes)
    // fusion level 3 (plan_chains): a first layer (3x3 / s1 / p1, 2 .. 4 input channels) whose only consumer is a Winograd layer is computed
    // inside that layer's input transform (fhip_winograd_f63_input_from_first): `head_of` = the consumer (this layer then launches nothing,
    // its top has no storage), `head` = the absorbed first layer (on the consumer); first_raw = this layer's filters as loaded
    ConvLayer* head_of = nullptr;
    ConvLayer* head = nullptr;
    DeviceVec first_raw;
    // fusion level 2 (plan_siblings, after Reshape): this 1x1 layer and the NEXT layer of the list -- a 1x1 layer on the same bottom with
    // the same stride: ResNet's projection shortcut and the first layer of the main branch -- run as ONE GEMM over the stacked filters
    // (fhip_conv_forward_siblings).  `sib` (on the first layer) = the second; `sib_of` (on the second, which launches nothing) = the first.
    ConvLayer* sib = nullptr;
    ConvLayer* sib_of = nullptr;
    ConvLayer* sib_packed_for = nullptr; // the partner sib_packed / sib_bias were built with
    DeviceVec sib_packed, sib_bias;
    bool sib_has_bias = false;
    bool first_candidate() const
    {
        return p.kernel_h == 3 && p.kernel_w == 3 && p.stride_h == 1 && p.stride_w == 1 && p.group == 1 && p.input_channels >= 2 && p.input_channels <= 4 && p.pad_left == 1 &&
               p.pad_right == 1 && p.pad_top == 1 && p.pad_bottom == 1;
    }

    ConvLayer()
    {
        memset(&p, 0, sizeof(p));
        memset(&poolq, 0, sizeof(poolq));
    }

    int LoadParam(const ParamDict& pd) override
    {
        const int dilation_w = pd.get(2, 1), dilation_h = pd.get(12, dilation_w);
        if (dilation_w > 1 || dilation_h > 1) return failf(NET_E_UNKNOWN_LAYER, "layer %s: dilated convolution is not supported", name.c_str());
        if (pd.get(8, 0)) return failf(NET_E_UNKNOWN_LAYER, "layer %s: int8 convolution is not supported", name.c_str());
        p.kernel_w = pd.get(1, 0);
        p.kernel_h = pd.get(11, p.kernel_w);
        p.stride_w = pd.get(3, 1);
        p.stride_h = pd.get(13, p.stride_w);
        p.pad_left = pd.get(4, 0);
        p.pad_bottom = pd.get(14, p.pad_left);
        p.pad_right = pd.get(4, 0);
        p.pad_top = pd.get(14, p.pad_left);
        p.group = pd.get(7, 1);
        p.output_channels = pd.get(0, 0);
        p.bias_term = pd.get(5, 0);
        p.activation = FHIP_ACT_NONE;
        const int weight_data_size = pd.get(6, 0);
        // conv_layer.h:69-75: output_channels is divided by group (AssignOutputDim restores it for depthwise)
        if (p.group == 0 || p.output_channels % p.group) return failf(NET_E_SHAPE, "layer %s: output_channels is not divisible by its group", name.c_str());
        p.output_channels /= p.group;
        if (p.output_channels <= 0 || p.kernel_h <= 0 || p.kernel_w <= 0) return failf(NET_E_SHAPE, "layer %s: bad convolution geometry", name.c_str());
        p.input_channels = weight_data_size / p.output_channels / p.kernel_h / p.kernel_w;
        return 0;
    }
    int LoadWeights(ModelBin& mb) override
    {
        const size_t wsize = (size_t)p.input_channels * p.output_channels * p.kernel_h * p.kernel_w;
        inited_algo = -2; // new weights: every packed form is rebuilt at the next Init
        sib_packed_for = nullptr;
        int rc = mb.load(wsize, 0, w_host);
        if (rc) return rc;
        if (p.bias_term)
        {
            const int k = p.group == p.input_channels ? p.input_channels : p.output_channels;
            rc = mb.load(k, 1, b_host);
        }
        return rc;
    }
    int Reshape() override
    {
        const Blob* b = bottoms[0];
        p.input_w = b->w;
        p.input_h = b->h;
        if (p.input_channels != b->c)
            return failf(NET_E_TOPOLOGY, "convolution layer %s has %d input channels while bottom blob has %d channels", name.c_str(), p.input_channels, b->c);
        fhip_conv_assign_output_dim(&p);
        if (p.output_h < 1 || p.output_w < 1) return failf(NET_E_SHAPE, "layer %s: empty output", name.c_str());
        int rc = net->tuned_selection ? fhip_conv_select_algo_tuned(&p, &algo_) : fhip_conv_select_algo(&p, &algo_);
        if (rc) return rc;
        if (pw)
        {
            // the absorbed 1x1 convolution reshapes against a stand-in for the depthwise output, then takes over this layer's top
            Blob dwout;
            dwout.n = b->n;
            dwout.c = p.output_channels;
            dwout.h = p.output_h;
            dwout.w = p.output_w;
            pw->bottoms.assign(1, &dwout);
            pw->tops = tops;
            pw->net = net;
            rc = pw->Reshape();
            pw->bottoms.clear();
            if (rc) return rc;
            pair_fast = fhip_conv_can_fuse_dw_pw(&p, &pw->p, b->n) != 0 && !pw->fuse_pool && !pw->residual && pw->algo_ == FHIP_IM2COL;
            if (!pair_fast && (rc = mid.resize((size_t)b->n * p.output_channels * p.output_h * p.output_w * sizeof(float)))) return rc;
            size_t bb = 0;
            if ((rc = fhip_conv_get_buffer_size(&p, algo_, b->n, &bb, &packed_bytes))) return rc;
            buffer_bytes = std::max(bb, pw->buffer_bytes);
            return 0;
        }
        res_fast = residual && fhip_conv_can_fuse_residual(&p, algo_) != 0;
        if (residual && (residual->n != b->n || residual->c != p.output_channels || residual->h != p.output_h || residual->w != p.output_w))
            return failf(NET_E_SHAPE, "Shape mismatch among bottoms of layer %s.", name.c_str());
        if (fuse_pool)
        {
            poolq.channels = p.output_channels;
            poolq.input_h = p.output_h;
            poolq.input_w = p.output_w;
            int ph, pw;
            if ((rc = fhip_pooling_output_dim(&poolq, &ph, &pw))) return rc;
            pool_fast = fhip_conv_can_fuse_maxpool2(&p, algo_) != 0;
            if (!pool_fast && (rc = pre_pool.resize((size_t)b->n * p.output_channels * p.output_h * p.output_w * sizeof(float)))) return rc;
            rc = tops[0]->reshape(b->n, p.output_channels, ph, pw);
        }
        else
            rc = tops[0]->reshape(b->n, p.output_channels, p.output_h, p.output_w);
        if (rc) return rc;
        return fhip_conv_get_buffer_size(&p, algo_, b->n, &buffer_bytes, &packed_bytes);
    }
    // filters and bias with the folded BatchNorm / Scale (fusion level 2) applied
    void folded(std::vector<float>& w, std::vector<float>& b) const
    {
        w = w_host;
        b = b_host;
        if (post_mul.empty()) return;
        const int K = p.output_channels;
        const size_t per = w.size() / K;
        if (b.empty()) b.assign(K, 0.f);
        for (int k = 0; k < K; ++k)
        {
            for (size_t i = 0; i < per; ++i) w[k * per + i] *= post_mul[k];
            b[k] = b[k] * post_mul[k] + post_add[k];
        }
    }
    int init_siblings(hipStream_t s)
    {
        std::vector<float> wa, ba, wb, bb;
        folded(wa, ba);
        sib->folded(wb, bb);
        fhip_conv_param pa = p, pb = sib->p, both;
        pa.bias_term = ba.empty() ? 0 : 1;
        pb.bias_term = bb.empty() ? 0 : 1;
        int rc = fhip_conv_siblings_geometry(&pa, &pb, &both);
        if (rc) return rc;
        wa.insert(wa.end(), wb.begin(), wb.end());
        sib_has_bias = both.bias_term != 0;
        if (sib_has_bias)
        {
            if (ba.empty()) ba.assign(p.output_channels, 0.f);
            if (bb.empty()) bb.assign(sib->p.output_channels, 0.f);
            ba.insert(ba.end(), bb.begin(), bb.end());
            if ((rc = sib_bias.upload(ba.data(), ba.size(), s))) return rc;
        }
        size_t bytes = 0, pk = 0;
        if ((rc = fhip_conv_get_buffer_size(&both, FHIP_IM2COL, bottoms[0]->n, &bytes, &pk))) return rc;
        DeviceVec raw;
        if ((rc = raw.upload(wa.data(), wa.size(), s))) return rc;
        if ((rc = sib_packed.resize(pk))) return rc;
        if ((rc = fhip_conv_init(&both, FHIP_IM2COL, sib_packed.d, raw.d, s))) return rc;
        FHIP_CHECK_HIP(hipStreamSynchronize(s)); // `raw` goes out of scope
        sib_packed_for = sib;
        return 0;
    }
    int Init(hipStream_t s) override
    {
        if (pw)
        {
            const int rc = pw->Init(s);
            if (rc) return rc;
        }
        if (sib && sib_packed_for != sib)
        {
            const int rc = init_siblings(s);
            if (rc) return rc;
        }
        if (inited_algo == algo_ && packed.bytes == packed_bytes) return 0;
        std::vector<float> w, b;
        folded(w, b);
        if (!post_mul.empty()) p.bias_term = 1;
        DeviceVec raw;
        int rc = raw.upload(w.data(), w.size(), s);
        if (rc) return rc;
        if (first_candidate() && (rc = first_raw.upload(w.data(), w.size(), s))) return rc; // 27 floats per output channel
        rc = packed.resize(packed_bytes);
        if (rc) return rc;
        rc = fhip_conv_init(&p, algo_, packed.d, raw.d, s);
        if (rc) return rc;
        if (p.bias_term)
        {
            rc = bias.upload(b.data(), b.size(), s);
            if (rc) return rc;
        }
        FHIP_CHECK_HIP(hipStreamSynchronize(s)); // `raw` and the host copies go out of scope
        inited_algo = algo_;
        return 0;
    }
    int Forward(hipStream_t s) override
    {
        const float* b = p.bias_term ? bias.d : nullptr;
        if (head_of) return 0; // computed inside head_of's input transform
        if (sib_of) return 0;  // computed by the layer before, in the same GEMM
        if (sib)
            return fhip_conv_forward_siblings(&p, &sib->p, bottoms[0]->n, tops[0]->data, sib->tops[0]->data, bottoms[0]->data, sib_packed.d,
                                              sib_has_bias ? sib_bias.d : nullptr, s);
        if (pw)
        {
            const float* pb = pw->p.bias_term ? pw->bias.d : nullptr;
            if (pair_fast)
                return fhip_conv_forward_dw_pw(&p, &pw->p, bottoms[0]->n, tops[0]->data, bottoms[0]->data, packed.d, b, pw->packed.d, pb, s);
            int rc = fhip_conv_forward(&p, algo_, bottoms[0]->n, mid.d, bottoms[0]->data, packed.d, (float*)net->arena.d, b, s);
            if (rc) return rc;
            Blob dwout; // stand-in for the depthwise output the pair no longer has a blob for
            dwout.n = bottoms[0]->n;
            dwout.c = p.output_channels;
            dwout.h = p.output_h;
            dwout.w = p.output_w;
            dwout.data = mid.d;
            pw->bottoms.assign(1, &dwout);
            pw->tops = tops;
            rc = pw->Forward(s);
            pw->bottoms.clear();
            dwout.data = nullptr;
            return rc;
        }
        if (chain_in || chain_next || head)
        {
            char* base = reinterpret_cast<char*>(net->arena.d);
            float* v = reinterpret_cast<float*>(base + net->chain_slot[chain_pos & 1]);
            float* vn = reinterpret_cast<float*>(base + net->chain_slot[(chain_pos + 1) & 1]);
            float* m = reinterpret_cast<float*>(base + net->chain_m);
            const float* in = chain_in || head ? nullptr : bottoms[0]->data;
            const int n = bottoms[0]->n;
            if (head)
            {
                const int rc = fhip_winograd_f63_input_from_first(&head->p, &p, n, v, head->bottoms[0]->data, head->first_raw.d,
                                                                  head->p.bias_term ? head->bias.d : nullptr, s);
                if (rc) return rc;
            }
            if (chain_next) return fhip_conv_forward_chained(&p, n, nullptr, in, packed.d, v, m, b, &chain_next->p, vn, fuse_pool ? 1 : 0, s);
            if (!fuse_pool || pool_fast) return fhip_conv_forward_chained(&p, n, tops[0]->data, in, packed.d, v, m, b, nullptr, nullptr, fuse_pool ? 1 : 0, s);
            const int rc = fhip_conv_forward_chained(&p, n, pre_pool.d, in, packed.d, v, m, b, nullptr, nullptr, 0, s);
            if (rc) return rc;
            return fhip_pooling(&poolq, n, tops[0]->data, pre_pool.d, s);
        }
        if (residual)
        {
            if (residual->alias) residual->data = residual->alias->data;
            if (res_fast)
                return fhip_conv_forward_residual(&p, algo_, bottoms[0]->n, tops[0]->data, bottoms[0]->data, packed.d, (float*)net->arena.d, b,
                                                  residual->data, s);
            fhip_conv_param q = p; // route without the fused epilogue: conv, then the add (+ReLU) in place on the top
            q.activation = FHIP_ACT_NONE;
            const int rc = fhip_conv_forward(&q, algo_, bottoms[0]->n, tops[0]->data, bottoms[0]->data, packed.d, (float*)net->arena.d, b, s);
            if (rc) return rc;
            return fhip_add(tops[0]->data, tops[0]->data, residual->data, tops[0]->count(), p.activation == FHIP_ACT_RELU, s);
        }
        if (!fuse_pool) return fhip_conv_forward(&p, algo_, bottoms[0]->n, tops[0]->data, bottoms[0]->data, packed.d, (float*)net->arena.d, b, s);
        if (pool_fast)
            return fhip_conv_forward_maxpool2(&p, algo_, bottoms[0]->n, tops[0]->data, bottoms[0]->data, packed.d, (float*)net->arena.d, b, s);
        const int rc = fhip_conv_forward(&p, algo_, bottoms[0]->n, pre_pool.d, bottoms[0]->data, packed.d, (float*)net->arena.d, b, s);
        if (rc) return rc;
        return fhip_pooling(&poolq, bottoms[0]->n, tops[0]->data, pre_pool.d, s);
    }
    int Fuse(Layer* next, int level) override;
    // absorb `elt` = Eltwise SUM of this layer's top and `other` (a blob produced earlier in the layer list)
    bool FuseResidual(Blob* other)
    {
        if (pw)
        {
            // behind an absorbed 1x1 convolution (MobileNet-V2 style dw -> pw(linear) -> add): the add goes into the POINTWISE layer's
            // epilogue; the pair then runs its two kernels one after the other (Reshape: pair_fast needs a pointwise layer without residual)
            if (fuse_pool || residual || !pw->FuseResidual(other)) return false;
            pw->bottoms.pop_back();   // the pointwise layer's bottoms are set per call (a stand-in for the depthwise output)
            bottoms.push_back(other); // dependency scans look at THIS layer
            return true;
        }
        if (fuse_pool || residual || p.activation != FHIP_ACT_NONE) return false;
        residual = other;
        bottoms.push_back(other); // so that dependency scans (fusion, branch concurrency) see the second input
        return true;
    }
    size_t weight_bytes() const override
    {
        return packed.bytes + bias.bytes + pre_pool.bytes + mid.bytes + first_raw.bytes + sib_packed.bytes + sib_bias.bytes + (pw ? pw->weight_bytes() : 0);
    }
    const fhip_conv_param* fused_pointwise(int* one_kernel) const override
    {
        if (one_kernel) *one_kernel = pair_fast ? 1 : 0;
        return pw ? &pw->p : nullptr;
    }
    size_t arena_bytes() const override { return std::max(buffer_bytes, chain_bytes); }
    const fhip_conv_param* conv_param() const override { return &p; }
    int algo() const override { return algo_; }
    int sibling_state() const override { return sib ? 1 : sib_of ? 2 : 0; }
    int residual_state() const override { return !residual ? 0 : res_fast ? 1 : 2; }
    void chain_state(int* v_from_previous, int* writes_next_v) const override
    {
        *v_from_previous = head ? 2 : chain_in ? 1 : 0;
        *writes_next_v = head_of ? 2 : chain_next ? 1 : 0;
    }
};

// feather::InnerProductLayer, layers/inner_product_layer.h:28-171: y = W x + b, W [out][in].  On the device it is a
// 1x1 convolution over a 1x1 image with `in` channels, i.e. one GEMM [out x in] * [in x batch] through the implicit
// GEMM path (the reference's GEMV, booster/avx/sgemv.cpp:317-395, is the batch = 1 case).
struct InnerProductLayer : Layer
{
    fhip_conv_param p;
    size_t input_size = 0, output_size = 0, weight_data_size = 0;
    std::vector<float> w_host, b_host;
    DeviceVec packed, bias;
    size_t buffer_bytes = 0, packed_bytes = 0;
    bool inited = false;

    InnerProductLayer() { memset(&p, 0, sizeof(p)); }
    int LoadParam(const ParamDict& pd) override
    {
        output_size = pd.get(0, 0);
        p.bias_term = pd.get(1, 0);
        weight_data_size = pd.get(2, 0);
        if (output_size == 0) return failf(NET_E_SHAPE, "layer %s: num_output is 0", name.c_str());
        input_size = weight_data_size / output_size;
        p.input_channels = (int)input_size;
        p.output_channels = (int)output_size;
        p.input_h = p.input_w = p.kernel_h = p.kernel_w = p.stride_h = p.stride_w = p.group = 1;
        fhip_conv_assign_output_dim(&p);
        return 0;
    }
    int LoadWeights(ModelBin& mb) override
    {
        int rc = mb.load(weight_data_size, 0, w_host);
        if (rc) return rc;
        if (p.bias_term) rc = mb.load(output_size, 1, b_host);
        return rc;
    }
    int Reshape() override
    {
        const Blob* b = bottoms[0];
        const size_t per_image = (size_t)b->c * b->h * b->w;
        if (input_size != per_image)
            return failf(NET_E_SHAPE, "In Layer %s: Bottom %s data size %zu is inconsistant with expected input size %zu.", name.c_str(), b->name.c_str(), per_image, input_size);
        int rc = tops[0]->reshape(b->n, (int)output_size, 1, 1);
        if (rc) return rc;
        return fhip_conv_get_buffer_size(&p, FHIP_IM2COL, b->n, &buffer_bytes, &packed_bytes);
    }
    int Init(hipStream_t s) override
    {
        if (inited) return 0;
        DeviceVec raw;
        int rc = raw.upload(w_host.data(), w_host.size(), s);
        if (rc) return rc;
        rc = packed.resize(packed_bytes);
        if (rc) return rc;
        rc = fhip_conv_init(&p, FHIP_IM2COL, packed.d, raw.d, s);
        if (rc) return rc;
        if (p.bias_term)
        {
            rc = bias.upload(b_host.data(), b_host.size(), s);
            if (rc) return rc;
        }
        FHIP_CHECK_HIP(hipStreamSynchronize(s));
        std::vector<float>().swap(w_host); // the packed copy is shape independent: the raw weights are not needed again
        inited = true;
        return 0;
    }
    int Forward(hipStream_t s) override
    {
        return fhip_conv_forward(&p, FHIP_IM2COL, bottoms[0]->n, tops[0]->data, bottoms[0]->data, packed.d, (float*)net->arena.d,
                                 p.bias_term ? bias.d : nullptr, s);
    }
    int Fuse(Layer* next, int) override
    {
        if (next->type == "ReLU")
        {
            p.activation = FHIP_ACT_RELU;
            return 1;
        }
        return 0;
    }
    size_t weight_bytes() const override { return packed.bytes + bias.bytes; }
    size_t arena_bytes() const override { return buffer_bytes; }
    int algo() const override { return FHIP_IM2COL; }
};

struct ReluLayer : Layer
{
    int Forward(hipStream_t s) override { return fhip_relu(tops[0]->data, bottoms[0]->data, bottoms[0]->count(), s); }
};

struct PoolingLayer : Layer
{
    fhip_pool_param q;
    PoolingLayer() { memset(&q, 0, sizeof(q)); }
    int LoadParam(const ParamDict& pd) override // pooling_layer.h:90-107
    {
        q.pooling_type = pd.get(0, 0);
        q.kernel_w = pd.get(1, 0);
        q.kernel_h = pd.get(11, q.kernel_w);
        q.stride_w = pd.get(2, 1);
        q.stride_h = pd.get(12, q.stride_w);
        q.pad_left = pd.get(3, 0);
        q.pad_right = pd.get(14, q.pad_left);
        q.pad_top = pd.get(13, q.pad_left);
        q.pad_bottom = pd.get(15, q.pad_top);
        q.global_pooling = pd.get(4, 0) != 0;
        return 0;
    }
    int Reshape() override
    {
        const Blob* b = bottoms[0];
        q.channels = b->c;
        q.input_h = b->h;
        q.input_w = b->w;
        int oh, ow;
        int rc = fhip_pooling_output_dim(&q, &oh, &ow);
        if (rc) return rc;
        if (oh < 1 || ow < 1) return failf(NET_E_SHAPE, "layer %s: empty pooling output", name.c_str());
        return tops[0]->reshape(b->n, b->c, oh, ow);
    }
    int Forward(hipStream_t s) override { return fhip_pooling(&q, bottoms[0]->n, tops[0]->data, bottoms[0]->data, s); }
};

struct SoftmaxLayer : Layer
{
    int Forward(hipStream_t s) override
    {
        const Blob* b = bottoms[0];
        return fhip_softmax(tops[0]->data, b->data, b->n, b->c * b->h * b->w, s);
    }
};

// Per-channel affine layers: BatchNorm (batchnorm_layer.h:36-75) and Scale (scale_layer.h:33-98).  Both keep
// (mul, add) on the host until Init so that fusion can compose them: (x*m1 + a1)*m2 + a2.
struct AffineLayer : Layer
{
    int channels = 0;
    std::vector<float> mul, add;
    bool has_add = false, relu = false;
    DeviceVec d_mul, d_add;
    bool inited = false;

    void compose(const AffineLayer& nx)
    {
        if (!has_add) add.assign(channels, 0.f);
        for (int i = 0; i < channels; ++i)
        {
            mul[i] = mul[i] * nx.mul[i];
            add[i] = add[i] * nx.mul[i] + (nx.has_add ? nx.add[i] : 0.f);
        }
        has_add = has_add || nx.has_add;
    }
    int Reshape() override
    {
        if (bottoms[0]->c != channels)
            return failf(NET_E_SHAPE, "Mismatch channel in layer %s, expected %d but the bottom %s has %d channels.", name.c_str(), channels,
                         bottoms[0]->name.c_str(), bottoms[0]->c);
        return Layer::Reshape();
    }
    int Init(hipStream_t s) override
    {
        if (inited) return 0;
        int rc = d_mul.upload(mul.data(), mul.size(), s);
        if (rc) return rc;
        if (has_add) rc = d_add.upload(add.data(), add.size(), s);
        if (rc) return rc;
        FHIP_CHECK_HIP(hipStreamSynchronize(s));
        inited = true;
        return 0;
    }
    int Forward(hipStream_t s) override
    {
        const Blob* b = bottoms[0];
        return fhip_affine(tops[0]->data, b->data, d_mul.d, has_add ? d_add.d : nullptr, b->n, b->c, b->h * b->w, relu, s);
    }
    int Fuse(Layer* next, int) override
    {
        if (next->type == "ReLU")
        {
            relu = true;
            return 1;
        }
        if (!relu && type == "BatchNorm" && next->type == "Scale") // BN-Scale(-ReLU), batchnorm_layer.h:107-131
        {
            AffineLayer* nx = static_cast<AffineLayer*>(next);
            if (nx->channels != channels) return 0;
            compose(*nx);
            return 1;
        }
        return 0;
    }
    size_t weight_bytes() const override { return d_mul.bytes + d_add.bytes; }
};

struct BatchNormLayer : AffineLayer
{
    float eps = 0.f;
    int LoadParam(const ParamDict& pd) override
    {
        channels = pd.get(0, 0);
        eps = pd.get(1, 0.f);
        return 0;
    }
    int LoadWeights(ModelBin& mb) override // slope, mean, var, bias -> alpha/beta (batchnorm_layer.h:43-75)
    {
        std::vector<float> slope, mean, var, bias;
        int rc;
        if ((rc = mb.load(channels, 1, slope)) || (rc = mb.load(channels, 1, mean)) || (rc = mb.load(channels, 1, var)) || (rc = mb.load(channels, 1, bias))) return rc;
        mul.resize(channels);
        add.resize(channels);
        for (int i = 0; i < channels; ++i)
        {
            const float sqrt_var = sqrtf(var[i] + eps);
            add[i] = bias[i] - slope[i] * mean[i] / sqrt_var; // alpha
            mul[i] = slope[i] / sqrt_var;                     // beta
        }
        has_add = true;
        return 0;
    }
};

struct ScaleLayer : AffineLayer
{
    int bias_term = 0, scale_data_size = 0;
    int LoadParam(const ParamDict& pd) override
    {
        scale_data_size = pd.get(0, 0);
        bias_term = pd.get(1, 0);
        if (scale_data_size < 0) return failf(NET_E_SHAPE, "layer %s: negative scale data size is not accepted (scale_layer.h:37-41)", name.c_str());
        channels = scale_data_size;
        return 0;
    }
    int LoadWeights(ModelBin& mb) override
    {
        int rc = mb.load(scale_data_size, 1, mul);
        if (rc) return rc;
        if (bias_term)
        {
            rc = mb.load(scale_data_size, 1, add);
            has_add = true;
        }
        return rc;
    }
};

int ConvLayer::Fuse(Layer* next, int level)
{
    if (pw) return pw->Fuse(next, level) == 1 ? 1 : 0; // behind the absorbed 1x1 convolution: its own fusions
    if (fuse_pool) return 0; // nothing is absorbed behind the pooling
    if (level >= 2 && !residual && next->type == "Convolution" && p.group == p.input_channels && p.group > 1 && p.group <= 256 && p.kernel_h == 3 &&
        p.kernel_w == 3 && p.stride_h == p.stride_w && (p.stride_h == 1 || p.stride_h == 2) && p.pad_left == 1 && p.pad_top == 1)
    {
        // depthwise 3x3 (at most 256 channels: fhip_conv_can_fuse_dw_pw's structural conditions) -> 1x1 convolution: the pair becomes one
        // layer -- one kernel where the shapes qualify too, else the two kernels one after the other inside this layer
        const fhip_conv_param& q = static_cast<ConvLayer*>(next)->p;
        if (q.group != 1 || q.kernel_h != 1 || q.kernel_w != 1 || q.stride_h != 1 || q.stride_w != 1 || q.pad_left || q.pad_right || q.pad_top || q.pad_bottom)
            return 0;
        // fhip_conv_can_fuse_dw_pw's profitable range -- or the band-staged kernel's pair (32 channels, stride 1, a multiple of 64 output channels:
        // MobileNet's first pair); the row width is only known at Reshape, where the pair falls back to its two kernels if it does not qualify
        const bool band_pair = p.input_channels == 32 && p.stride_h == 1 && q.output_channels % 64 == 0 && q.output_channels <= 128; // = dwpw_band_applicable's bound
        if (!band_pair && (q.output_channels <= 64 || q.output_channels >= (p.stride_h == 1 ? 160 : 400))) return 0;
        return 2; // the pass hands `next` over (fuse_layers)
    }
    if (residual && next->type != "ReLU") return 0; // behind the residual add only its ReLU
    if (level >= 2 && next->type == "Pooling")
    {
        const fhip_pool_param& q = static_cast<PoolingLayer*>(next)->q;
        if (q.pooling_type != 0 || q.global_pooling || q.kernel_h != 2 || q.kernel_w != 2 || q.stride_h != 2 || q.stride_w != 2 || q.pad_left ||
            q.pad_right || q.pad_top || q.pad_bottom)
            return 0;
        poolq = q;
        fuse_pool = true;
        return 1;
    }
    if (next->type == "ReLU")
    {
        p.activation = FHIP_ACT_RELU;
        return 1;
    }
    // level 2 (beyond the reference): fold a following BatchNorm / Scale into the weights and bias
    if (level >= 2 && p.activation == FHIP_ACT_NONE && (next->type == "BatchNorm" || next->type == "Scale"))
    {
        AffineLayer* nx = static_cast<AffineLayer*>(next);
        const int K = p.group == p.input_channels ? p.input_channels : p.output_channels;
        if (nx->channels != K || nx->relu) return 0;
        if (post_mul.empty())
        {
            post_mul.assign(K, 1.f);
            post_add.assign(K, 0.f);
        }
        for (int k = 0; k < K; ++k)
        {
            post_mul[k] *= nx->mul[k];
            post_add[k] = post_add[k] * nx->mul[k] + (nx->has_add ? nx->add[k] : 0.f);
        }
        return 1;
    }
    return 0;
}

struct EltwiseLayer : Layer
{
    bool relu = false;
    int LoadParam(const ParamDict& pd) override // eltwise_layer.h:53-68
    {
        if (pd.has_array(1)) return failf(NET_E_SHAPE, "layer %s: coeffs in eltwise layer are not supported", name.c_str());
        if (pd.get(0, 0) != 1) return failf(NET_E_SHAPE, "layer %s: only eltwise SUM is supported", name.c_str());
        return 0;
    }
    int Reshape() override
    {
        if (bottoms.size() < 2) return failf(NET_E_TOPOLOGY, "layer %s: eltwise needs two bottoms", name.c_str());
        const Blob* a = bottoms[0];
        for (size_t i = 1; i < bottoms.size(); ++i)
            if (bottoms[i]->n != a->n || bottoms[i]->c != a->c || bottoms[i]->h != a->h || bottoms[i]->w != a->w)
                return failf(NET_E_SHAPE, "Shape mismatch among bottoms of layer %s.", name.c_str());
        for (Blob* t : tops)
        {
            int rc = t->reshape(a->n, a->c, a->h, a->w);
            if (rc) return rc;
        }
        return 0;
    }
    int Forward(hipStream_t s) override // the reference adds bottoms 0 and 1 only (eltwise_layer.h:71-79)
    {
        return fhip_add(tops[0]->data, bottoms[0]->data, bottoms[1]->data, bottoms[0]->count(), relu, s);
    }
    int Fuse(Layer* next, int) override
    {
        if (next->type == "ReLU")
        {
            relu = true;
            return 1;
        }
        return 0;
    }
};

struct ConcatLayer : Layer
{
    int axis = 0;
    int LoadParam(const ParamDict& pd) override
    {
        axis = pd.get(0, 0);
        return 0;
    }
    int Reshape() override // concat_layer.h:50-80
    {
        if (axis != 0) return failf(NET_E_SHAPE, "layer %s: only concat at axis = 0 (channels) is supported", name.c_str());
        const Blob* a = bottoms[0];
        int channels = a->c;
        for (size_t i = 1; i < bottoms.size(); ++i)
        {
            if (bottoms[i]->w != a->w || bottoms[i]->h != a->h || bottoms[i]->n != a->n)
                return failf(NET_E_SHAPE, "layer %s: images of different shapes cannot be concatenated together", name.c_str());
            channels += bottoms[i]->c;
        }
        return tops[0]->reshape(a->n, channels, a->h, a->w);
    }
    int Forward(hipStream_t s) override
    {
        Blob* t = tops[0];
        const size_t hw = (size_t)t->h * t->w;
        size_t c_off = 0;
        for (Blob* b : bottoms)
        {
            const size_t row = (size_t)b->c * hw * sizeof(float);
            FHIP_CHECK_HIP(hipMemcpy2DAsync(t->data + c_off * hw, (size_t)t->c * hw * sizeof(float), b->data, row, row, b->n, hipMemcpyDeviceToDevice, s));
            c_off += b->c;
        }
        return 0;
    }
};

struct SplitLayer : Layer
{
    int Reshape() override
    {
        for (Blob* t : tops) t->share(bottoms[0]);
        return 0;
    }
    int Forward(hipStream_t) override { return 0; } // tops alias the bottom (the reference memcpy's, split_layer.h:43-52)
};

struct DropoutLayer : Layer
{
    float scale = 1.f;
    DeviceVec d_scale;
    int LoadParam(const ParamDict& pd) override
    {
        scale = pd.get(0, 1.f);
        return 0;
    }
    int Reshape() override
    {
        if (scale == 1.f)
        {
            tops[0]->share(bottoms[0]);
            return 0;
        }
        return Layer::Reshape();
    }
    int Init(hipStream_t s) override
    {
        if (scale == 1.f || d_scale.d) return 0;
        std::vector<float> v(bottoms[0]->c, scale);
        int rc = d_scale.upload(v.data(), v.size(), s);
        if (rc) return rc;
        FHIP_CHECK_HIP(hipStreamSynchronize(s));
        return 0;
    }
    int Forward(hipStream_t s) override
    {
        if (scale == 1.f) return 0;
        const Blob* b = bottoms[0];
        return fhip_affine(tops[0]->data, b->data, d_scale.d, nullptr, b->n, b->c, b->h * b->w, 0, s);
    }
};

static Layer* create_layer(const std::string& type) // layer_factory.cpp:55-67
{
    if (type == "Input") return new InputLayer;
    if (type == "Convolution" || type == "ConvolutionDepthWise") return new ConvLayer;
    if (type == "ReLU") return new ReluLayer;
    if (type == "Pooling") return new PoolingLayer;
    if (type == "InnerProduct") return new InnerProductLayer;
    if (type == "Dropout") return new DropoutLayer;
    if (type == "Softmax") return new SoftmaxLayer;
    if (type == "BatchNorm") return new BatchNormLayer;
    if (type == "Scale") return new ScaleLayer;
    if (type == "Split") return new SplitLayer;
    if (type == "Eltwise") return new EltwiseLayer;
    if (type == "Concat") return new ConcatLayer;
    return nullptr;
}

// ---- Net::LoadParam (net.cpp:67-170) over a token stream: fscanf("%s") semantics ------------------------------------
static int load_param_text(Net& net, const char* text, size_t len)
{
    if (net.param_loaded) return failf(NET_E_IO, "a param file is already loaded");
    std::vector<std::string> tok;
    {
        size_t i = 0;
        while (i < len)
        {
            while (i < len && isspace((unsigned char)text[i])) ++i;
            const size_t a = i;
            while (i < len && !isspace((unsigned char)text[i])) ++i;
            if (i > a) tok.emplace_back(text + a, i - a);
        }
    }
    size_t t = 0;
    auto next = [&](std::string& out) {
        if (t >= tok.size()) return false;
        out = tok[t++];
        return true;
    };
    std::string s;
    if (!next(s)) return failf(NET_E_IO, "issue with param file");
    if (atoi(s.c_str()) != 7767517) return failf(NET_E_IO, "param is too old, please regenerate"); // utils.cpp:27-44
    std::string a, b;
    if (!next(a) || !next(b)) return failf(NET_E_IO, "issue with param file");
    const int layer_count = atoi(a.c_str()), blob_count = atoi(b.c_str());
    if (layer_count <= 0 || blob_count <= 0) return failf(NET_E_IO, "issue with param file");

    ParamDict pd;
    for (int i = 0; i < layer_count; ++i)
    {
        std::string type, name, nb, nt;
        if (!next(type) || !next(name) || !next(nb) || !next(nt)) return failf(NET_E_IO, "param file ends after %d of %d layers", i, layer_count);
        std::unique_ptr<Layer> layer(create_layer(type));
        if (!layer) return failf(NET_E_UNKNOWN_LAYER, "layer %s not exists or registered", type.c_str());
        layer->type = type;
        layer->name = name;
        layer->net = &net;
        const int bottom_count = atoi(nb.c_str()), top_count = atoi(nt.c_str());
        for (int j = 0; j < bottom_count; ++j)
        {
            std::string bn;
            if (!next(bn)) return failf(NET_E_IO, "param file truncated in layer %s", name.c_str());
            Blob* blob = net.find(bn);
            if (!blob) return failf(NET_E_TOPOLOGY, "Topology error: bottom blob %s of layer %s type %s not found in map.", bn.c_str(), name.c_str(), type.c_str());
            layer->bottoms.push_back(blob);
        }
        for (int j = 0; j < top_count; ++j)
        {
            std::string tn;
            if (!next(tn)) return failf(NET_E_IO, "param file truncated in layer %s", name.c_str());
            std::unique_ptr<Blob> blob(new Blob);
            blob->name = tn;
            layer->tops.push_back(blob.get());
            auto& slot = net.blobs[tn];
            if (slot) net.shadowed.push_back(std::move(slot)); // earlier layers (and this layer's bottoms) still point at it
            slot = std::move(blob);
        }
        pd.clear();
        while (t < tok.size() && ParamDict::looks_like_pair(tok[t]))
        {
            const int rc = pd.parse(tok[t++]);
            if (rc) return rc;
        }
        const int rc = layer->LoadParam(pd);
        if (rc) return rc;
        net.layers.push_back(std::move(layer));
    }
    net.param_loaded = true;
    return 0;
}

static int load_weights_mem(Net& net, const void* data, size_t len)
{
    if (net.initialized) return failf(NET_E_IO, "Net is already initialized. Are you repeatedly loading models?");
    if (net.layers.empty()) return failf(NET_E_IO, "Network has not been loaded. Please load the param file first.");
    ModelBin mb{(const unsigned char*)data, (const unsigned char*)data + len};
    for (auto& l : net.layers)
    {
        const int rc = l->LoadWeights(mb);
        if (rc)
        {
            const std::string why = fhip_last_error();
            return failf(NET_E_IO, "Layer %s loading weights failed: %s", l->name.c_str(), why.c_str());
        }
    }
    net.weights_loaded = true;
    return 0;
}

static int read_file(const char* path, std::vector<char>& out)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) return failf(NET_E_IO, "Cannot open file, path: %s", path);
    fseek(fp, 0, SEEK_END);
    const long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    out.resize(sz > 0 ? (size_t)sz : 0);
    const size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), fp);
    fclose(fp);
    if (got != out.size()) return failf(NET_E_IO, "short read on %s", path);
    return 0;
}

// The fusion pass of layer.cpp:82-101 (TryFuse): a layer absorbs the single consumer of its single top.
static void fuse_layers(Net& net)
{
    if (net.fused) return;
    net.fused = true;
    if (net.fusion <= 0) return;
    for (size_t i = 0; i < net.layers.size(); ++i)
    {
        for (;;)
        {
            Layer* cur = net.layers[i].get();
            if (cur->tops.size() != 1) break;
            Blob* top = cur->tops[0];
            size_t consumer = 0;
            int uses = 0;
            for (size_t j = i + 1; j < net.layers.size(); ++j)
                for (Blob* b : net.layers[j]->bottoms)
                    if (b == top)
                    {
                        ++uses;
                        consumer = j;
                    }
            if (uses != 1) break;
            Layer* nx = net.layers[consumer].get();
            if (net.fusion >= 2 && nx->type == "Eltwise" && nx->bottoms.size() == 2 && nx->tops.size() == 1 && nx->bottoms[0] != nx->bottoms[1] &&
                (cur->type == "Convolution" || cur->type == "ConvolutionDepthWise"))
            {
                // conv -> Eltwise(sum with an EARLIER blob) [-> ReLU]: the add moves into the conv's epilogue
                Blob* other = nx->bottoms[0] == top ? nx->bottoms[1] : nx->bottoms[0];
                bool earlier = false;
                for (size_t j = 0; j < i; ++j)
                    for (Blob* t : net.layers[j]->tops) earlier = earlier || t == other;
                if (!earlier || !static_cast<ConvLayer*>(cur)->FuseResidual(other)) break;
                top->fused_away = true;
                cur->tops[0] = nx->tops[0];
                net.layers.erase(net.layers.begin() + consumer);
                continue;
            }
            if (nx->bottoms.size() != 1 || nx->tops.size() != 1) break;
            const int fr = cur->Fuse(nx, net.fusion);
            if (fr != 1 && fr != 2) break;
            if (fr == 2) static_cast<ConvLayer*>(cur)->pw.reset(static_cast<ConvLayer*>(net.layers[consumer].release())); // adopted, not dropped
            top->fused_away = true;
            cur->tops[0] = nx->tops[0];
            net.layers.erase(net.layers.begin() + consumer);
        }
    }
}

// Which layers may leave the main stream (see Net::concurrency).  Needs the algorithms chosen by Reshape: only layers without
// scratch (the arena is shared) qualify.
static int plan_concurrency(Net& net)
{
    const size_t L = net.layers.size();
    net.side.assign(L, -1);
    net.waits.assign(L, std::vector<int>());
    if (!net.concurrency) return 0;
    int pairs = 0;
    for (size_t i = 0; i < L; ++i)
    {
        Layer* l = net.layers[i].get();
        if ((l->type != "Convolution" && l->type != "ConvolutionDepthWise") || l->arena_bytes() != 0 || l->tops.size() != 1) continue;
        if (l->sibling_state()) continue; // writes (or is written with) the top of its neighbour: stays on the net's stream
        size_t consumer = 0;
        int uses = 0;
        for (size_t j = i + 1; j < L; ++j)
            for (Blob* b : net.layers[j]->bottoms)
                if (b == l->tops[0] || b->alias == l->tops[0])
                {
                    ++uses;
                    consumer = j;
                }
        if (uses != 1 || consumer <= i + 1) continue;
        bool work_between = false; // something worth overlapping with
        for (size_t j = i + 1; j < consumer; ++j) work_between = work_between || net.layers[j]->algo() >= 0;
        if (!work_between) continue;
        net.side[i] = pairs;
        net.waits[consumer].push_back(pairs);
        ++pairs;
    }
    if (pairs && !net.side_stream) FHIP_CHECK_HIP(hipStreamCreateWithFlags(&net.side_stream, hipStreamNonBlocking));
    while (net.events.size() < (size_t)2 * pairs)
    {
        hipEvent_t e;
        FHIP_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        net.events.push_back(e);
    }
    return 0;
}

// Fusion level 3, after Reshape (needs the routes and shapes): a Winograd layer directly followed by the 3x3 / stride-1 / pad-1 Winograd
// layer that is the only consumer of its top hands over its output already transformed (fhip_conv_forward_chained); the blob between
// them gets no storage.  VGG-16: conv1_2 ... conv5_3 become one run.
static int plan_chains(Net& net)
{
    const size_t L = net.layers.size();
    std::vector<ConvLayer*> conv(L, nullptr);
    for (size_t i = 0; i < L; ++i)
        if (net.layers[i]->type == "Convolution" || net.layers[i]->type == "ConvolutionDepthWise")
        {
            conv[i] = static_cast<ConvLayer*>(net.layers[i].get());
            conv[i]->chain_next = nullptr;
            conv[i]->chain_in = false;
            conv[i]->chain_pos = 0;
            conv[i]->chain_bytes = 0;
            conv[i]->head = conv[i]->head_of = nullptr;
        }
    for (auto& kv : net.blobs) kv.second->chained = false;
    for (auto& b : net.shadowed) b->chained = false; // blobs whose name a later top re-used (in-place style .param files) re-plan too
    net.chain_slot[0] = net.chain_slot[1] = net.chain_m = 0;
    if (net.fusion < 3) return 0;
    auto plain = [](const ConvLayer* c) { return c && !c->pw && !c->residual && c->algo_ == FHIP_WINOGRADF63 && c->tops.size() == 1 && c->bottoms.size() == 1; };
    for (size_t i = 0; i + 1 < L; ++i)
    {
        ConvLayer *a = conv[i], *b = conv[i + 1];
        if (!plain(a) || !plain(b) || b->bottoms[0] != a->tops[0]) continue;
        if (a->fuse_pool && !a->pool_fast) continue;
        int uses = 0;
        for (size_t j = 0; j < L; ++j)
            for (Blob* x : net.layers[j]->bottoms) uses += (x == a->tops[0] || x->alias == a->tops[0]) ? 1 : 0;
        if (uses != 1) continue;
        if (!fhip_conv_can_chain_winograd(&a->p, a->algo_, &b->p, b->algo_, a->fuse_pool ? 1 : 0)) continue;
        a->chain_next = b;
        b->chain_in = true;
        b->chain_pos = a->chain_pos + 1;
        a->tops[0]->chained = true;
        a->tops[0]->drop_storage();
    }
    // a first layer in front of a Winograd layer (VGG-16: conv1_1 -> conv1_2) is computed inside that layer's input transform
    for (size_t i = 0; i + 1 < L; ++i)
    {
        ConvLayer *a = conv[i], *b = conv[i + 1];
        if (!a || a->pw || a->residual || a->fuse_pool || a->chain_in || a->chain_next || a->tops.size() != 1 || a->bottoms.size() != 1) continue;
        if (!a->first_candidate() || !plain(b) || b->chain_in || b->bottoms[0] != a->tops[0]) continue;
        int uses = 0;
        for (size_t j = 0; j < L; ++j)
            for (Blob* x : net.layers[j]->bottoms) uses += (x == a->tops[0] || x->alias == a->tops[0]) ? 1 : 0;
        if (uses != 1) continue;
        if (!fhip_conv_can_fuse_first_winograd(&a->p, &b->p, b->algo_, a->bottoms[0]->n)) continue;
        a->head_of = b;
        b->head = a;
        a->tops[0]->chained = true;
        a->tops[0]->drop_storage();
    }
    // blobs the previous plan had chained and this one did not: Reshape left them without storage
    auto restore = [](Blob* b) -> int {
        if (b->chained || b->data || b->alias || b->fused_away || b->count() == 0) return 0;
        FHIP_CHECK_HIP(hipMalloc((void**)&b->data, b->count() * sizeof(float)));
        b->capacity = b->count();
        return 0;
    };
    for (auto& kv : net.blobs)
        if (const int rc = restore(kv.second.get())) return rc;
    for (auto& b : net.shadowed)
        if (const int rc = restore(b.get())) return rc;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t slot[2] = {0, 0}, msz = 0;
    for (size_t i = 0; i < L; ++i)
    {
        ConvLayer* c = conv[i];
        if (!c || (!c->chain_in && !c->chain_next && !c->head)) continue;
        fhip_winograd_plan pl;
        const int rc = fhip_winograd_f63_plan(&c->p, c->bottoms[0]->n, &pl);
        if (rc) return rc;
        slot[c->chain_pos & 1] = std::max(slot[c->chain_pos & 1], up(pl.v_bytes));
        msz = std::max(msz, up(pl.m_bytes));
    }
    net.chain_slot[0] = 0;
    net.chain_slot[1] = slot[0];
    net.chain_m = slot[0] + slot[1];
    for (size_t i = 0; i < L; ++i)
        if (conv[i] && (conv[i]->chain_in || conv[i]->chain_next || conv[i]->head)) conv[i]->chain_bytes = slot[0] + slot[1] + msz;
    return 0;
}

// Fusion level 2, after Reshape (needs routes, shapes and the batch): two 1x1 convolutions that follow each other in the layer list and
// read the same blob with the same stride run as one GEMM (fhip_conv_forward_siblings).  ResNet-50: res3a / res4a / res5a branch1 + branch2a.
static int plan_siblings(Net& net)
{
    const size_t L = net.layers.size();
    std::vector<ConvLayer*> conv(L, nullptr);
    for (size_t i = 0; i < L; ++i)
        if (net.layers[i]->type == "Convolution")
        {
            conv[i] = static_cast<ConvLayer*>(net.layers[i].get());
            conv[i]->sib = conv[i]->sib_of = nullptr;
        }
    if (net.fusion < 2) return 0;
    auto root = [](const Blob* b) { return b->alias ? b->alias : b; };
    auto plain = [](const ConvLayer* c) {
        return c && !c->pw && !c->residual && !c->fuse_pool && !c->chain_in && !c->chain_next && !c->head && !c->head_of && !c->sib && !c->sib_of &&
               c->algo_ == FHIP_IM2COL && c->tops.size() == 1 && c->bottoms.size() == 1;
    };
    for (size_t i = 0; i + 1 < L; ++i)
    {
        ConvLayer *a = conv[i], *b = conv[i + 1];
        if (!plain(a) || !plain(b) || root(a->bottoms[0]) != root(b->bottoms[0]) || a->tops[0] == b->tops[0]) continue;
        // both layers on the 128-row tile when alone (a 64-row layer -- ResNet's res2a_branch2a -- is faster on its own 64 x 128 tile:
        // tools/sibling_bench.py 116 vs 126 us for the pair)
        if (a->p.output_channels <= 64 || b->p.output_channels <= 64) continue;
        if (!fhip_conv_can_fuse_siblings(&a->p, a->algo_, &b->p, b->algo_, a->bottoms[0]->n)) continue;
        a->sib = b;
        b->sib_of = a;
    }
    // a pair that a new shape un-planned keeps no stacked copy of its filters (the layers' own packed weights stay -- they are what
    // the pair falls back to, and what Extract-driven level changes run)
    for (size_t i = 0; i < L; ++i)
        if (conv[i] && !conv[i]->sib && conv[i]->sib_packed.bytes)
        {
            conv[i]->sib_packed.release();
            conv[i]->sib_bias.release();
            conv[i]->sib_packed_for = nullptr;
        }
    return 0;
}

static int reshape_all(Net& net)
{
    net.drop_graph();
    size_t need = 0;
    for (auto& l : net.layers)
    {
        const int rc = l->Reshape();
        if (rc) return rc;
    }
    {
        int rc = plan_chains(net);
        if (rc) return rc;
        if ((rc = plan_siblings(net))) return rc;
    }
    for (auto& l : net.layers) need = std::max(need, l->arena_bytes());
    // one scratch arena shared by every layer = max over layers (mempool.cpp:88-92)
    if (need > net.arena.bytes)
    {
        const int rc = net.arena.resize(need);
        if (rc) return rc;
    }
    net.shapes_dirty = false;
    return plan_concurrency(net);
}

static int prepare(Net& net)
{
    if (!net.param_loaded) return failf(NET_E_IO, "Network has not been loaded. Please load the param file first.");
    if (!net.weights_loaded)
    {
        bool needs = false;
        for (auto& l : net.layers) needs = needs || (l->type != "Input" && l->type != "ReLU" && l->type != "Pooling" && l->type != "Softmax" &&
                                                     l->type != "Split" && l->type != "Eltwise" && l->type != "Concat" && l->type != "Dropout");
        if (needs) return failf(NET_E_IO, "weights have not been loaded");
    }
    fuse_layers(net);
    for (auto& kv : net.blobs)
        if (!kv.second->fused_away && kv.second->count() == 0)
        {
            bool is_input = false;
            for (auto& l : net.layers)
                if (l->type == "Input")
                    for (Blob* tb : l->tops) is_input = is_input || tb == kv.second.get();
            if (is_input) return failf(NET_E_SHAPE, "input blob %s has not been fed", kv.first.c_str());
        }
    if (net.shapes_dirty)
    {
        const int rc = reshape_all(net);
        if (rc) return rc;
    }
    for (auto& l : net.layers) // Init is a no-op for a layer whose packed weights are still valid
    {
        const int rc = l->Init(net.stream);
        if (rc) return rc;
    }
    net.initialized = true;
    return 0;
}

static int run_layers(Net& net)
{
    for (size_t i = 0; i < net.layers.size(); ++i)
    {
        Layer* l = net.layers[i].get();
        if (i < net.side.size() && net.side[i] >= 0)
        {
            hipEvent_t fork = net.events[2 * net.side[i]], join = net.events[2 * net.side[i] + 1];
            FHIP_CHECK_HIP(hipEventRecord(fork, net.stream)); // everything this layer reads is produced earlier on the main stream
            FHIP_CHECK_HIP(hipStreamWaitEvent(net.side_stream, fork, 0));
            const int rc = l->Forward(net.side_stream);
            if (rc) return rc;
            FHIP_CHECK_HIP(hipEventRecord(join, net.side_stream));
            continue;
        }
        if (i < net.waits.size())
            for (int pair : net.waits[i]) FHIP_CHECK_HIP(hipStreamWaitEvent(net.stream, net.events[2 * pair + 1], 0));
        const int rc = l->Forward(net.stream);
        if (rc) return rc;
    }
    return 0;
}

static int forward(Net& net)
{
    int rc = prepare(net);
    if (rc) return rc;
    if (!net.use_graph) return run_layers(net);
    if (!net.graph_exec)
    {
        hipGraph_t graph = nullptr;
        FHIP_CHECK_HIP(hipStreamBeginCapture(net.stream, hipStreamCaptureModeThreadLocal));
        rc = run_layers(net);
        const hipError_t e = hipStreamEndCapture(net.stream, &graph);
        if (rc)
        {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        FHIP_CHECK_HIP(e);
        const hipError_t ei = hipGraphInstantiate(&net.graph_exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        FHIP_CHECK_HIP(ei);
    }
    FHIP_CHECK_HIP(hipGraphLaunch(net.graph_exec, net.stream));
    return 0;
}

} // namespace net
} // namespace fhip

using namespace fhip;
using namespace fhip::net;

struct fhip_net
{
    Net impl; // the net -- or, with fhip_net_set_sub_batches(R > 1), the replica that takes the first share of every batch
    // Sub-batch replicas: R - 1 further complete nets (own layers, weights, blobs, arena, stream, graph).  FeedInput deals the images
    // out in contiguous shares, Forward runs all replicas concurrently (fork / join events against impl.stream), Extract puts the
    // shares back together.  Images are independent, so no replica ever reads another's data.
    std::vector<std::unique_ptr<fhip_net>> more;
    std::vector<hipStream_t> more_streams;
    hipEvent_t fork = nullptr;
    std::vector<hipEvent_t> joins;
    std::vector<int> share;                    // images per replica at the last FeedInput (index 0 = impl)
    std::map<std::string, DeviceVec> gathered; // Extract: blobs put back together (device pointer API)
    ~fhip_net()
    {
        more.clear();
        for (hipStream_t st : more_streams) (void)hipStreamDestroy(st);
        if (fork) (void)hipEventDestroy(fork);
        for (hipEvent_t e : joins) (void)hipEventDestroy(e);
    }
};

// contiguous shares of `num` images for `parts` replicas, the remainder dealt to the first ones (feathercnn_amd/shard.py's rule)
static void deal(int num, int parts, std::vector<int>& share)
{
    share.assign(parts, num / parts);
    for (int r = 0; r < num % parts; ++r) ++share[r];
}

#define NET_GUARD(n) \
    if (!(n)) return fail(FHIP_E_BADARG, "null net")

extern "C"
{

int fhip_net_create(fhip_net** out)
{
    if (!out) return fail(FHIP_E_BADARG, "null argument");
    *out = new fhip_net;
    return FHIP_OK;
}

int fhip_net_destroy(fhip_net* n)
{
    NET_GUARD(n);
    for (auto& m : n->more)
        if (m->impl.stream) (void)hipStreamSynchronize(m->impl.stream);
    if (n->impl.stream || n->impl.initialized) (void)hipStreamSynchronize(n->impl.stream);
    delete n;
    return FHIP_OK;
}

int fhip_net_set_sub_batches(fhip_net* n, int replicas)
{
    NET_GUARD(n);
    if (replicas < 1 || replicas > 16) return fail(FHIP_E_BADARG, "1 .. 16 sub-batches");
    if (n->impl.param_loaded || !n->more.empty()) return fail(FHIP_E_BADARG, "set the number of sub-batches once, before LoadParam");
    for (int r = 1; r < replicas; ++r)
    {
        hipStream_t st = nullptr;
        FHIP_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        n->more_streams.push_back(st);
        hipEvent_t e = nullptr;
        FHIP_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        n->joins.push_back(e);
        std::unique_ptr<fhip_net> m(new fhip_net);
        m->impl.stream = st;
        m->impl.fusion = n->impl.fusion;
        m->impl.use_graph = n->impl.use_graph;
        m->impl.tuned_selection = n->impl.tuned_selection;
        m->impl.concurrency = n->impl.concurrency;
        n->more.push_back(std::move(m));
    }
    if (replicas > 1 && !n->fork) FHIP_CHECK_HIP(hipEventCreateWithFlags(&n->fork, hipEventDisableTiming));
    return FHIP_OK;
}

int fhip_net_set_stream(fhip_net* n, void* stream)
{
    NET_GUARD(n);
    n->impl.stream = (hipStream_t)stream;
    n->impl.drop_graph();
    return FHIP_OK;
}

int fhip_net_set_fusion(fhip_net* n, int on)
{
    NET_GUARD(n);
    if (n->impl.fused) return fail(FHIP_E_BADARG, "fusion already ran; set it before the first Forward");
    n->impl.fusion = on;
    for (auto& m : n->more) m->impl.fusion = on;
    return FHIP_OK;
}

int fhip_net_set_tuned_selection(fhip_net* n, int on)
{
    NET_GUARD(n);
    n->impl.tuned_selection = on != 0;
    n->impl.shapes_dirty = true;
    for (auto& m : n->more) (void)fhip_net_set_tuned_selection(m.get(), on);
    return FHIP_OK;
}

int fhip_net_set_concurrency(fhip_net* n, int on)
{
    NET_GUARD(n);
    n->impl.concurrency = on != 0;
    n->impl.shapes_dirty = true; // re-plan (and drop a captured graph) at the next Forward
    for (auto& m : n->more) (void)fhip_net_set_concurrency(m.get(), on);
    return FHIP_OK;
}

int fhip_net_set_graph(fhip_net* n, int on)
{
    NET_GUARD(n);
    n->impl.use_graph = on != 0;
    if (!on) n->impl.drop_graph();
    if (on && !n->impl.stream)
    {
        if (!n->impl.owned_stream) FHIP_CHECK_HIP(hipStreamCreateWithFlags(&n->impl.owned_stream, hipStreamNonBlocking));
        n->impl.stream = n->impl.owned_stream;
    }
    for (auto& m : n->more) (void)fhip_net_set_graph(m.get(), on); // the replicas already run on streams of their own
    return FHIP_OK;
}

int fhip_net_load_param_mem(fhip_net* n, const char* text, size_t len)
{
    NET_GUARD(n);
    if (!text) return fail(FHIP_E_BADARG, "null text");
    for (auto& m : n->more)
    {
        const int rc = load_param_text(m->impl, text, len);
        if (rc) return rc;
    }
    return load_param_text(n->impl, text, len);
}

int fhip_net_load_param(fhip_net* n, const char* path)
{
    NET_GUARD(n);
    if (!path) return fail(FHIP_E_BADARG, "null path");
    std::vector<char> buf;
    const int rc = read_file(path, buf);
    if (rc) return rc;
    return fhip_net_load_param_mem(n, buf.data(), buf.size());
}

int fhip_net_load_weights_mem(fhip_net* n, const void* data, size_t len)
{
    NET_GUARD(n);
    if (!data && len) return fail(FHIP_E_BADARG, "null data");
    for (auto& m : n->more)
    {
        const int rc = load_weights_mem(m->impl, data, len);
        if (rc) return rc;
    }
    return load_weights_mem(n->impl, data, len);
}

// The .bin image handed over in DEVICE memory -- the buffer an RCCL broadcast (ncclBroadcast of the model from rank 0)
// delivered on this rank's GPU.  The image is staged through host memory once, because the loaders keep the raw weights on the host
// until Init (BatchNorm / Scale folding into the convolution weights happens there); the caller keeps ownership of `device_data` and
// may free it when the call returns.
int fhip_net_load_weights_device(fhip_net* n, const void* device_data, size_t len)
{
    NET_GUARD(n);
    if (!device_data && len) return fail(FHIP_E_BADARG, "null data");
    std::vector<char> host(len);
    if (len) FHIP_CHECK_HIP(hipMemcpy(host.data(), device_data, len, hipMemcpyDeviceToHost));
    return fhip_net_load_weights_mem(n, host.data(), len);
}

int fhip_net_load_weights(fhip_net* n, const char* path)
{
    NET_GUARD(n);
    if (!path) return fail(FHIP_E_BADARG, "null path");
    std::vector<char> buf;
    const int rc = read_file(path, buf);
    if (rc) return rc;
    return fhip_net_load_weights_mem(n, buf.data(), buf.size());
}

int fhip_net_feed_input(fhip_net* n, const char* blob_name, int num, int c, int h, int w, const float* data, int on_device)
{
    NET_GUARD(n);
    if (!blob_name || !data || num < 1 || c < 1 || h < 1 || w < 1) return fail(FHIP_E_BADARG, "bad argument");
    if (!n->more.empty())
    {
        // deal the images out; a replica left without images (batch < replicas) sits the forward out
        deal(num, (int)n->more.size() + 1, n->share);
        if (on_device) FHIP_CHECK_HIP(hipEventRecord(n->fork, n->impl.stream)); // the caller's data is ordered on the net's stream
        const size_t chw = (size_t)c * h * w;
        size_t first = n->share[0];
        for (size_t r = 0; r < n->more.size(); ++r)
        {
            const int cnt = n->share[r + 1];
            if (cnt > 0)
            {
                if (on_device) FHIP_CHECK_HIP(hipStreamWaitEvent(n->more[r]->impl.stream, n->fork, 0));
                const int rc = fhip_net_feed_input(n->more[r].get(), blob_name, cnt, c, h, w, data + first * chw, on_device);
                if (rc) return rc;
                // join the replica's copy back into the net's stream: whatever the caller enqueues there next (re-using or overwriting
                // the source buffer, a synchronize of the net's stream only) is ordered behind it -- "the caller's stream order is
                // unchanged" holds for FeedInput too
                FHIP_CHECK_HIP(hipEventRecord(n->joins[r], n->more[r]->impl.stream));
                FHIP_CHECK_HIP(hipStreamWaitEvent(n->impl.stream, n->joins[r], 0));
            }
            first += cnt;
        }
        num = n->share[0];
    }
    Blob* b = n->impl.find(blob_name);
    if (!b) return failf(NET_E_IO, "Invalid input blob %s, not found in map.", blob_name);
    if (b->n != num || b->c != c || b->h != h || b->w != w)
    {
        const float* old = b->data;
        const int rc = b->reshape(num, c, h, w);
        if (rc) return rc;
        n->impl.shapes_dirty = true;
        if (old != b->data) n->impl.drop_graph();
    }
    FHIP_CHECK_HIP(hipMemcpyAsync(b->data, data, b->count() * sizeof(float), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, n->impl.stream));
    return FHIP_OK;
}

int fhip_net_forward(fhip_net* n)
{
    NET_GUARD(n);
    if (n->more.empty()) return forward(n->impl);
    if (n->share.empty()) return failf(NET_E_SHAPE, "no input has been fed");
    // fork: whatever the caller enqueued on the net's stream happens before every replica's forward; join: whatever the caller
    // enqueues next (Extract, the next FeedInput) happens after all of them
    FHIP_CHECK_HIP(hipEventRecord(n->fork, n->impl.stream));
    for (size_t r = 0; r < n->more.size(); ++r)
    {
        if (n->share[r + 1] < 1) continue;
        Net& m = n->more[r]->impl;
        FHIP_CHECK_HIP(hipStreamWaitEvent(m.stream, n->fork, 0));
        const int rc = forward(m);
        if (rc) return rc;
        FHIP_CHECK_HIP(hipEventRecord(n->joins[r], m.stream));
    }
    const int rc = forward(n->impl);
    if (rc) return rc;
    for (size_t r = 0; r < n->more.size(); ++r)
        if (n->share[r + 1] > 0) FHIP_CHECK_HIP(hipStreamWaitEvent(n->impl.stream, n->joins[r], 0));
    return FHIP_OK;
}

int fhip_net_extract(fhip_net* n, const char* blob_name, float** ptr, int* num, int* c, int* h, int* w)
{
    NET_GUARD(n);
    if (!blob_name || !ptr) return fail(FHIP_E_BADARG, "null argument");
    Blob* b = n->impl.find(blob_name);
    if (!b) return failf(NET_E_IO, "Cannot find output blob %s", blob_name);
    if (b->fused_away) return failf(NET_E_IO, "blob %s was fused into its consumer; disable fusion to extract it", blob_name);
    if (b->chained)
        return failf(NET_E_IO, "blob %s exists only as the next layer's transformed input at fusion level 3; use level 2 to extract it", blob_name);
    if (b->alias) b->data = b->alias->data;
    *ptr = b->data;
    if (num) *num = b->n;
    if (c) *c = b->c;
    if (h) *h = b->h;
    if (w) *w = b->w;
    if (n->more.empty() || n->share.size() < 2 || n->share[1] < 1) return FHIP_OK;
    // the blob lives in one piece per replica: put the pieces together (device-to-device, on the net's stream, after the join)
    size_t total = b->count();
    int images = b->n;
    std::vector<std::pair<float*, size_t>> pieces(1, std::make_pair(b->data, b->count()));
    for (size_t r = 0; r < n->more.size(); ++r)
    {
        if (n->share[r + 1] < 1) continue;
        float* d = nullptr;
        int pn, pc, ph, pw;
        const int rc = fhip_net_extract(n->more[r].get(), blob_name, &d, &pn, &pc, &ph, &pw);
        if (rc) return rc;
        if (pc != b->c || ph != b->h || pw != b->w) return failf(NET_E_SHAPE, "blob %s has different shapes in the sub-batch replicas", blob_name);
        pieces.push_back(std::make_pair(d, (size_t)pn * pc * ph * pw));
        total += pieces.back().second;
        images += pn;
    }
    // The returned pointer stays valid until the next FeedInput with a different shape (like a blob pointer of a plain net): the
    // buffer of a name is only ever re-allocated when that blob's total size grows, which takes a new input shape.
    DeviceVec& g = n->gathered[blob_name];
    if (g.bytes < total * sizeof(float))
    {
        FHIP_CHECK_HIP(hipStreamSynchronize(n->impl.stream)); // earlier gathers of this name may still be in flight
        const int rc = g.resize(total * sizeof(float));
        if (rc) return rc;
    }
    size_t at = 0;
    for (auto& pc_ : pieces)
    {
        if (pc_.first && pc_.second)
            FHIP_CHECK_HIP(hipMemcpyAsync(g.d + at, pc_.first, pc_.second * sizeof(float), hipMemcpyDeviceToDevice, n->impl.stream));
        at += pc_.second;
    }
    *ptr = g.d;
    if (num) *num = images;
    return FHIP_OK;
}

int fhip_net_extract_host(fhip_net* n, const char* blob_name, float* host, size_t capacity)
{
    float* d = nullptr;
    int num, c, h, w;
    const int rc = fhip_net_extract(n, blob_name, &d, &num, &c, &h, &w);
    if (rc) return rc;
    const size_t count = (size_t)num * c * h * w;
    if (!host || capacity < count) return fail(FHIP_E_BADARG, "host buffer too small");
    if (!d) return fail(FHIP_E_BADARG, "blob has no data yet (run Forward first)");
    FHIP_CHECK_HIP(hipMemcpyAsync(host, d, count * sizeof(float), hipMemcpyDeviceToHost, n->impl.stream));
    FHIP_CHECK_HIP(hipStreamSynchronize(n->impl.stream));
    return FHIP_OK;
}

int fhip_net_layer_count(fhip_net* n)
{
    NET_GUARD(n);
    return (int)n->impl.layers.size();
}

int fhip_net_layer_info(fhip_net* n, int index, char* type, char* name, int len, int* algo)
{
    NET_GUARD(n);
    if (index < 0 || index >= (int)n->impl.layers.size()) return fail(FHIP_E_BADARG, "layer index out of range");
    Layer* l = n->impl.layers[index].get();
    if (type && len > 0) snprintf(type, len, "%s", l->type.c_str());
    if (name && len > 0) snprintf(name, len, "%s", l->name.c_str());
    if (algo) *algo = l->algo();
    return FHIP_OK;
}

int fhip_net_layer_conv_param(fhip_net* n, int index, fhip_conv_param* param, int* batch)
{
    NET_GUARD(n);
    if (index < 0 || index >= (int)n->impl.layers.size() || !param) return fail(FHIP_E_BADARG, "layer index out of range");
    Layer* l = n->impl.layers[index].get();
    const fhip_conv_param* cp = l->conv_param();
    if (!cp) return fail(FHIP_E_BADARG, "not a convolution layer");
    *param = *cp;
    if (batch) *batch = l->bottoms.empty() ? 0 : l->bottoms[0]->n;
    return FHIP_OK;
}

int fhip_net_layer_fused_pointwise(fhip_net* n, int index, fhip_conv_param* param, int* one_kernel)
{
    NET_GUARD(n);
    if (index < 0 || index >= (int)n->impl.layers.size() || !param) return fail(FHIP_E_BADARG, "layer index out of range");
    const fhip_conv_param* cp = n->impl.layers[index]->fused_pointwise(one_kernel);
    if (!cp) return fail(FHIP_E_BADARG, "no pointwise convolution was absorbed into this layer");
    *param = *cp;
    return FHIP_OK;
}

int fhip_net_layer_sibling(fhip_net* n, int index, int* state)
{
    NET_GUARD(n);
    if (index < 0 || index >= (int)n->impl.layers.size() || !state) return fail(FHIP_E_BADARG, "layer index out of range");
    *state = n->impl.layers[index]->sibling_state();
    return FHIP_OK;
}

int fhip_net_layer_residual(fhip_net* n, int index, int* state)
{
    NET_GUARD(n);
    if (index < 0 || index >= (int)n->impl.layers.size() || !state) return fail(FHIP_E_BADARG, "layer index out of range");
    *state = n->impl.layers[index]->residual_state();
    return FHIP_OK;
}

int fhip_net_layer_chain(fhip_net* n, int index, int* v_from_previous, int* writes_next_v)
{
    NET_GUARD(n);
    if (index < 0 || index >= (int)n->impl.layers.size() || !v_from_previous || !writes_next_v) return fail(FHIP_E_BADARG, "layer index out of range");
    n->impl.layers[index]->chain_state(v_from_previous, writes_next_v);
    return FHIP_OK;
}

int fhip_net_forward_timed(fhip_net* n, float* ms)
{
    NET_GUARD(n);
    if (!ms) return fail(FHIP_E_BADARG, "null argument");
    if (!n->more.empty())
        return fail(FHIP_E_UNSUPPORTED, "per-layer timing of a net with sub-batch replicas: kernels of concurrent replicas share the chip, "
                                        "so their durations are not layer times -- time a net created without fhip_net_set_sub_batches");
    Net& net = n->impl;
    int rc = prepare(net);
    if (rc) return rc;
    const size_t L = net.layers.size();
    std::vector<hipEvent_t> ev(L + 1);
    for (auto& e : ev) FHIP_CHECK_HIP(hipEventCreate(&e));
    FHIP_CHECK_HIP(hipEventRecord(ev[0], net.stream));
    for (size_t i = 0; i < L && rc == 0; ++i)
    {
        rc = net.layers[i]->Forward(net.stream);
        (void)hipEventRecord(ev[i + 1], net.stream);
    }
    (void)hipStreamSynchronize(net.stream);
    if (rc == 0)
        for (size_t i = 0; i < L; ++i) (void)hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

int fhip_net_memory(fhip_net* n, size_t* blob_bytes, size_t* weight_bytes, size_t* arena_bytes)
{
    NET_GUARD(n);
    size_t bb = 0, wb = 0;
    for (auto& kv : n->impl.blobs) bb += kv.second->capacity * sizeof(float);
    for (auto& b : n->impl.shadowed) bb += b->capacity * sizeof(float);
    for (auto& l : n->impl.layers) wb += l->weight_bytes();
    size_t ab = n->impl.arena.bytes;
    for (auto& m : n->more)
    {
        size_t b2 = 0, w2 = 0, a2 = 0;
        (void)fhip_net_memory(m.get(), &b2, &w2, &a2);
        bb += b2;
        wb += w2;
        ab += a2;
    }
    for (auto& kv : n->gathered) bb += kv.second.bytes;
    if (blob_bytes) *blob_bytes = bb;
    if (weight_bytes) *weight_bytes = wb;
    if (arena_bytes) *arena_bytes = ab;
    return FHIP_OK;
}

} // extern "C"
