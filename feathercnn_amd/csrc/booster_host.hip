// booster_host.hip -- C++ host mirror of the reference's ConvBooster (reference src/booster/avx/booster.cpp:273-355):
// the class binds three plain function pointers per algorithm; each of them forwards to the C-ABI with the
// algorithm baked in, the batch taken from ConvParam::batch and the stream from booster::SetStream().
// Error behaviour follows the reference: 0 ok, -1 for an unsupported algo / partial group with the function
// pointers left NULL (avx/booster.cpp:304-308, 348-354).
#include <limits.h>

#include "booster/booster.h"
#include "feather_hip/feather_hip.h"

#define BOOSTER_EXPORT __attribute__((visibility("default")))

namespace booster
{

static thread_local void* g_stream = nullptr;

BOOSTER_EXPORT void SetStream(void* hip_stream) { g_stream = hip_stream; }
BOOSTER_EXPORT void* GetStream() { return g_stream; }

static fhip_conv_param to_c(const ConvParam* p)
{
    fhip_conv_param c;
    c.output_channels = p->output_channels;
    c.input_channels = p->input_channels;
    c.input_h = p->input_h;
    c.input_w = p->input_w;
    c.kernel_h = p->kernel_h;
    c.kernel_w = p->kernel_w;
    c.output_h = p->output_h;
    c.output_w = p->output_w;
    c.stride_h = p->stride_h;
    c.stride_w = p->stride_w;
    c.pad_left = p->pad_left;
    c.pad_bottom = p->pad_bottom;
    c.pad_right = p->pad_right;
    c.pad_top = p->pad_top;
    c.group = p->group;
    c.bias_term = p->bias_term ? 1 : 0;
    c.activation = p->activation == ReLU ? FHIP_ACT_RELU : FHIP_ACT_NONE;
    return c;
}

static int batch_of(const ConvParam* p) { return p->batch > 0 ? p->batch : 1; }

// ---- ConvParam members (reference booster.h:113-148), on top of the C-ABI so there is one definition of every rule -------------
BOOSTER_EXPORT void ConvParam::AssignOutputDim()
{
    fhip_conv_param c = to_c(this);
    fhip_conv_assign_output_dim(&c);
    group = c.group;
    stride_h = c.stride_h;
    stride_w = c.stride_w;
    output_h = c.output_h;
    output_w = c.output_w;
    output_channels = c.output_channels;
}

BOOSTER_EXPORT void ConvParam::AssignPaddedDim()
{
    input_h += pad_top + pad_bottom;
    input_w += pad_left + pad_right;
    pad_left = pad_bottom = pad_right = pad_top = 0;
}

BOOSTER_EXPORT void ConvParam::LogParams(const char* layer_name)
{
    printf("ConvParam of layer %s: in %d x %d x %d, out %d x %d x %d, kernel %d x %d, stride %d x %d, pads l%d b%d r%d t%d, group %d, bias %d, "
           "activation %d, batch %d\n",
           layer_name, input_channels, input_h, input_w, output_channels, output_h, output_w, kernel_h, kernel_w, stride_h, stride_w, pad_left,
           pad_bottom, pad_right, pad_top, group, bias_term ? 1 : 0, (int)activation, batch_of(this));
}

BOOSTER_EXPORT double ConvParam::GetFLOPS()
{
    const fhip_conv_param c = to_c(this);
    return fhip_conv_flops(&c);
}

template <int ALGO>
static int T_GetBufferSize(ConvParam* param, int* buffer_size, int* processed_kernel_size)
{
    const fhip_conv_param c = to_c(param);
    size_t buf = 0, pk = 0;
    int rc = fhip_conv_get_buffer_size(&c, ALGO, batch_of(param), &buf, &pk);
    if (rc != 0) return rc;
    // the reference reports FLOAT COUNTS in int (booster.h:151)
    if (buf / sizeof(float) > (size_t)INT_MAX || pk / sizeof(float) > (size_t)INT_MAX) return -1;
    *buffer_size = (int)(buf / sizeof(float));
    *processed_kernel_size = (int)(pk / sizeof(float));
    return 0;
}

template <int ALGO>
static int T_Init(ConvParam* param, float* processed_kernel, float* kernel)
{
    const fhip_conv_param c = to_c(param);
    return fhip_conv_init(&c, ALGO, processed_kernel, kernel, g_stream);
}

template <int ALGO>
static int T_Forward(ConvParam* param, float* output, float* input, float* kernel, float* buffer, float* bias_arr,
                     int /*num_threads*/)
{
    const fhip_conv_param c = to_c(param);
    return fhip_conv_forward(&c, ALGO, batch_of(param), output, input, kernel, buffer, bias_arr, g_stream);
}

BOOSTER_EXPORT ConvBooster::ConvBooster() : GetBufferSize(NULL), Init(NULL), Forward(NULL), algo(NAIVE) {}

BOOSTER_EXPORT int ConvBooster::SelectAlgo(ConvParam* param)
{
    const fhip_conv_param c = to_c(param);
    int a = -1;
    if (fhip_conv_select_algo(&c, &a) != 0)
    {
        fprintf(stderr, "Partial group conv is not yet supported.\n");
        return -1;
    }
    this->algo = (ConvAlgo)a;
    return this->SetFuncs();
}

BOOSTER_EXPORT int ConvBooster::ForceSelectAlgo(ConvAlgo a)
{
    this->algo = a;
    return this->SetFuncs();
}

BOOSTER_EXPORT int ConvBooster::SetFuncs()
{
#define BIND(A)                                 \
    this->GetBufferSize = T_GetBufferSize<A>;   \
    this->Init = T_Init<A>;                     \
    this->Forward = T_Forward<A>;               \
    return 0
    switch (this->algo)
    {
        case NAIVE: BIND(FHIP_NAIVE);
        case IM2COL: BIND(FHIP_IM2COL);
        case WINOGRADF63: BIND(FHIP_WINOGRADF63);
        case DEPTHWISE: BIND(FHIP_DEPTHWISE);
        default:
            fprintf(stderr, "This algo is not supported on gfx950.\n");
            this->GetBufferSize = NULL;
            this->Init = NULL;
            this->Forward = NULL;
            return -1;
    }
#undef BIND
}

BOOSTER_EXPORT int ConvBooster::GetBufferSizeBytes(ConvParam* param, size_t* buffer_bytes, size_t* processed_kernel_bytes)
{
    const fhip_conv_param c = to_c(param);
    return fhip_conv_get_buffer_size(&c, (int)this->algo, batch_of(param), buffer_bytes, processed_kernel_bytes);
}

} // namespace booster
