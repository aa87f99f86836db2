"""Python host-side mirror of the reference operator interface for the conv hot path.

Same names, argument meaning and error behaviour as ``booster::ConvParam`` / ``booster::ConvBooster``
(reference src/booster/include/booster/booster.h:59-170, src/booster/avx/booster.cpp:283-355) and the caller
contract of ``feather::ConvLayer`` (reference src/layers/conv_layer.h:92-172), so the parity tests read like
code written against the reference.  Tensors are torch CUDA tensors (device memory + stream plumbing only);
every call goes through the C-ABI of ``libfeather_hip.so``.  No fallback path exists.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

from . import _lib

# booster::ConvAlgo (booster.h:42-51)
NAIVE, IM2COL, SGECONV, DEPTHWISE, WINOGRADF63, WINOGRADF63FUSED, WINOGRADF23 = range(7)
ALGO_NAMES = {NAIVE: "NAIVE", IM2COL: "IM2COL", SGECONV: "SGECONV", DEPTHWISE: "DEPTHWISE", WINOGRADF63: "WINOGRADF63",
              WINOGRADF63FUSED: "WINOGRADF63FUSED", WINOGRADF23: "WINOGRADF23"}
# booster::ActivationType (booster.h:53-57); `None` is a Python keyword, hence None_
None_, ReLU = 0, 1


class FeatherHipError(RuntimeError):
    pass


def _check(rc: int, what: str):
    if rc != 0:
        msg = _lib.load_library().fhip_last_error().decode(errors="replace")
        raise FeatherHipError(f"{what} failed with code {rc}: {msg}")


@dataclass
class ConvParam:
    """booster::ConvParam (booster.h:59-77) + the GPU-side ``batch`` extension (0/1 = the reference's N=1)."""
    output_channels: int = 0
    input_channels: int = 0
    input_h: int = 0
    input_w: int = 0
    kernel_h: int = 0
    kernel_w: int = 0
    output_h: int = 0
    output_w: int = 0
    stride_h: int = 0
    stride_w: int = 0
    pad_left: int = 0
    pad_bottom: int = 0
    pad_right: int = 0
    pad_top: int = 0
    group: int = 0
    bias_term: bool = False
    activation: int = None_
    batch: int = 1

    def _c(self) -> _lib.fhip_conv_param:
        return _lib.fhip_conv_param(self.output_channels, self.input_channels, self.input_h, self.input_w, self.kernel_h,
                                    self.kernel_w, self.output_h, self.output_w, self.stride_h, self.stride_w,
                                    self.pad_left, self.pad_bottom, self.pad_right, self.pad_top, self.group,
                                    1 if self.bias_term else 0, int(self.activation))

    def AssignOutputDim(self):
        """ConvParam::AssignOutputDim (booster.h:113-125), computed by the library so host and device agree."""
        c = self._c()
        _check(_lib.load_library().fhip_conv_assign_output_dim(ctypes.byref(c)), "fhip_conv_assign_output_dim")
        self.group, self.stride_h, self.stride_w = c.group, c.stride_h, c.stride_w
        self.output_h, self.output_w, self.output_channels = c.output_h, c.output_w, c.output_channels

    def GetFLOPS(self) -> float:
        """ConvParam::GetFLOPS (booster.h:145-148), per image."""
        c = self._c()
        return _lib.load_library().fhip_conv_flops(ctypes.byref(c))

    @staticmethod
    def make(ic, oc, h, k=3, s=1, p=0, group=1, bias=True, act=ReLU, w=None, batch=1) -> "ConvParam":
        q = ConvParam(output_channels=oc, input_channels=ic, input_h=h, input_w=h if w is None else w, kernel_h=k,
                      kernel_w=k, stride_h=s, stride_w=s, pad_left=p, pad_bottom=p, pad_right=p, pad_top=p, group=group,
                      bias_term=bool(bias), activation=act, batch=batch)
        q.AssignOutputDim()
        return q


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise FeatherHipError("feathercnn_amd needs device tensors (there is no CPU path)")
    if not t.is_contiguous():
        raise FeatherHipError("tensors must be dense NCHW")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class ConvBooster:
    """booster::ConvBooster (booster.h:156-170).  Does not allocate: the caller owns every tensor."""

    def __init__(self):
        self.algo = None

    def SelectAlgo(self, param: ConvParam, tuned: bool = False) -> int:
        """avx/booster.cpp:283-310.  Returns 0, or -1 for partial groups (callers may ignore it, as ConvLayer does).
        tuned=True asks for the MI355X cost model instead of the reference rule (fhip_conv_select_algo_tuned)."""
        a = ctypes.c_int(-1)
        c = param._c()
        lib = _lib.load_library()
        rc = (lib.fhip_conv_select_algo_tuned if tuned else lib.fhip_conv_select_algo)(ctypes.byref(c), ctypes.byref(a))
        if rc != 0:
            self.algo = None
            return -1
        self.algo = a.value
        return 0

    def ForceSelectAlgo(self, algo: int) -> int:
        """avx/booster.cpp:313-317 + SetFuncs :319-355: -1 and no bound functions for unsupported algos."""
        if algo not in (NAIVE, IM2COL, DEPTHWISE, WINOGRADF63):
            self.algo = None
            return -1
        self.algo = algo
        return 0

    def _need_algo(self):
        if self.algo is None:
            raise FeatherHipError("no algorithm bound (SelectAlgo / ForceSelectAlgo returned -1)")

    def GetBufferSize(self, param: ConvParam):
        """GET_BUFFER_SIZE_FUNC: returns (buffer_bytes, processed_kernel_bytes) for param.batch images."""
        self._need_algo()
        b, k = ctypes.c_size_t(), ctypes.c_size_t()
        c = param._c()
        _check(_lib.load_library().fhip_conv_get_buffer_size(ctypes.byref(c), self.algo, max(param.batch, 1), ctypes.byref(b),
                                                            ctypes.byref(k)), "fhip_conv_get_buffer_size")
        return b.value, k.value

    def Init(self, param: ConvParam, processed_kernel, kernel) -> int:
        """INIT_FUNC: one-time weight pre-processing on the device."""
        self._need_algo()
        c = param._c()
        _check(_lib.load_library().fhip_conv_init(ctypes.byref(c), self.algo, _ptr(processed_kernel), _ptr(kernel), _stream()),
               "fhip_conv_init")
        return 0

    def Forward(self, param: ConvParam, output, input, processed_kernel, buffer, bias_arr, num_threads: int = 1) -> int:
        """FORWARD_FUNC; num_threads is accepted and ignored."""
        self._need_algo()
        c = param._c()
        _check(_lib.load_library().fhip_conv_forward(ctypes.byref(c), self.algo, max(param.batch, 1), _ptr(output), _ptr(input),
                                                    _ptr(processed_kernel), _ptr(buffer), _ptr(bias_arr), _stream()),
               "fhip_conv_forward")
        return 0


class ConvLayer:
    """The caller side of the boundary, after feather::ConvLayer (reference src/layers/conv_layer.h:92-172):
    Reshape (AssignOutputDim + SelectAlgo + GetBufferSize), Init once (packed weights replace raw weights),
    Forward per batch.  The scratch arena is handed in by the owner, like CommonMemPool (mempool.cpp:88-109)."""

    def __init__(self, param: ConvParam, weight, bias=None, algo: int | None = None, tuned: bool = False):
        import torch
        self.param = param
        self.param.AssignOutputDim()
        self.booster = ConvBooster()
        rc = self.booster.SelectAlgo(param, tuned) if algo is None else self.booster.ForceSelectAlgo(algo)
        if rc != 0:
            raise FeatherHipError("unsupported convolution (partial group or algo)")
        self.buffer_bytes, self.packed_bytes = self.booster.GetBufferSize(param)
        self.bias = bias
        self.packed = torch.empty(max(self.packed_bytes // 4, 1), dtype=torch.float32, device=weight.device)
        self.booster.Init(param, self.packed, weight.contiguous())

    def out_shape(self):
        p = self.param
        return (max(p.batch, 1), p.output_channels, p.output_h, p.output_w)

    def Forward(self, x, out=None, scratch=None):
        import torch
        if out is None:
            out = torch.empty(self.out_shape(), dtype=torch.float32, device=x.device)
        if scratch is None and self.buffer_bytes:
            scratch = torch.empty(self.buffer_bytes // 4, dtype=torch.float32, device=x.device)
        self.booster.Forward(self.param, out, x, self.packed, scratch, self.bias)
        return out


class SiblingConvs:
    """Two 1x1 convolutions of the same input as one GEMM (feather_net.h: fhip_conv_forward_siblings) -- ResNet's projection shortcut and the
    first layer of the main branch.  Built from the two layers' geometries, filters [K][C][1][1] and biases (or None)."""

    def __init__(self, pa: ConvParam, wa, ba, pb: ConvParam, wb, bb):
        import torch
        lib = _lib.load_library()
        pa.AssignOutputDim()
        pb.AssignOutputDim()
        self.pa, self.pb = pa, pb
        ca, cb, both = pa._c(), pb._c(), _lib.fhip_conv_param()
        _check(lib.fhip_conv_siblings_geometry(ctypes.byref(ca), ctypes.byref(cb), ctypes.byref(both)), "fhip_conv_siblings_geometry")
        self.both = ConvParam(output_channels=both.output_channels, input_channels=both.input_channels, input_h=both.input_h, input_w=both.input_w,
                              kernel_h=1, kernel_w=1, stride_h=both.stride_h, stride_w=both.stride_w, pad_left=0, pad_right=0, pad_top=0, pad_bottom=0,
                              group=1, bias_term=bool(both.bias_term), activation=0, batch=pa.batch)
        dev = wa.device
        w = torch.cat([wa.reshape(pa.output_channels, -1), wb.reshape(pb.output_channels, -1)]).contiguous()
        zeros = lambda k: torch.zeros(k, dtype=torch.float32, device=dev)
        self.bias = torch.cat([ba if ba is not None else zeros(pa.output_channels), bb if bb is not None else zeros(pb.output_channels)]) \
            if self.both.bias_term else None
        self.layer = ConvLayer(self.both, w, self.bias, algo=IM2COL)  # packs the stacked filters (fhip_conv_init of the stacked geometry)

    def applicable(self, batch: int) -> bool:
        ca, cb = self.pa._c(), self.pb._c()
        return bool(_lib.load_library().fhip_conv_can_fuse_siblings(ctypes.byref(ca), IM2COL, ctypes.byref(cb), IM2COL, int(batch)))

    def Forward(self, x):
        import torch
        n = x.shape[0]
        ya = torch.empty((n, self.pa.output_channels, self.pa.output_h, self.pa.output_w), dtype=torch.float32, device=x.device)
        yb = torch.empty((n, self.pb.output_channels, self.pb.output_h, self.pb.output_w), dtype=torch.float32, device=x.device)
        ca, cb = self.pa._c(), self.pb._c()
        _check(_lib.load_library().fhip_conv_forward_siblings(ctypes.byref(ca), ctypes.byref(cb), n, _ptr(ya), _ptr(yb), _ptr(x), _ptr(self.layer.packed),
                                                              _ptr(self.bias) if self.bias is not None else None, _stream()), "fhip_conv_forward_siblings")
        return ya, yb


def stage_timing(enable: bool):
    _check(_lib.load_library().fhip_stage_timing_enable(1 if enable else 0), "fhip_stage_timing_enable")


def stage_timing_collect():
    """-> {stage_name: (total_ms, launches)} since the last collect."""
    n = len(_lib.STAGE_NAMES)
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_longlong * n)()
    _check(_lib.load_library().fhip_stage_timing_collect(ms, cnt), "fhip_stage_timing_collect")
    return {name: (ms[i], cnt[i]) for i, name in enumerate(_lib.STAGE_NAMES)}


def winograd_plan(param: ConvParam):
    pl = _lib.fhip_winograd_plan()
    c = param._c()
    _check(_lib.load_library().fhip_winograd_f63_plan(ctypes.byref(c), max(param.batch, 1), ctypes.byref(pl)),
           "fhip_winograd_f63_plan")
    return pl


def winograd_rows(buf, plan, rows: int):
    """A Winograd scratch tensor (V: rows = input channels, M: rows = output channels) as [xi][rows][Pp], whatever its storage: the
    library keeps V and M in blocks of plan.column_block columns, [Pp / BP][xi][rows][BP] (include/feather_hip/feather_hip.h; BP == Pp is
    the whole-row form).  xi runs over plan.frequency_points: 64 for F(6x6,3x3), 36 on the planes that run F(4x4,3x3).  Works on torch
    tensors and numpy arrays (flat, at least frequency_points * rows * Pp floats)."""
    pp, bp, nxi = plan.columns_padded, plan.column_block, plan.frequency_points
    flat = buf.reshape(-1)[:nxi * rows * pp]
    if bp >= pp:
        return flat.reshape(nxi, rows, pp)
    blocked = flat.reshape(pp // bp, nxi, rows, bp)
    return (blocked.permute(1, 2, 0, 3) if hasattr(blocked, "permute") else blocked.transpose(1, 2, 0, 3)).reshape(nxi, rows, pp)


def can_chain_winograd(a: "ConvLayer", b: "ConvLayer", pool: bool = False) -> bool:
    ca, cb = a.param._c(), b.param._c()
    return bool(_lib.load_library().fhip_conv_can_chain_winograd(ctypes.byref(ca), a.booster.algo, ctypes.byref(cb), b.booster.algo, int(pool)))


def can_fuse_first_winograd(first: "ConvParam", nxt: "ConvLayer", batch: int) -> bool:
    """feather_net.h: can `first` (3x3 / s1 / p1, 2 .. 4 input channels) be computed inside the input transform of the Winograd layer `nxt`?"""
    cf, cn = first._c(), nxt.param._c()
    return bool(_lib.load_library().fhip_conv_can_fuse_first_winograd(ctypes.byref(cf), ctypes.byref(cn), nxt.booster.algo, int(batch)))


def forward_chained(layers, x, pools=None, first=None):
    """A run of Winograd ConvLayers through fhip_conv_forward_chained: the activations between them never exist (V ping-pongs
    between two scratch buffers).  pools[i] = a 2x2 / stride-2 max pooling follows layer i.  Every adjacent pair must satisfy
    can_chain_winograd.  first = (ConvParam, filters [K][C][3][3], bias or None) of a convolution in FRONT of layers[0] that is computed
    inside layers[0]'s input transform (fhip_winograd_f63_input_from_first; `x` is then that convolution's input).
    -> output of the last layer (pooled if pools[-1])."""
    import torch
    lib = _lib.load_library()
    pools = list(pools) if pools is not None else [False] * len(layers)
    batch = x.shape[0]
    plans = []
    for l in layers:
        l.param.batch = batch
        plans.append(winograd_plan(l.param))
    dev = x.device
    vbuf = [torch.empty(max(pl.v_bytes for pl in plans[k::2]) // 4, dtype=torch.float32, device=dev) if plans[k::2] else None for k in (0, 1)]
    m = torch.empty(max(pl.m_bytes for pl in plans) // 4, dtype=torch.float32, device=dev)
    last = layers[-1].param
    oh, ow = (last.output_h // 2, last.output_w // 2) if pools[-1] else (last.output_h, last.output_w)
    out = torch.empty((batch, last.output_channels, oh, ow), dtype=torch.float32, device=dev)
    if first is not None:
        fprm, fw, fb = first
        fprm.batch = batch
        cf, c0 = fprm._c(), layers[0].param._c()
        _check(lib.fhip_winograd_f63_input_from_first(ctypes.byref(cf), ctypes.byref(c0), batch, _ptr(vbuf[0]), _ptr(x), _ptr(fw),
                                                      _ptr(fb) if fb is not None else None, _stream()), "fhip_winograd_f63_input_from_first")
    for i, l in enumerate(layers):
        c = l.param._c()
        nxt = layers[i + 1].param._c() if i + 1 < len(layers) else None
        _check(lib.fhip_conv_forward_chained(ctypes.byref(c), batch, _ptr(out) if nxt is None else None, _ptr(x) if i == 0 and first is None else None,
                                             _ptr(l.packed), _ptr(vbuf[i & 1]), _ptr(m), _ptr(l.bias) if l.bias is not None else None,
                                             ctypes.byref(nxt) if nxt is not None else None, _ptr(vbuf[(i + 1) & 1]) if nxt is not None else None,
                                             int(pools[i]), _stream()), "fhip_conv_forward_chained")
    return out


def calibrate_mfma_f32():
    """-> (TFLOP/s the fp32 matrix pipe of this device sustains under full-chip load, shader clock in MHz it ran at) -- feather_hip.h."""
    tf, mhz = ctypes.c_double(), ctypes.c_double()
    _check(_lib.load_library().fhip_calibrate_mfma_f32(ctypes.byref(tf), ctypes.byref(mhz), _stream()), "fhip_calibrate_mfma_f32")
    return tf.value, mhz.value
