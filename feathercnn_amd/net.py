"""Python host-side mirror of ``feather::Net`` (reference src/net.h:30-70, src/net.cpp) over the C-ABI in
``include/feather_hip/feather_net.h``: ``LoadParam`` / ``LoadWeights`` / ``FeedInput`` / ``Forward`` / ``Extract`` with the
reference's names, plus the batch dimension.  Blobs live in HBM; the wrapper only moves pointers.  No fallback path."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .booster import ALGO_NAMES, FeatherHipError, _check, _stream


class Net:
    def __init__(self, fusion: int = 1, graph: bool = False, stream=None, tuned: bool = False, concurrency: bool = False,
                 sub_batches: int = 1):
        self._lib = _lib.load_library()
        h = ctypes.c_void_p()
        _check(self._lib.fhip_net_create(ctypes.byref(h)), "fhip_net_create")
        self._h = h
        _check(self._lib.fhip_net_set_fusion(h, int(fusion)), "fhip_net_set_fusion")
        _check(self._lib.fhip_net_set_graph(h, int(bool(graph))), "fhip_net_set_graph")
        _check(self._lib.fhip_net_set_tuned_selection(h, int(bool(tuned))), "fhip_net_set_tuned_selection")
        _check(self._lib.fhip_net_set_concurrency(h, int(bool(concurrency))), "fhip_net_set_concurrency")
        if stream is not None:
            _check(self._lib.fhip_net_set_stream(h, ctypes.c_void_p(stream)), "fhip_net_set_stream")
        if sub_batches != 1:  # replicas of the net on streams of their own, each taking a share of every batch (feather_net.h)
            _check(self._lib.fhip_net_set_sub_batches(h, int(sub_batches)), "fhip_net_set_sub_batches")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fhip_net_destroy(self._h)
            self._h = None

    __del__ = close

    def set_graph(self, on: bool):
        _check(self._lib.fhip_net_set_graph(self._h, int(bool(on))), "fhip_net_set_graph")

    def use_current_stream(self):
        """Enqueue on torch's current stream (so torch events and tensors order against the net's work)."""
        _check(self._lib.fhip_net_set_stream(self._h, _stream()), "fhip_net_set_stream")

    # -- reference API ------------------------------------------------------------------------------------------
    def LoadParam(self, param) -> int:
        """Net::LoadParam (net.cpp:54-170): a path, or the .param text itself as bytes."""
        if isinstance(param, bytes):
            _check(self._lib.fhip_net_load_param_mem(self._h, param, len(param)), "fhip_net_load_param_mem")
        else:
            _check(self._lib.fhip_net_load_param(self._h, str(param).encode()), "fhip_net_load_param")
        return 0

    def LoadWeights(self, weights) -> int:
        """Net::LoadWeights (net.cpp:172-233): a path, or the .bin image as bytes."""
        if isinstance(weights, (bytes, bytearray, memoryview)):
            buf = (ctypes.c_char * len(weights)).from_buffer_copy(weights)
            _check(self._lib.fhip_net_load_weights_mem(self._h, buf, len(weights)), "fhip_net_load_weights_mem")
        else:
            _check(self._lib.fhip_net_load_weights(self._h, str(weights).encode()), "fhip_net_load_weights")
        return 0

    def FeedInput(self, input_name: str, data) -> int:
        """Net::FeedInput (net.cpp:235-246) with a batch: `data` is [N][C][H][W] (or [C][H][W]) fp32, a numpy array
        (host) or a torch CUDA tensor (device, copied device-to-device)."""
        import torch
        if isinstance(data, torch.Tensor) and data.is_cuda:
            t = data.contiguous().float()
            torch.cuda.current_stream().synchronize()  # the net's stream may differ from the producer's
            shape, ptr, dev = tuple(t.shape), ctypes.c_void_p(t.data_ptr()), 1
            keep = t
        else:
            a = np.ascontiguousarray(np.asarray(data, dtype=np.float32))
            shape, ptr, dev = a.shape, a.ctypes.data_as(ctypes.c_void_p), 0
            keep = a
        if len(shape) == 3:
            shape = (1,) + tuple(shape)
        if len(shape) != 4:
            raise FeatherHipError("FeedInput wants [N][C][H][W] or [C][H][W]")
        n, c, h, w = (int(v) for v in shape)
        _check(self._lib.fhip_net_feed_input(self._h, input_name.encode(), n, c, h, w, ptr, dev), "fhip_net_feed_input")
        self.synchronize()  # the host array / device tensor may be freed or overwritten by the caller
        del keep
        return 0

    def Forward(self) -> int:
        _check(self._lib.fhip_net_forward(self._h), "fhip_net_forward")
        return 0

    def Extract(self, blob_name: str) -> np.ndarray:
        """Net::Extract (net.cpp:263-296): the blob as a host [N][C][H][W] array (synchronises the stream)."""
        ptr, shape = self.ExtractDevice(blob_name)
        out = np.empty(shape, dtype=np.float32)
        _check(self._lib.fhip_net_extract_host(self._h, blob_name.encode(), out.ctypes.data_as(ctypes.c_void_p), out.size),
               "fhip_net_extract_host")
        return out

    def ExtractDevice(self, blob_name: str):
        """(device pointer, (n, c, h, w)) of a blob -- the float** form of Net::Extract."""
        p = ctypes.c_void_p()
        dims = [ctypes.c_int() for _ in range(4)]
        _check(self._lib.fhip_net_extract(self._h, blob_name.encode(), ctypes.byref(p), *[ctypes.byref(d) for d in dims]),
               "fhip_net_extract")
        return p.value, tuple(d.value for d in dims)

    # -- introspection ------------------------------------------------------------------------------------------
    def synchronize(self):
        import torch
        torch.cuda.synchronize()  # device-wide: covers the net's own stream too

    def layers(self):
        out = []
        n = self._lib.fhip_net_layer_count(self._h)
        for i in range(n):
            t, nm, algo = ctypes.create_string_buffer(64), ctypes.create_string_buffer(256), ctypes.c_int()
            _check(self._lib.fhip_net_layer_info(self._h, i, t, nm, 64, ctypes.byref(algo)), "fhip_net_layer_info")
            out.append((t.value.decode(), nm.value.decode(), ALGO_NAMES.get(algo.value)))
        return out

    def conv_params(self):
        """{layer index: (fhip_conv_param, batch)} for the convolution layers as they run (after Reshape)."""
        out = {}
        for i in range(self._lib.fhip_net_layer_count(self._h)):
            p, b = _lib.fhip_conv_param(), ctypes.c_int()
            if self._lib.fhip_net_layer_conv_param(self._h, i, ctypes.byref(p), ctypes.byref(b)) == 0:
                out[i] = (p, b.value)
        return out

    def fused_pointwise(self):
        """{layer index: (fhip_conv_param of the 1x1 convolution a depthwise layer absorbed, runs_as_one_kernel)} (fusion level 2)."""
        out = {}
        for i in range(self._lib.fhip_net_layer_count(self._h)):
            p, one = _lib.fhip_conv_param(), ctypes.c_int()
            if self._lib.fhip_net_layer_fused_pointwise(self._h, i, ctypes.byref(p), ctypes.byref(one)) == 0:
                out[i] = (p, bool(one.value))
        return out

    def siblings(self):
        """{layer index: 1 | 2} -- 1: this 1x1 convolution's launch also computes the next layer (fhip_conv_forward_siblings), 2: that next layer."""
        out = {}
        for i in range(self._lib.fhip_net_layer_count(self._h)):
            st = ctypes.c_int()
            if self._lib.fhip_net_layer_sibling(self._h, i, ctypes.byref(st)) == 0 and st.value:
                out[i] = st.value
        return out

    def residuals(self):
        """{layer index: 1 | 2} -- 1: the absorbed Eltwise SUM operand is added in this convolution's GEMM epilogue (one more output-sized read by
        the same launch, fhip_conv_forward_residual), 2: absorbed but added by a launch of its own (fhip_net_layer_residual)."""
        out = {}
        for i in range(self._lib.fhip_net_layer_count(self._h)):
            st = ctypes.c_int()
            if self._lib.fhip_net_layer_residual(self._h, i, ctypes.byref(st)) == 0 and st.value:
                out[i] = st.value
        return out

    def chains(self, raw=False):
        """{layer index: (v_from_previous, writes_next_v)} for the convolutions that are part of a chained Winograd run (fusion level 3).
        raw=True keeps the library's values: 2 marks the pair "first layer computed inside the next layer's input transform" -- (0, 2) on the
        first layer, which is not a Winograd layer and launches nothing, (2, x) on its consumer.  The boolean view (raw=False) is about chained
        Winograd layers only: the first layer is left out of it and the consumer's V is not "from the previous layer's chained transform"."""
        out = {}
        for i in range(self._lib.fhip_net_layer_count(self._h)):
            a, b = ctypes.c_int(), ctypes.c_int()
            if self._lib.fhip_net_layer_chain(self._h, i, ctypes.byref(a), ctypes.byref(b)) == 0 and (a.value or b.value):
                if raw:
                    out[i] = (a.value, b.value)
                elif a.value == 1 or b.value == 1:
                    out[i] = (a.value == 1, b.value == 1)
        return out

    def forward_timed(self):
        """One eager forward with HIP events around every layer: [(type, name, algo, ms)]."""
        n = self._lib.fhip_net_layer_count(self._h)
        ms = (ctypes.c_float * max(n, 1))()
        _check(self._lib.fhip_net_forward_timed(self._h, ms), "fhip_net_forward_timed")
        info = self.layers()  # fusion may have shrunk the list during the first forward
        return [(t, nm, a, ms[i]) for i, (t, nm, a) in enumerate(info)]

    def memory(self):
        v = [ctypes.c_size_t() for _ in range(3)]
        _check(self._lib.fhip_net_memory(self._h, *[ctypes.byref(x) for x in v]), "fhip_net_memory")
        return {"blob_bytes": v[0].value, "weight_bytes": v[1].value, "arena_bytes": v[2].value}


# ---- the layer kernels on torch CUDA tensors (tests and ad-hoc use) ---------------------------------------------
def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def relu(x):
    import torch
    y = torch.empty_like(x)
    _check(_lib.load_library().fhip_relu(_p(y), _p(x), x.numel(), _stream()), "fhip_relu")
    return y


def add(a, b, relu: bool = False):
    import torch
    y = torch.empty_like(a)
    _check(_lib.load_library().fhip_add(_p(y), _p(a), _p(b), a.numel(), int(relu), _stream()), "fhip_add")
    return y


def affine(x, mul, add_=None, relu: bool = False):
    import torch
    y = torch.empty_like(x)
    n, c, h, w = x.shape
    _check(_lib.load_library().fhip_affine(_p(y), _p(x), _p(mul), _p(add_), n, c, h * w, int(relu), _stream()),
           "fhip_affine")
    return y


def pool_param(c, h, w, kernel, stride=1, pad=(0, 0, 0, 0), pooling_type=0, global_pooling=False):
    """pad = (left, right, top, bottom) as PoolingLayer::LoadParam names them."""
    kh, kw = (kernel, kernel) if isinstance(kernel, int) else kernel
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    return _lib.fhip_pool_param(c, h, w, kh, kw, sh, sw, pad[0], pad[1], pad[2], pad[3], pooling_type, int(global_pooling))


def pooling(x, q):
    import torch
    oh, ow = ctypes.c_int(), ctypes.c_int()
    lib = _lib.load_library()
    _check(lib.fhip_pooling_output_dim(ctypes.byref(q), ctypes.byref(oh), ctypes.byref(ow)), "fhip_pooling_output_dim")
    y = torch.empty((x.shape[0], x.shape[1], oh.value, ow.value), device=x.device, dtype=torch.float32)
    _check(lib.fhip_pooling(ctypes.byref(q), x.shape[0], _p(y), _p(x), _stream()), "fhip_pooling")
    return y


def softmax(x):
    import torch
    y = torch.empty_like(x)
    _check(_lib.load_library().fhip_softmax(_p(y), _p(x), x.shape[0], x[0].numel(), _stream()), "fhip_softmax")
    return y
