"""oracle.netcheck -- TEST INFRASTRUCTURE ONLY: the two CPU checkers of the whole-net path.

* ``RefNet``  : the REAL reference runtime ``feather::Net`` (reference src/net.cpp) behind oracle/net_shim.cpp, compiled
                by oracle/Makefile into ``oracle/_ref/libfeather_net_ref.so``.  N = 1: a batch is a loop over images.
* ``PortNet`` : our restatement -- an ncnn ``.param``/``.bin`` reader and the layer arithmetic in numpy, convolutions
                through the plain-C restatement (oracle/conv_port.c).  Each function cites the reference lines it follows
                (paths relative to /root/reference/src).  Pinned against ``RefNet`` by tests/test_oracle.py.

Known reference defects the restatement does NOT reproduce (they are crashes, not results): Winograd without bias_term
dereferences a NULL bias; depthwise bias is loaded with length 1 (2.3 #5); InnerProduct without
bias_term dereferences weights[1] (inner_product_layer.h:91).
"""
from __future__ import annotations

import ctypes
import math
import os
import struct
import tempfile

import numpy as np

from . import Geom, _HERE, port

NET_REF_SO = os.path.join(_HERE, "_ref", "libfeather_net_ref.so")


def have_ref_net() -> bool:
    return os.path.exists(NET_REF_SO)


class RefNet:
    def __init__(self, param: bytes | None, weights: bytes | None, param_path: str | None = None, bin_path: str | None = None):
        """Either the model images (written to a temporary directory) or paths of existing .param/.bin files."""
        lib = ctypes.CDLL(NET_REF_SO)
        lib.ref_net_open.restype = ctypes.c_void_p
        lib.ref_net_open.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        lib.ref_net_run.restype = ctypes.c_long
        lib.ref_net_run.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_char_p, ctypes.c_void_p, ctypes.c_long, ctypes.POINTER(ctypes.c_int)]
        lib.ref_net_time.restype = ctypes.c_double
        lib.ref_net_time.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.ref_net_close.argtypes = [ctypes.c_void_p]
        lib.ref_net_extract.restype = ctypes.c_long
        lib.ref_net_extract.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_long, ctypes.POINTER(ctypes.c_int)]
        lib.ref_net_time_each.restype = None
        lib.ref_net_time_each.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        self.lib = lib
        if param_path is not None:
            self.h = lib.ref_net_open(param_path.encode(), bin_path.encode())
        else:
          with tempfile.TemporaryDirectory() as d:
              pp, bp = os.path.join(d, "m.param"), os.path.join(d, "m.bin")
              open(pp, "wb").write(param)
              open(bp, "wb").write(weights)
              self.h = lib.ref_net_open(pp.encode(), bp.encode())
        if not self.h:
            raise RuntimeError("reference feather::Net failed to load the model")

    def run(self, input_name: str, x: np.ndarray, output_name: str, capacity: int = 1 << 24) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        outs = []
        for img in x:
            buf = np.empty(capacity, np.float32)
            dims = (ctypes.c_int * 4)()
            n = self.lib.ref_net_run(self.h, input_name.encode(), img.ctypes.data_as(ctypes.c_void_p), *img.shape,
                                     output_name.encode(), buf.ctypes.data_as(ctypes.c_void_p), capacity, dims)
            if n < 0:
                raise RuntimeError(f"ref_net_run rc={n}")
            outs.append(buf[:n].reshape(dims[1], dims[2], dims[3]).copy())
        return np.stack(outs)

    def run_blobs(self, input_name: str, image: np.ndarray, names, capacity: int = 1 << 24):
        """ONE reference forward of one image [c][h][w]; the named blobs of that run, each as [1][c][h][w]."""
        first = self.run(input_name, image[None], names[0], capacity)
        outs = [first]
        for nm in names[1:]:
            buf = np.empty(capacity, np.float32)
            dims = (ctypes.c_int * 4)()
            n = self.lib.ref_net_extract(self.h, nm.encode(), buf.ctypes.data_as(ctypes.c_void_p), capacity, dims)
            if n < 0:
                raise RuntimeError(f"ref_net_extract rc={n}")
            outs.append(buf[:n].reshape(1, dims[1], dims[2], dims[3]).copy())
        return tuple(outs)

    def time_forward(self, reps: int = 3) -> float:
        return self.lib.ref_net_time(self.h, reps)

    def time_each(self, warmup: int = 1, reps: int = 3):
        """`warmup` untimed + `reps` individually timed forwards of the image already fed -> list of seconds."""
        secs = (ctypes.c_double * reps)()
        self.lib.ref_net_time_each(self.h, warmup, reps, secs)
        return list(secs)

    def close(self):
        if self.h:
            self.lib.ref_net_close(self.h)
            self.h = None


# ---- restatement ------------------------------------------------------------------------------------------------------
def parse_param(text: bytes):
    """net.cpp:67-170 + ncnn/paramdict.cpp:92-174: token stream; 'id=value', float iff the value has '.' or 'e'."""
    tok = text.decode().split()
    assert int(tok[0]) == 7767517, "param is too old"  # utils.cpp:27-44
    nlayers = int(tok[1])
    t, layers = 3, []
    for _ in range(nlayers):
        type_, name, nb, nt = tok[t], tok[t + 1], int(tok[t + 2]), int(tok[t + 3])
        t += 4
        bottoms, tops = tok[t:t + nb], tok[t + nb:t + nb + nt]
        t += nb + nt
        pd = {}
        while t < len(tok) and "=" in tok[t] and tok[t].split("=")[0].lstrip("-").isdigit():
            k, v = tok[t].split("=", 1)
            pd[int(k)] = float(v) if ("." in v or "e" in v.lower()) else int(v)
            t += 1
        layers.append((type_, name, bottoms, tops, pd))
    return layers


class _Bin:
    def __init__(self, data: bytes):
        self.d, self.o = data, 0

    def load(self, n: int, tagged: bool) -> np.ndarray:  # ncnn/modelbin.cpp:52-189, raw fp32 payloads only
        if tagged:
            (tag,) = struct.unpack_from("<I", self.d, self.o)
            assert tag == 0, "restatement reads raw fp32 weights only"
            self.o += 4
        a = np.frombuffer(self.d, "<f4", n, self.o).copy()
        self.o += 4 * n
        return a


def _pool(x, pd):
    """pooling_layer.h:37-131: ceil output dims; window origin j*stride - pad_top - pad_bottom (both pads); clipped
    windows; average divides by the in-range tap count."""
    n, c, h, w = x.shape
    typ, kw = pd.get(0, 0), pd.get(1, 0)
    kh, sw = pd.get(11, kw), pd.get(2, 1)
    sh, pl = pd.get(12, sw), pd.get(3, 0)
    pr, pt = pd.get(14, pl), pd.get(13, pl)
    pb = pd.get(15, pt)
    if pd.get(4, 0):
        kh, kw, oh, ow = h, w, 1, 1
    else:
        oh = int(math.ceil(np.float32(h + pt + pb - kh) / np.float32(sh))) + 1
        ow = int(math.ceil(np.float32(w + pl + pr - kw) / np.float32(sw))) + 1
    y = np.empty((n, c, oh, ow), np.float32)
    for j in range(oh):
        y0 = j * sh - pt - pb
        ya, yb = max(y0, 0), min(y0 + kh, h)
        for k in range(ow):
            x0 = k * sw - pl - pr
            xa, xb = max(x0, 0), min(x0 + kw, w)
            win = x[:, :, ya:yb, xa:xb].reshape(n, c, -1)
            if typ != 0:
                cnt = win.shape[2]
                y[:, :, j, k] = win.sum(axis=2, dtype=np.float32) / np.float32(cnt) if cnt else np.nan
            else:
                y[:, :, j, k] = win.max(axis=2) if win.shape[2] else -np.finfo(np.float32).max
    return y


class PortNet:
    def __init__(self, param: bytes, weights: bytes):
        self.layers = parse_param(param)
        mb = _Bin(weights)
        self.w = {}
        for type_, name, _, _, pd in self.layers:
            if type_ in ("Convolution", "ConvolutionDepthWise"):  # conv_layer.h:41-131
                group, kw = pd.get(7, 1), pd.get(1, 0)
                kh = pd.get(11, kw)
                oc = pd.get(0, 0) // group
                ic = pd.get(6, 0) // oc // kh // kw
                k_out = ic if group == ic and group != 1 else oc
                wgt = mb.load(ic * oc * kh * kw, True)
                b = mb.load(k_out, False) if pd.get(5, 0) else None
                self.w[name] = (wgt, b, ic, oc, group)
            elif type_ == "InnerProduct":  # inner_product_layer.h:112-162
                out = pd.get(0, 0)
                wgt = mb.load(pd.get(2, 0), True).reshape(out, -1)
                self.w[name] = (wgt, mb.load(out, False) if pd.get(1, 0) else None)
            elif type_ == "BatchNorm":  # batchnorm_layer.h:43-75
                c = pd.get(0, 0)
                slope, mean, var, bias = (mb.load(c, False) for _ in range(4))
                sq = np.sqrt(var + np.float32(pd.get(1, 0.0)), dtype=np.float32)
                self.w[name] = (slope / sq, bias - slope * mean / sq)  # beta, alpha
            elif type_ == "Scale":  # scale_layer.h:45-69
                c = pd.get(0, 0)
                s = mb.load(c, False)
                self.w[name] = (s, mb.load(c, False) if pd.get(1, 0) else None)

    def run(self, input_name: str, x: np.ndarray, output_name: str, keep: bool = False):
        blobs = {input_name: np.ascontiguousarray(x, np.float32)}
        for type_, name, bottoms, tops, pd in self.layers:
            if type_ == "Input":
                continue
            a = blobs[bottoms[0]]
            if type_ in ("Convolution", "ConvolutionDepthWise"):
                wgt, b, ic, oc, group = self.w[name]
                kw, sw, pw = pd.get(1, 0), pd.get(3, 1), pd.get(4, 0)
                kh, sh, ph = pd.get(11, kw), pd.get(13, sw), pd.get(14, pw)
                g = Geom(ic, oc, a.shape[2], a.shape[3], kh, kw, sh, sw, pw, pw, ph, ph, group, 1 if b is not None else 0, 0)
                y = port().forward(g, a, wgt, b)
            elif type_ == "ReLU":  # relu_layer.h:29-41
                y = np.where(a > 0, a, np.float32(0))
            elif type_ == "Pooling":
                y = _pool(a, pd)
            elif type_ == "InnerProduct":
                wgt, b = self.w[name]
                y = a.reshape(a.shape[0], -1).astype(np.float64) @ wgt.T.astype(np.float64)
                if b is not None:
                    y = y + b
                y = y.astype(np.float32).reshape(a.shape[0], -1, 1, 1)
            elif type_ == "BatchNorm":  # generic_kernels.cpp:237-279: beta*x + alpha
                beta, alpha = self.w[name]
                y = a * beta[None, :, None, None] + alpha[None, :, None, None]
            elif type_ == "Scale":  # generic_kernels.cpp:203-233
                s, b = self.w[name]
                y = a * s[None, :, None, None]
                if b is not None:
                    y = y + b[None, :, None, None]
            elif type_ == "Eltwise":  # eltwise_layer.h:69-80 (SUM of bottoms 0 and 1)
                y = a + blobs[bottoms[1]]
            elif type_ == "Concat":  # concat_layer.h:37-48
                y = np.concatenate([blobs[b] for b in bottoms], axis=1)
            elif type_ == "Split":  # split_layer.h:43-52
                for t in tops:
                    blobs[t] = a
                continue
            elif type_ == "Dropout":  # dropout_layer.h:36-58
                sc = pd.get(0, 1.0)
                y = a if sc == 1.0 else a * np.float32(sc)
            elif type_ == "Softmax":  # softmax_layer.h:33-53: over the whole image
                f = a.reshape(a.shape[0], -1)
                e = np.exp(f - f.max(axis=1, keepdims=True), dtype=np.float32)
                y = (e / e.sum(axis=1, keepdims=True, dtype=np.float32)).reshape(a.shape)
            else:
                raise RuntimeError(f"layer {type_} not exists or registered")  # net.cpp:108-112
            blobs[tops[0]] = np.ascontiguousarray(y, np.float32)
        return blobs if keep else blobs[output_name]
