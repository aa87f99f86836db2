"""oracle -- TEST INFRASTRUCTURE ONLY (checker, never the product).

ctypes loaders for the two CPU checkers of the FeatherCNN conv hot path:

* ``ref``  : the REAL reference (``booster::ConvBooster``, reference src/booster/avx/booster.cpp:283-355)
             compiled from /root/reference by ``oracle/Makefile`` into ``oracle/_ref/libfeather_ref.so``.
* ``port`` : our plain-C restatement ``oracle/conv_port.c`` -> ``oracle/_build/liboracle_port.so``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  ``feathercnn_amd`` must never import it (tests/test_boundary.py enforces that).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libfeather_ref.so")
PORT_SO = os.path.join(_HERE, "_build", "liboracle_port.so")

# booster::ConvAlgo (reference include/booster/booster.h:42-51)
NAIVE, IM2COL, SGECONV, DEPTHWISE, WINOGRADF63, WINOGRADF63FUSED, WINOGRADF23 = range(7)
ALGO_NAMES = {NAIVE: "NAIVE", IM2COL: "IM2COL", DEPTHWISE: "DEPTHWISE", WINOGRADF63: "WINOGRADF63"}


@dataclass(frozen=True)
class Geom:
    """Field-for-field the inputs of booster::ConvParam (booster.h:59-77); no batch, like the reference."""

    ic: int
    oc: int
    ih: int
    iw: int
    kh: int = 3
    kw: int = 3
    sh: int = 1
    sw: int = 1
    pl: int = 0
    pr: int = 0
    pt: int = 0
    pb: int = 0
    group: int = 1
    bias: int = 1
    act: int = 1

    def arr(self):
        return (ctypes.c_int * 15)(self.ic, self.oc, self.ih, self.iw, self.kh, self.kw, self.sh, self.sw,
                                   self.pl, self.pr, self.pt, self.pb, self.group, self.bias, self.act)


def conv_geom(ic, oc, h, k=3, s=1, p=0, group=1, bias=1, act=1, w=None) -> Geom:
    """Square-kernel / symmetric-pad shorthand, the only form feather::ConvLayer can express (conv_layer.h:60-63)."""
    return Geom(ic, oc, h, h if w is None else w, k, k, s, s, p, p, p, p, group, bias, act)


def build(which=("port", "ref")) -> None:
    """Build the checkers (``make -C oracle``).  ``ref`` is a no-op where /root/reference is absent."""
    for tgt in which:
        subprocess.run(["make", "-s", "-C", _HERE, tgt], check=True)


_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int)
_D = ctypes.POINTER(ctypes.c_double)


def _fp(a):
    return a.ctypes.data_as(_F) if a is not None else None


class _Lib:
    prefix = ""

    def __init__(self, path):
        self.path = path
        self.lib = ctypes.CDLL(path)
        p = self.prefix
        getattr(self.lib, p + "conv_output_dims").argtypes = [_I, _I, _I, _I]
        getattr(self.lib, p + "conv_select_algo").argtypes = [_I]
        fwd = getattr(self.lib, p + "conv_forward")
        fwd.argtypes = [_I, ctypes.c_int, ctypes.c_int, _F, _F, _F, _F]
        fl = getattr(self.lib, p + "conv_flops")
        fl.argtypes = [_I]
        fl.restype = ctypes.c_double

    def output_dims(self, g: Geom):
        c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        getattr(self.lib, self.prefix + "conv_output_dims")(g.arr(), ctypes.byref(c), ctypes.byref(h), ctypes.byref(w))
        return c.value, h.value, w.value

    def select_algo(self, g: Geom) -> int:
        return getattr(self.lib, self.prefix + "conv_select_algo")(g.arr())

    def flops(self, g: Geom) -> float:
        return getattr(self.lib, self.prefix + "conv_flops")(g.arr())

    def forward(self, g: Geom, x: np.ndarray, w: np.ndarray, b: np.ndarray | None, algo: int = -1) -> np.ndarray:
        """x [N][C][H][W] fp32 -> [N][K][Ho][Wo]; algo<0 = the reference's SelectAlgo."""
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        oc, oh, ow = self.output_dims(g)
        if b is None:
            b = np.zeros(oc, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        n = x.shape[0]
        out = np.empty((n, oc, oh, ow), np.float32)
        rc = getattr(self.lib, self.prefix + "conv_forward")(g.arr(), algo, n, _fp(x), _fp(w), _fp(b), _fp(out))
        if rc != 0:
            raise RuntimeError(f"{self.prefix}conv_forward rc={rc} for {g}")
        return out


class RefLib(_Lib):
    prefix = "ref_"

    def __init__(self, path=REF_SO):
        super().__init__(path)
        self.lib.ref_conv_time.argtypes = [_I, ctypes.c_int, _F, _F, _F, ctypes.c_int, ctypes.c_int, _D, _D]
        self.lib.ref_conv_buffer_size.argtypes = [_I, ctypes.c_int, _I, _I]

    def buffer_size(self, g: Geom, algo: int = -1):
        a, b = ctypes.c_int(), ctypes.c_int()
        rc = self.lib.ref_conv_buffer_size(g.arr(), algo, ctypes.byref(a), ctypes.byref(b))
        if rc != 0:
            raise RuntimeError("ref_conv_buffer_size failed")
        return a.value, b.value

    def time_forward(self, g: Geom, x, w, b, warmup=1, reps=3, algo=-1):
        """Seconds per single-image Forward (best, mean); Init untimed; 1 thread."""
        best, mean = ctypes.c_double(), ctypes.c_double()
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        rc = self.lib.ref_conv_time(g.arr(), algo, _fp(x), _fp(w), _fp(b), warmup, reps, ctypes.byref(best), ctypes.byref(mean))
        if rc != 0:
            raise RuntimeError("ref_conv_time failed")
        return best.value, mean.value


class PortLib(_Lib):
    prefix = "port_"

    def __init__(self, path=PORT_SO):
        super().__init__(path)
        self.lib.port_conv_direct_f64.argtypes = [_I, ctypes.c_int, _F, _F, _F, _F]

    def direct_f64(self, g: Geom, x, w, b) -> np.ndarray:
        """fp64-accumulated direct convolution (the external yardstick of SURVEY.md 8c/8d)."""
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        oc, oh, ow = self.output_dims(g)
        if b is None:
            b = np.zeros(oc, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        out = np.empty((x.shape[0], oc, oh, ow), np.float32)
        rc = self.lib.port_conv_direct_f64(g.arr(), x.shape[0], _fp(x), _fp(w), _fp(b), _fp(out))
        if rc != 0:
            raise RuntimeError("port_conv_direct_f64 failed")
        return out


_cache: dict = {}


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def ref() -> RefLib:
    if "ref" not in _cache:
        _cache["ref"] = RefLib()
    return _cache["ref"]


def port() -> PortLib:
    if "port" not in _cache:
        if not os.path.exists(PORT_SO):
            build(("port",))
        _cache["port"] = PortLib()
    return _cache["port"]


def best() -> _Lib:
    """The strongest checker available: the real reference if its .so is present, else the restatement."""
    return ref() if have_ref() else port()


def synth(g: Geom, batch: int, seed: int = 1234):
    """Seeded synthetic tensors, SURVEY.md 8(d): input U(-1,1), weights U(-1,1)/sqrt(C/group*kh*kw), bias U(-.1,.1)."""
    rng = np.random.default_rng(seed)
    cpg = g.ic // max(g.group, 1)
    oc = g.ic if g.group == g.ic and g.group != 1 else g.oc
    x = rng.uniform(-1, 1, (batch, g.ic, g.ih, g.iw)).astype(np.float32)
    w = (rng.uniform(-1, 1, (oc, cpg, g.kh, g.kw)) / np.sqrt(cpg * g.kh * g.kw)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, (oc,)).astype(np.float32)
    return x, w, b


def nerr(a: np.ndarray, ref_: np.ndarray) -> float:
    """Normalised max error max|a-ref|/max|ref| -- the parity metric of SURVEY.md 8(d)."""
    d = float(np.max(np.abs(a.astype(np.float64) - ref_.astype(np.float64))))
    m = float(np.max(np.abs(ref_.astype(np.float64))))
    return d / m if m > 0 else d
