"""Synthetic ncnn-format models (``.param`` text + ``.bin`` weights) of the benchmark networks.

The reference ships no model files and there is no network, so whole-net runs use the standard public architectures
with seeded random weights written in the format ``feather::Net`` loads (reference
src/net.cpp:67-170, src/ncnn/paramdict.cpp:92-174, src/ncnn/modelbin.cpp:47-197).  The same files feed the reference
``feather::Net`` (the CPU checker of the tests) and this package's ``Net``.

Weight distribution: He-uniform ``U(-1,1)*sqrt(6/fan_in)`` so activations keep O(1) magnitude through 50 layers;
bias ``U(-0.1,0.1)``; BatchNorm slope/var ``U(0.5,1.5)``, mean/bias ``U(-0.1,0.1)``.
"""
from __future__ import annotations

import struct

import numpy as np


class GraphBuilder:
    def __init__(self, seed: int = 1234, dry: bool = False):
        """dry=True: build the .param only and count the .bin bytes (``nbytes``) without drawing any weights."""
        self.rng = np.random.default_rng(seed)
        self.lines: list[str] = []
        self.bin = bytearray()
        self.blob_count = 0
        self.dry = dry
        self.nbytes = 0

    # -- low level ----------------------------------------------------------------------------------------------
    def layer(self, type_, name, bottoms, tops, params=None):
        kv = " ".join(f"{k}={v}" for k, v in (params or {}).items())
        self.lines.append(f"{type_} {name} {len(bottoms)} {len(tops)} {' '.join(list(bottoms) + list(tops))} {kv}".rstrip())
        self.blob_count += len(tops)
        return tops[0] if tops else None

    def _tagged(self, a):  # mb.load(n, 0): 4-byte zero tag + raw fp32
        self.nbytes += 4
        if not self.dry:
            self.bin += struct.pack("<I", 0)
        self._raw(a)

    def _raw(self, a):  # mb.load(n, 1)
        if self.dry:
            self.nbytes += 4 * a
        else:
            self.bin += np.ascontiguousarray(a, dtype="<f4").tobytes()
            self.nbytes += 4 * a.size

    def _uniform(self, n, lo, hi, scale=None):
        if self.dry:
            return n
        a = self.rng.uniform(lo, hi, size=n).astype(np.float32)
        return a if scale is None else a * np.float32(scale)

    # -- layers -------------------------------------------------------------------------------------------------
    def input(self, name, c, h, w):
        return self.layer("Input", name, [], [name], {0: w, 1: h, 2: c})

    def conv(self, name, bottom, cin, cout, k, s=1, p=0, group=1, bias=True, top=None):
        fan_in = cin // group * k * k
        wsize = cout * (cin // group) * k * k
        type_ = "ConvolutionDepthWise" if group > 1 else "Convolution"
        top = self.layer(type_, name, [bottom], [top or name], {0: cout, 1: k, 3: s, 4: p, 5: int(bias), 6: wsize, 7: group})
        self._tagged(self._uniform(wsize, -1, 1, np.sqrt(6.0 / fan_in)))
        if bias:
            self._raw(self._uniform(cout, -0.1, 0.1))
        return top

    def relu(self, name, bottom):
        return self.layer("ReLU", name, [bottom], [name])

    def pool(self, name, bottom, k=2, s=2, p=0, avg=False, global_=False):
        return self.layer("Pooling", name, [bottom], [name], {0: int(avg), 1: k, 2: s, 3: p, 4: int(global_)})

    def fc(self, name, bottom, cin, cout, bias=True):
        top = self.layer("InnerProduct", name, [bottom], [name], {0: cout, 1: int(bias), 2: cin * cout})
        self._tagged(self._uniform(cin * cout, -1, 1, np.sqrt(6.0 / cin)))
        if bias:
            self._raw(self._uniform(cout, -0.1, 0.1))
        return top

    def bn(self, name, bottom, c, eps=1e-5):
        top = self.layer("BatchNorm", name, [bottom], [name], {0: c, 1: f"{eps:.6e}"})
        self._raw(self._uniform(c, 0.5, 1.5))    # slope
        self._raw(self._uniform(c, -0.1, 0.1))   # mean
        self._raw(self._uniform(c, 0.5, 1.5))    # var
        self._raw(self._uniform(c, -0.1, 0.1))   # bias
        return top

    def scale(self, name, bottom, c, bias=True):
        top = self.layer("Scale", name, [bottom], [name], {0: c, 1: int(bias)})
        self._raw(self._uniform(c, 0.5, 1.5))
        if bias:
            self._raw(self._uniform(c, -0.1, 0.1))
        return top

    def split(self, name, bottom, n=2):
        tops = [f"{name}_{i}" for i in range(n)]
        self.layer("Split", name, [bottom], tops)
        return tops

    def eltwise(self, name, a, b):
        return self.layer("Eltwise", name, [a, b], [name], {0: 1})

    def concat(self, name, bottoms):
        return self.layer("Concat", name, list(bottoms), [name], {0: 0})

    def dropout(self, name, bottom, scale=None):
        return self.layer("Dropout", name, [bottom], [name], {} if scale is None else {0: f"{scale:.6f}"})

    def softmax(self, name, bottom):
        return self.layer("Softmax", name, [bottom], [name])

    def conv_bn_relu(self, name, bottom, cin, cout, k, s=1, p=0, group=1, relu=True):
        # bias on every dense conv: the reference Winograd output transform reads bias[k] unconditionally and
        # ConvLayer passes NULL without bias_term; none on depthwise, whose bias blob the reference
        # Net sizes wrongly
        x = self.conv(name, bottom, cin, cout, k, s, p, group, bias=(group == 1))
        x = self.bn(name + "_bn", x, cout)
        x = self.scale(name + "_scale", x, cout)
        return self.relu(name + "_relu", x) if relu else x

    def finish(self):
        text = "7767517\n%d %d\n" % (len(self.lines), self.blob_count) + "\n".join(self.lines) + "\n"
        return text.encode(), (self.nbytes if self.dry else bytes(self.bin))  # dry: the .bin size instead of the .bin


def vgg16(seed=1234, classes=1000, size=224, dry=False):
    g = GraphBuilder(seed, dry)
    x = g.input("data", 3, size, size)
    cin = 3
    for stage, (n, c) in enumerate([(2, 64), (2, 128), (3, 256), (3, 512), (3, 512)], 1):
        for i in range(1, n + 1):
            x = g.relu(f"relu{stage}_{i}", g.conv(f"conv{stage}_{i}", x, cin, c, 3, 1, 1))
            cin = c
        x = g.pool(f"pool{stage}", x, 2, 2)
    feat = 512 * (size // 32) ** 2
    x = g.dropout("drop6", g.relu("relu6", g.fc("fc6", x, feat, 4096)))
    x = g.dropout("drop7", g.relu("relu7", g.fc("fc7", x, 4096, 4096)))
    x = g.softmax("prob", g.fc("fc8", x, 4096, classes))
    return g.finish() + ("data", "prob")


def resnet50(seed=1234, classes=1000, size=224, dry=False):
    """Caffe ResNet-50: conv-BN-Scale-ReLU, stride on the first 1x1 of each stage, explicit Split for the shortcut."""
    g = GraphBuilder(seed, dry)
    x = g.input("data", 3, size, size)
    x = g.conv_bn_relu("conv1", x, 3, 64, 7, 2, 3)
    x = g.pool("pool1", x, 3, 2)
    cin = 64
    for si, (mid, out, blocks, stride) in enumerate([(64, 256, 3, 1), (128, 512, 4, 2), (256, 1024, 6, 2), (512, 2048, 3, 2)]):
        for b in range(blocks):
            s = stride if b == 0 else 1
            tag = f"res{si + 2}{chr(ord('a') + b)}"
            main, short = g.split(tag + "_split", x)
            if b == 0:
                short = g.conv_bn_relu(tag + "_branch1", short, cin, out, 1, s, 0, relu=False)
            y = g.conv_bn_relu(tag + "_branch2a", main, cin, mid, 1, s, 0)
            y = g.conv_bn_relu(tag + "_branch2b", y, mid, mid, 3, 1, 1)
            y = g.conv_bn_relu(tag + "_branch2c", y, mid, out, 1, 1, 0, relu=False)
            x = g.relu(tag + "_relu", g.eltwise(tag, short, y))
            cin = out
    x = g.pool("pool5", x, 7, 1, avg=True, global_=True)
    x = g.softmax("prob", g.fc("fc1000", x, 2048, classes))
    return g.finish() + ("data", "prob")


def mobilenet_v1(seed=1234, classes=1000, size=224, dry=False):
    g = GraphBuilder(seed, dry)
    x = g.input("data", 3, size, size)
    x = g.conv_bn_relu("conv1", x, 3, 32, 3, 2, 1)
    cfg = [(32, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2), (256, 256, 1), (256, 512, 2)] + [(512, 512, 1)] * 5 + \
          [(512, 1024, 2), (1024, 1024, 1)]
    for i, (c, k, s) in enumerate(cfg, 2):
        x = g.conv_bn_relu(f"conv{i}_dw", x, c, c, 3, s, 1, group=c)
        x = g.conv_bn_relu(f"conv{i}_pw", x, c, k, 1, 1, 0)
    x = g.pool("pool6", x, 7, 1, avg=True, global_=True)
    x = g.softmax("prob", g.fc("fc7", x, 1024, classes))
    return g.finish() + ("data", "prob")


def squeezenet_v11(seed=1234, classes=1000, size=224, dry=False):
    g = GraphBuilder(seed, dry)
    x = g.input("data", 3, size, size)
    x = g.relu("relu_conv1", g.conv("conv1", x, 3, 64, 3, 2, 0))
    x = g.pool("pool1", x, 3, 2)
    cin = 64

    def fire(name, x, cin, sq, ex):
        s = g.relu(name + "_relu_squeeze", g.conv(name + "_squeeze1x1", x, cin, sq, 1))
        a, b = g.split(name + "_split", s)
        a = g.relu(name + "_relu_e1", g.conv(name + "_expand1x1", a, sq, ex, 1))
        b = g.relu(name + "_relu_e3", g.conv(name + "_expand3x3", b, sq, ex, 3, 1, 1))
        return g.concat(name + "_concat", [a, b])

    for i, (sq, ex) in enumerate([(16, 64), (16, 64), (32, 128), (32, 128), (48, 192), (48, 192), (64, 256), (64, 256)], 2):
        x = fire(f"fire{i}", x, cin, sq, ex)
        cin = 2 * ex
        if i in (3, 5):
            x = g.pool(f"pool{i}", x, 3, 2)
    x = g.dropout("drop9", x)
    x = g.relu("relu_conv10", g.conv("conv10", x, 512, classes, 1))
    x = g.pool("pool10", x, 13, 1, avg=True, global_=True)
    x = g.softmax("prob", x)
    return g.finish() + ("data", "prob")


def tiny_allsorts(seed=7, size=20, dry=False):
    """A small net touching every registered layer type (layer_factory.cpp:55-67) for the Net parity tests."""
    g = GraphBuilder(seed, dry)
    x = g.input("data", 3, size, size)
    x = g.relu("relu1", g.conv("conv1", x, 3, 16, 3, 1, 1))             # IM2COL (C=3)
    x = g.conv_bn_relu("conv2", x, 16, 16, 3, 1, 1)                      # Winograd + BN + Scale + ReLU
    a, b = g.split("split1", x)
    a = g.conv_bn_relu("dw", a, 16, 16, 3, 1, 1, group=16)               # depthwise
    b = g.scale("scale_b", g.conv("pw", b, 16, 16, 1), 16, bias=False)   # 1x1 + bare Scale
    x = g.relu("relu_sum", g.eltwise("sum", a, b))
    x = g.pool("pool1", x, 3, 2)                                         # max, ceil mode
    c, d = g.split("split2", x)
    c = g.relu("relu_c", g.conv("conv_c", c, 16, 8, 1))
    d = g.pool("pool_d", d, 3, 1, p=1, avg=True)                         # average with the reference's pad rule
    x = g.concat("cat", [c, d])
    x = g.dropout("drop", x, scale=0.5)
    x = g.pool("gap", x, 1, 1, avg=True, global_=True)
    x = g.relu("relu_fc", g.fc("fc1", x, 24, 32))
    x = g.softmax("prob", g.fc("fc2", x, 32, 10))
    return g.finish() + ("data", "prob")


MODELS = {"vgg16": vgg16, "resnet50": resnet50, "mobilenet_v1": mobilenet_v1, "squeezenet_v1.1": squeezenet_v11,
          "tiny_allsorts": tiny_allsorts}
