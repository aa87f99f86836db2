"""Conv-layer shape lists of the benchmark networks (standard public architectures --
the reference ships no model files).  Each entry is (name, C_in, C_out, H_in, kernel, stride, pad, group).
Bias + ReLU are fused on every conv."""
from __future__ import annotations

from .booster import ConvParam, ReLU


def vgg16():
    cfg = [(3, 64, 224), (64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56), (256, 256, 56),
           (256, 512, 28), (512, 512, 28), (512, 512, 28), (512, 512, 14), (512, 512, 14), (512, 512, 14)]
    names = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3",
             "conv5_1", "conv5_2", "conv5_3"]
    return [(n, c, k, h, 3, 1, 1, 1) for n, (c, k, h) in zip(names, cfg)]


def resnet50():
    """Caffe topology: stride on the first 1x1 of a stage."""
    layers = [("conv1", 3, 64, 224, 7, 2, 3, 1)]
    cin, h = 64, 56
    for si, (mid, out, blocks, stride) in enumerate([(64, 256, 3, 1), (128, 512, 4, 2), (256, 1024, 6, 2), (512, 2048, 3, 2)]):
        for b in range(blocks):
            s = stride if b == 0 else 1
            tag = f"res{si + 2}{chr(ord('a') + b)}"
            if b == 0:
                layers.append((tag + "_proj", cin, out, h, 1, s, 0, 1))
            layers.append((tag + "_2a", cin, mid, h, 1, s, 0, 1))
            h2 = (h - 1) // s + 1
            layers.append((tag + "_2b", mid, mid, h2, 3, 1, 1, 1))
            layers.append((tag + "_2c", mid, out, h2, 1, 1, 0, 1))
            cin, h = out, h2
    return layers


def mobilenet_v1():
    layers = [("conv1", 3, 32, 224, 3, 2, 1, 1)]
    cfg = [(32, 64, 112, 1), (64, 128, 112, 2), (128, 128, 56, 1), (128, 256, 56, 2), (256, 256, 28, 1), (256, 512, 28, 2)] + \
          [(512, 512, 14, 1)] * 5 + [(512, 1024, 14, 2), (1024, 1024, 7, 1)]
    for i, (c, k, h, s) in enumerate(cfg):
        layers.append((f"conv{i + 2}_dw", c, c, h, 3, s, 1, c))
        ho = (h + 2 - 3) // s + 1
        layers.append((f"conv{i + 2}_pw", c, k, ho, 1, 1, 0, 1))
    return layers


def squeezenet_v11():
    layers = [("conv1", 3, 64, 224, 3, 2, 0, 1)]
    fires = [(64, 16, 64, 55), (128, 16, 64, 55), (128, 32, 128, 27), (256, 32, 128, 27), (256, 48, 192, 13),
             (384, 48, 192, 13), (384, 64, 256, 13), (512, 64, 256, 13)]
    for i, (cin, sq, ex, h) in enumerate(fires):
        layers.append((f"fire{i + 2}_squeeze", cin, sq, h, 1, 1, 0, 1))
        layers.append((f"fire{i + 2}_expand1x1", sq, ex, h, 1, 1, 0, 1))
        layers.append((f"fire{i + 2}_expand3x3", sq, ex, h, 3, 1, 1, 1))
    layers.append(("conv10", 512, 1000, 13, 1, 1, 0, 1))
    return layers


NETS = {"vgg16": vgg16, "resnet50": resnet50, "mobilenet_v1": mobilenet_v1, "squeezenet_v1.1": squeezenet_v11}


def layer_param(layer, batch=1) -> ConvParam:
    _, c, k, h, ks, s, p, g = layer
    return ConvParam.make(c, k, h, ks, s, p, group=g, bias=True, act=ReLU, batch=batch)
