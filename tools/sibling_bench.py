"""tools/sibling_bench.py -- ResNet-50 b64: projection shortcut + first main-branch layer as two launches vs one (fhip_conv_forward_siblings)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from feathercnn_amd import ConvLayer, ConvParam  # noqa: E402
from feathercnn_amd.booster import SiblingConvs  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, c, ka, kb, h, s in (("res2a", 64, 256, 64, 56, 1), ("res3a", 256, 512, 128, 56, 2), ("res4a", 512, 1024, 256, 28, 2), ("res5a", 1024, 2048, 512, 14, 2)):
    def one(k, relu):
        w = torch.from_numpy((rng.standard_normal((k, c, 1, 1)) / np.sqrt(c)).astype(np.float32)).to(dev)
        b = torch.from_numpy(rng.uniform(-0.1, 0.1, k).astype(np.float32)).to(dev)
        p = ConvParam(output_channels=k, input_channels=c, input_h=h, input_w=h, kernel_h=1, kernel_w=1, stride_h=s, stride_w=s, group=1, bias_term=True,
                      activation=1 if relu else 0, batch=batch)
        return p, w, b
    pa, wa, ba = one(ka, False)
    pb, wb, bb = one(kb, True)
    la, lb = ConvLayer(pa, wa, ba, tuned=True), ConvLayer(pb, wb, bb, tuned=True)
    sib = SiblingConvs(pa, wa, ba, pb, wb, bb)
    x = torch.from_numpy(rng.uniform(-1, 1, (batch, c, h, h)).astype(np.float32)).to(dev)
    oa, ob = la.Forward(x), lb.Forward(x)
    sa = torch.empty(max(la.buffer_bytes, lb.buffer_bytes, 4) // 4, dtype=torch.float32, device=dev)
    ts = []
    for _ in range(3):
        t_a = timeit(lambda: la.Forward(x, out=oa, scratch=sa))
        t_b = timeit(lambda: lb.Forward(x, out=ob, scratch=sa))
        t_ab = timeit(lambda: sib.Forward(x)) if sib.applicable(batch) else float("nan")
        ts.append((t_a, t_b, t_ab))
    t_a, t_b, t_ab = (min(t[i] for t in ts) for i in range(3))
    print(f"{name}: C {c} -> {ka} + {kb} @{h} s{s}: {t_a:.1f} + {t_b:.1f} = {t_a + t_b:.1f} us separately, {t_ab:.1f} us as one GEMM (applicable: {sib.applicable(batch)})")
