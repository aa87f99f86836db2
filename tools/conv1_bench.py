"""First layers of the benchmark nets through the C-ABI: time and write bandwidth.  python tools/conv1_bench.py"""
import numpy as np
import torch

from feathercnn_amd import ConvLayer, ConvParam

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for name, K, k, s, p, batch in (("vgg16 conv1_1 b32", 64, 3, 1, 1, 32), ("mobilenet conv1 b256", 32, 3, 2, 1, 256), ("resnet50 conv1 b64", 64, 7, 2, 3, 64)):
    w = torch.from_numpy((rng.standard_normal((K, 3, k, k)) / np.sqrt(3 * k * k)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.uniform(-0.1, 0.1, K).astype(np.float32)).to(dev)
    prm = ConvParam(output_channels=K, input_channels=3, input_h=224, input_w=224, kernel_h=k, kernel_w=k, stride_h=s, stride_w=s, pad_left=p,
                    pad_right=p, pad_top=p, pad_bottom=p, group=1, bias_term=True, activation=1, batch=batch)
    l = ConvLayer(prm, w, b, tuned=True)
    x = torch.from_numpy(rng.uniform(-1, 1, (batch, 3, 224, 224)).astype(np.float32)).to(dev)
    out = torch.empty(l.out_shape(), dtype=torch.float32, device=dev)
    scratch = torch.empty(max(l.buffer_bytes // 4, 1), dtype=torch.float32, device=dev)
    for _ in range(5):
        l.Forward(x, out, scratch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        l.Forward(x, out, scratch)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50
    by = (x.numel() + out.numel()) * 4
    print(f"{name}: {t * 1e3:.1f} us  {by / t / 1e9:.2f} TB/s (input + output)  {2.0 * K * 3 * k * k * out.shape[2] * out.shape[3] * batch / t / 1e9:.1f} TF")
