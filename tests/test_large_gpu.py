"""Tensors beyond 2^31 elements (the 288 GB of an MI355X invite batches the CPU reference never saw): every route is run
once on a batch whose input, output or Winograd scratch holds more than 2^31 floats and checked at random output positions
drawn across the WHOLE tensor against an fp64 direct convolution evaluated on the device from gathered input taps.  A 32-bit
index anywhere in a kernel shows up as garbage in the upper part of the tensor."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4

# name, C, K, H, k, s, p, group, batch, what exceeds 2^31 floats
BIG = [("winograd_V_and_M", 64, 64, 224, 3, 1, 1, 1, 512, "V/M = 64*64*739328 = 3.0e9 floats each; in/out 1.6e9"),
       ("igemm_1x1_input", 256, 64, 56, 1, 1, 0, 1, 2800, "input 2.25e9 floats"),
       ("depthwise_in_out", 32, 32, 112, 3, 1, 1, 32, 5400, "input and output 2.17e9 floats each")]


@pytest.mark.parametrize("cfg", BIG, ids=[c[0] for c in BIG])
def test_sampled_fp64_direct_conv_across_the_whole_tensor(cfg, cuda):
    import torch

    from feathercnn_amd import ConvLayer, ConvParam
    name, ic, oc, h, k, s, p, group, batch, _ = cfg
    prm = ConvParam.make(ic, oc, h, k, s, p, group=group, bias=True, act=1, batch=batch)
    gen = torch.Generator(device=cuda).manual_seed(5)
    cpg = ic // group
    w = (torch.rand((prm.output_channels, cpg, k, k), device=cuda, generator=gen) * 2 - 1) / (cpg * k * k) ** 0.5
    b = (torch.rand((prm.output_channels,), device=cuda, generator=gen) * 2 - 1) * 0.1
    lyr = ConvLayer(prm, w, b)
    x = torch.empty((batch, ic, h, h), device=cuda)
    for i in range(0, batch, 64):  # generate in pieces: torch.rand materialises temporaries
        x[i:i + 64].uniform_(-1, 1, generator=gen)
    assert max(x.numel(), lyr.buffer_bytes // 4) > 2 ** 31
    y = lyr.Forward(x)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (batch, prm.output_channels, prm.output_h, prm.output_w)
    # 4000 positions, half of them forced into the last images (the highest addresses)
    S = 4000
    g2 = torch.Generator().manual_seed(9)
    n_ = torch.randint(0, batch, (S,), generator=g2)
    n_[S // 2:] = torch.randint(max(batch - 8, 0), batch, (S - S // 2,), generator=g2)
    k_ = torch.randint(0, prm.output_channels, (S,), generator=g2)
    oy = torch.randint(0, prm.output_h, (S,), generator=g2)
    ox = torch.randint(0, prm.output_w, (S,), generator=g2)
    n_, k_, oy, ox = (t.to(cuda) for t in (n_, k_, oy, ox))
    cc = torch.arange(cpg, device=cuda)
    acc = b.double()[k_].clone()
    for u in range(k):
        for v in range(k):
            yy, xx = oy * s - p + u, ox * s - p + v
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < h)
            c_idx = (k_[:, None] if group > 1 else torch.zeros_like(k_)[:, None]) + cc[None, :]
            taps = x[n_[:, None], c_idx, yy.clamp(0, h - 1)[:, None], xx.clamp(0, h - 1)[:, None]].double()   # [S][cpg]
            acc += torch.where(ok, (taps * w.double()[k_, :, u, v]).sum(1), torch.zeros_like(acc))
    want = acc.clamp_min(0)
    got = y[n_, k_, oy, ox].double()
    scale = float(y[:64].abs().max())
    assert float((got - want).abs().max()) <= TOL * scale, name
    assert float(got[S // 2:].abs().max()) > 0, "the last images came back empty"
