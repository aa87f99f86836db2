"""Row pieces for the last tiles of a short Winograd tile-GEMM launch (round 5; wino_gemm_glds.h: wino_gemm_row_split, kernel template SPLIT; the
function replaced is reference TensorGEMM, src/booster/avx/winograd_kernels_F63.cpp:518-692).  A launch whose tile count leaves a remainder of at
most a quarter (half) of the CUs cuts those tiles into 4 (2) row pieces.  Geometries for each class on a 256-CU device -- quarter pieces, half
pieces, a remainder that is not a multiple of 8 (surplus blocks exit), F(6x6,3x3) with 64 and F(4x4,3x3) with 36 frequency points -- checked
against the oracle and, bit for bit, against the same images run in small batches (few tiles: whole tiles only; the k-order of every output
element is the same whatever the tiling)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import conv_geom, nerr, synth

pytestmark = pytest.mark.gpu

# (C, K, H, batch, small batch, tiles, pieces): tiles = frequency points x row tiles x column tiles of 64
CASES = [(128, 512, 7, 64, 8, 576, 4),    # ResNet-50's res5 class: 36 x 4 x 4, remainder 64 -> quarters
         (128, 256, 7, 80, 8, 360, 2),    # 36 x 2 x 5, remainder 104 -> halves
         (128, 128, 7, 136, 8, 324, 2),   # 36 x 1 x 9, remainder 68 (not a multiple of 8) -> halves, 4 surplus blocks
         (128, 128, 28, 12, 3, 320, 4),   # 64 x 1 x 5, remainder 64 -> quarters
         (128, 128, 28, 13, 1, 384, 2)]   # 64 x 1 x 6, remainder 128 -> halves


def _layer(cuda, c, k, h, batch, wb=None):
    from feathercnn_amd import ConvLayer, ConvParam
    from feathercnn_amd.booster import WINOGRADF63
    g = conv_geom(c, k, h, 3, 1, 1)
    x, w, b = synth(g, batch, seed=c + k + h)
    if wb is not None:
        w, b = wb
    p = ConvParam(output_channels=k, input_channels=c, input_h=h, input_w=h, kernel_h=3, kernel_w=3, stride_h=1, stride_w=1, pad_left=1, pad_right=1,
                  pad_top=1, pad_bottom=1, group=1, bias_term=True, activation=1, batch=batch)
    return g, x, w, b, p, ConvLayer(p, torch.from_numpy(w).to(cuda), torch.from_numpy(b).to(cuda), algo=WINOGRADF63)


@pytest.mark.parametrize("c,k,h,batch,small,tiles,pieces", CASES)
def test_row_pieces_match_whole_tiles_and_the_oracle(cuda, c, k, h, batch, small, tiles, pieces):
    from feathercnn_amd import booster
    g, x, w, b, p, layer = _layer(cuda, c, k, h, batch)
    pl = booster.winograd_plan(p)
    n_tiles = -(-pl.columns // 64)
    assert pl.frequency_points * (-(-k // 128)) * n_tiles == tiles
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert 0 < tiles % 256 <= 256 // pieces  # on the MI355X's 256 CUs the geometry is in the row-split class it claims
    # (another CU count puts the launch in another class -- whole tiles, or other pieces: the two comparisons below hold for every class)
    got = layer.Forward(torch.from_numpy(x).to(cuda)).cpu().numpy()
    _, _, _, _, _, small_layer = _layer(cuda, c, k, h, small, wb=(w, b))
    parts = [small_layer.Forward(torch.from_numpy(np.ascontiguousarray(x[i:i + small])).to(cuda)).cpu().numpy() for i in range(0, batch, small)]
    assert np.array_equal(got, np.concatenate(parts)[:batch])
    want = oracle.best().forward(g, x[:3], w, b)
    assert nerr(got[:3], want) <= 1e-4
