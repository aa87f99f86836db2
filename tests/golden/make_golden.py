#!/usr/bin/env python
"""Generate tests/golden/conv_golden.npz by running the REAL reference (oracle/_ref = FeatherCNN's AVX2 booster compiled from
/root/reference, see oracle/Makefile) on small seeded inputs.  The reference ships no golden vectors, so
these fixtures ARE the known answers; they travel to the GPU box where /root/reference does not exist.

    python tests/golden/make_golden.py          (needs oracle/_ref/libfeather_ref.so -> run where /root/reference exists)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import conv_geom, synth  # noqa: E402

# (name, geom, batch, forced algo or -1): every route of ConvBooster::SelectAlgo plus the edge cases the boundary has
CASES = [
    ("wino_min_9x9", conv_geom(8, 8, 9, 3, 1, 1), 2, -1),
    ("wino_ragged_31x17", conv_geom(8, 12, 31, 3, 1, 1, w=17), 2, -1),
    ("wino_nopad", conv_geom(12, 8, 14, 3, 1, 0), 1, -1),
    ("wino_c48_k64", conv_geom(48, 64, 13, 3, 1, 1), 2, -1),
    ("wino_nobias_norelu", conv_geom(16, 16, 20, 3, 1, 1, bias=0, act=0), 1, -1),
    ("wino_bias_only", conv_geom(16, 16, 20, 3, 1, 1, bias=1, act=0), 1, -1),
    ("im2col_1x1", conv_geom(32, 24, 14, 1, 1, 0), 2, -1),
    ("im2col_1x1_s2", conv_geom(16, 40, 14, 1, 2, 0), 2, -1),
    ("im2col_7x7_s2", conv_geom(3, 16, 40, 7, 2, 3), 1, -1),
    ("im2col_3x3_s2_c3", conv_geom(3, 8, 31, 3, 2, 1), 2, -1),
    ("im2col_small_h", conv_geom(16, 16, 7, 3, 1, 1), 2, -1),
    ("im2col_c_not_mult4", conv_geom(6, 10, 12, 3, 1, 1), 2, -1),
    ("naive_ignores_relu", conv_geom(8, 8, 10, 3, 1, 1), 1, oracle.NAIVE),
    ("dw_s1", conv_geom(16, 16, 28, 3, 1, 1, group=16), 2, -1),
    ("dw_s2", conv_geom(16, 16, 28, 3, 2, 1, group=16), 2, -1),
    ("dw_7x7", conv_geom(32, 32, 7, 3, 1, 1, group=32), 2, -1),
    ("dw_5x5", conv_geom(8, 8, 12, 5, 1, 2, group=8), 1, -1),
    ("dw_global", conv_geom(8, 8, 6, 6, 1, 0, group=8), 2, -1),
]


def main():
    if not oracle.have_ref():
        raise SystemExit("oracle/_ref/libfeather_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    ref = oracle.ref()
    out = {}
    for name, g, batch, algo in CASES:
        x, w, b = synth(g, batch, seed=20260923)
        y = ref.forward(g, x, w, b, algo=algo)
        out[name + "/geom"] = np.array(list(g.arr()) + [batch, algo], np.int32)
        out[name + "/x"] = x
        out[name + "/w"] = w
        out[name + "/b"] = b
        out[name + "/y"] = y
        out[name + "/algo"] = np.array([ref.select_algo(g)], np.int32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(CASES)} cases, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
