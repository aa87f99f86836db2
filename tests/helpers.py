import os

import numpy as np

import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "conv_golden.npz")


def golden_cases():
    """-> list of (name, Geom, batch, forced_algo, x, w, b, y_ref, selected_algo) from the committed reference fixtures."""
    z = np.load(GOLDEN)
    names = sorted({k.split("/")[0] for k in z.files})
    out = []
    for n in names:
        g15 = z[n + "/geom"]
        g = oracle.Geom(*[int(v) for v in g15[:15]])
        out.append((n, g, int(g15[15]), int(g15[16]), z[n + "/x"], z[n + "/w"], z[n + "/b"], z[n + "/y"], int(z[n + "/algo"][0])))
    return out
