"""Generates tests/golden/net_golden.npz from the REAL reference feather::Net (oracle/_ref/libfeather_net_ref.so, built
from /root/reference by oracle/Makefile).  Run in the build container only:  python tests/golden/make_net_golden.py

Contents: the tiny_allsorts model files themselves (18 KB, every registered layer type), two seeded input images, and the
reference's blobs after every interesting layer; plus the reference's class probabilities for the seeded SqueezeNet-v1.1
(the model is regenerated from its seed at test time and pinned by the SHA-256 of its .bin)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from feathercnn_amd import model_zoo  # noqa: E402
from oracle import netcheck  # noqa: E402

TINY_BLOBS = ["relu1", "conv2_relu", "dw_relu", "scale_b", "relu_sum", "pool1", "relu_c", "pool_d", "cat", "drop", "gap",
              "relu_fc", "fc2", "prob"]


def main():
    out = {}
    p, b, i, o = model_zoo.tiny_allsorts()
    x = np.random.default_rng(42).uniform(-1, 1, (2, 3, 20, 20)).astype(np.float32)
    ref = netcheck.RefNet(p, b)
    out["tiny/param"] = np.frombuffer(p, np.uint8)
    out["tiny/bin"] = np.frombuffer(b, np.uint8)
    out["tiny/x"] = x
    for name in TINY_BLOBS:
        out["tiny/blob/" + name] = ref.run(i, x, name)
    ref.close()

    p, b, i, o = model_zoo.squeezenet_v11()
    x = np.random.default_rng(43).uniform(-1, 1, (2, 3, 224, 224)).astype(np.float32)
    ref = netcheck.RefNet(p, b)
    out["squeezenet/bin_sha256"] = np.frombuffer(hashlib.sha256(b).digest(), np.uint8)
    out["squeezenet/prob"] = ref.run(i, x, o)
    out["squeezenet/fire5"] = ref.run(i, x, "fire5_concat")[:, :8]
    ref.close()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "net_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
