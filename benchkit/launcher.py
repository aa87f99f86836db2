"""`python bench.py --gpus N` without a launcher: start the N ranks."""
from __future__ import annotations

import os
import sys


def launch_ranks(n, script=None):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks here -- this process becomes
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`,
    one rank per GPU over RCCL, exactly the command the driver contract names -- and hand its exit status on.  Rank 0's JSON line goes to
    this process's stdout unchanged.  Fails loudly when the node has fewer than N GPUs (FHIP_BENCH_SHARE_GPU=1: the one-GPU rehearsal,
    every rank on cuda:0 over gloo; its numbers mean nothing)."""
    import socket
    import subprocess

    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get("FHIP_BENCH_SHARE_GPU") != "1" and have < n:
        raise SystemExit(f"bench.py --gpus {n}: this node shows {have} GPU(s); one rank per GPU is the only mode that measures anything "
                         "(FHIP_BENCH_SHARE_GPU=1 rehearses the N-rank path on one GPU)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(script or sys.argv[0])] + sys.argv[1:]
    print(f"bench: --gpus {n} without a launcher: starting {n} ranks ({' '.join(cmd[1:9])} ...)", file=sys.stderr, flush=True)
    # rank 0's JSON line is the ONLY thing this process prints on stdout; whatever else the ranks or their libraries write there (gloo's
    # connection notes in the one-GPU rehearsal, for one) goes to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        out = sys.stdout if line.startswith("{") else sys.stderr
        out.write(line)
        out.flush()
    return proc.wait()
