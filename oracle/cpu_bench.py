"""oracle.cpu_bench -- TEST / BENCH INFRASTRUCTURE ONLY: the CPU-baseline leg of bench.py (SURVEY.md 8(d) "CPU baseline timing").

The REAL reference runtime (feather::Net + AVX2 booster, oracle/_ref/libfeather_net_ref.so, compiled from /root/reference) is
single-threaded on AVX, so the multi-core figure is P independent single-thread copies pinned to distinct cores,
images/s summed.  This helper

  * loads the model ONCE (LoadParam / LoadWeights / Init + one forward), then fork()s the P workers: the weights and packed kernels
    are shared copy-on-write, so P processes do not hold P copies of a 550 MB model (the round-2 leg spent its time and the host's
    memory bandwidth on exactly that);
  * sweeps P over an ascending list, each worker doing 1 untimed warm-up, then -- behind a start line all P workers wait at, so that the timed
    forwards really run P at a time -- `reps` (>= 3) individually timed forwards
    (clock_gettime inside the shim, like the reference's helper.cpp:89-98);  aggregate(P) = sum over workers of 1 / mean(times);
  * stops the sweep when the next P would not fit the time budget (predicted from the previous P), or when the aggregate has fallen
    under half of the best so far (past the knee more processes only thrash the memory system), and says which were skipped; a worker
    that runs into the deadline stops after the forward it is in (never fewer than one timed forward);
  * prints ONE JSON object on stdout.

It runs in a fresh interpreter WITHOUT torch / HIP (fork() from a process with GPU runtime threads is not safe), which is why it is a
separate module:  python -m oracle.cpu_bench --param m.param --bin m.bin --input data --output prob [--procs 1,16,64,128,256]
"""
from __future__ import annotations

import argparse
import json
import os
import struct
import sys
import time


def worker(ref, core, reps, deadline, wfd, ready_w, go_r):
    try:
        os.sched_setaffinity(0, {core})
    except OSError:
        pass
    try:
        ref.time_each(1, 0)  # the untimed warm-up (it also takes this process's copy-on-write page faults)
        # start line: every worker reports in and waits until all P have warmed up, so that the TIMED forwards of all workers run at the same
        # time (fork()ing 256 workers takes seconds: without the line the first ones would time their forwards on a nearly idle host)
        os.write(ready_w, b"r")
        os.close(ready_w)  # every worker drops its write end as soon as it has reported: the parent's EOF then means "all reported or died"
        ready_w = -1
        os.read(go_r, 1)  # returns (EOF) when the parent closes the write end
        secs = ref.time_each(0, 1)
        while len(secs) < reps and time.monotonic() < deadline:
            secs += ref.time_each(0, 1)
        os.write(wfd, struct.pack(f"<i{len(secs)}d", len(secs), *secs))
    finally:
        if ready_w >= 0:  # died before reporting in (exception in the warm-up): close the end so the parent does not wait for this worker
            try:
                os.close(ready_w)
            except OSError:
                pass
        os._exit(0)


def read_exact(fd, n):
    buf = b""
    while len(buf) < n:
        chunk = os.read(fd, n - len(buf))
        if not chunk:
            return None
        buf += chunk
    return buf


def run_procs(ref, cores, procs, reps, deadline):
    """fork `procs` workers (worker k on cores[k % len]), each: 1 warm-up + up to `reps` timed forwards (fewer only if the deadline
    passes; never fewer than one) -> list of per-worker time lists, wall seconds."""
    pipes = []
    t0 = time.perf_counter()
    ready_r, ready_w = os.pipe()
    go_r, go_w = os.pipe()
    for k in range(procs):
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            os.close(go_w)  # only the parent may hold the write end: closing it releases everybody
            os.close(ready_r)
            worker(ref, cores[k % len(cores)], reps, deadline, w, ready_w, go_r)
        os.close(w)
        pipes.append((pid, r))
    os.close(ready_w)
    os.close(go_r)
    got = 0
    # all warmed up.  A worker closes its write end right after reporting (or when it dies, OOM kill included), so EOF here means "every
    # worker has reported or is gone"; the select timeout is the belt to that brace (a worker stuck in its warm-up past the deadline)
    import select
    limit = max(deadline - time.monotonic(), 0.0) + 120.0
    t_wait = time.monotonic()
    while got < procs:
        left = limit - (time.monotonic() - t_wait)
        if left <= 0 or not select.select([ready_r], [], [], left)[0]:
            break
        chunk = os.read(ready_r, procs - got)
        if not chunk:
            break
        got += len(chunk)
    os.close(ready_r)
    t_go = time.perf_counter()
    os.close(go_w)  # the start line
    out = []
    for pid, r in pipes:
        head = read_exact(r, 4)
        body = read_exact(r, 8 * struct.unpack("<i", head)[0]) if head else None
        os.close(r)
        os.waitpid(pid, 0)
        if body:
            out.append(list(struct.unpack(f"<{len(body) // 8}d", body)))
    run_procs.timed_wall = time.perf_counter() - t_go  # wall time of the timed part (all workers, from the start line)
    return out, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--param", required=True)
    ap.add_argument("--bin", required=True)
    ap.add_argument("--input", required=True)
    ap.add_argument("--output", required=True)
    ap.add_argument("--shape", default="3,224,224")
    ap.add_argument("--procs", default="", help="ascending list of process counts (default: 1,8,16,32,64,128,<host cores>)")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--budget", type=float, default=30.0, help="seconds for the whole sweep (model load not included)")
    a = ap.parse_args()
    os.environ["OMP_NUM_THREADS"] = "1"
    import numpy as np

    from oracle import netcheck
    cores = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    ncpu = len(cores)
    plist = [int(v) for v in a.procs.split(",") if v] if a.procs else [1, 8, 16, 32, 64, 128, ncpu]
    plist = sorted({min(p, ncpu) for p in plist if p >= 1})
    reps = max(3, a.reps)
    shape = tuple(int(v) for v in a.shape.split(","))
    t0 = time.perf_counter()
    ref = netcheck.RefNet(None, None, a.param, a.bin)
    x = np.random.default_rng(7).uniform(-1, 1, (1,) + shape).astype(np.float32)
    ref.run(a.input, x, a.output)  # Reshape + Init + first forward, untimed
    load_s = time.perf_counter() - t0
    sweep, skipped = [], []
    t_start = time.perf_counter()
    prev = None  # (procs, wall seconds of that sweep point)
    for p in plist:
        elapsed = time.perf_counter() - t_start
        if prev is not None and elapsed + prev[1] * max(1.0, (p / prev[0]) ** 0.5) > a.budget:
            skipped.append(p)  # predicted from the previous point: memory-bound beyond a few cores, so a forward slows down as P grows
            continue
        times, wall = run_procs(ref, cores, p, reps, time.monotonic() + max(a.budget - elapsed, 1.0))
        if len(times) != p:
            skipped.append(p)
            continue
        means = [sum(t) / len(t) for t in times]
        sweep.append({"procs": p, "images_per_s": round(sum(1.0 / m for m in means), 3),
                      "wall_images_per_s": round(sum(len(t) for t in times) / run_procs.timed_wall, 3), "mean_forward_s": round(sum(means) / len(means), 4),
                      "best_forward_s": round(min(min(t) for t in times), 4), "timed_forwards_per_worker": min(len(t) for t in times),
                      "wall_s": round(wall, 2)})
        prev = (p, wall)
        if sweep[-1]["images_per_s"] < 0.5 * max(r["images_per_s"] for r in sweep):
            # past the knee: more processes only thrash the memory system (each forward streams the whole model and its activations)
            skipped += [q for q in plist if q > p]
            break
    nproc_point = next((r for r in sweep if r["procs"] == ncpu), None)
    if nproc_point is None and not os.environ.get("FHIP_CPU_BENCH_NO_NPROC"):
        # SURVEY.md 8(d) names P = nproc: keep that figure in the line even when the sweep stopped before it (past the knee it is far from
        # the best point).  One warm-up + ONE timed forward per worker: the deadline is already over when the workers start, and a worker
        # never does fewer than one timed forward.
        times, wall = run_procs(ref, cores, ncpu, 1, time.monotonic())
        if len(times) == ncpu:
            means = [sum(t) / len(t) for t in times]
            nproc_point = {"procs": ncpu, "images_per_s": round(sum(1.0 / m for m in means), 3),
                           "wall_images_per_s": round(sum(len(t) for t in times) / run_procs.timed_wall, 3), "mean_forward_s": round(sum(means) / len(means), 4),
                           "best_forward_s": round(min(min(t) for t in times), 4), "timed_forwards_per_worker": min(len(t) for t in times),
                           "wall_s": round(wall, 2), "outside_sweep": True}
    ref.close()
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    best = max(sweep, key=lambda r: r["images_per_s"]) if sweep else None
    json.dump({"sweep": sweep, "skipped": skipped, "best": best, "nproc_point": nproc_point, "reps": reps, "warmup": 1, "host_cores": ncpu, "cpu_model": model,
               "load_s": round(load_s, 2), "sweep_s": round(time.perf_counter() - t_start, 2), "budget_s": a.budget}, sys.stdout)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
