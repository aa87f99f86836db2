#!/usr/bin/env python
"""tools/collect_profiles.py <round tag> -- copy what tools/profile.sh wrote under gpurun_out/prof_<tag>_<net>/ (summary.md, traffic.json,
by_grid.txt, the kernel-stats csv, the manifest) into profiles/<tag>_<net>/ (tracked; gpurun_out/ is scratch)."""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_*"))):
    name = os.path.basename(d)[len("prof_"):]
    dst = os.path.join(ROOT, "profiles", name)
    os.makedirs(dst, exist_ok=True)
    n = 0
    for f in ("summary.md", "traffic.json", "by_grid.txt", "manifest.json"):
        if os.path.exists(os.path.join(d, f)):
            shutil.copy(os.path.join(d, f), os.path.join(dst, f))
            n += 1
    stats = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
        n += 1
    print(f"{name}: {n} files -> profiles/{name}")
