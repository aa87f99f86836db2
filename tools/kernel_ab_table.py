#!/usr/bin/env python
"""tools/kernel_ab_table.py <gpurun_out/dir> [substr ...] -- table of tools/r5_kernel_ab.sh's by_grid_<build>_<round>.txt files: mean of the per-round
median duration (us) of every (kernel, grid, workgroup), one column per build, and the launches-weighted sum."""
import collections
import glob
import re
import statistics
import sys


def main(d, subs):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    cnt = collections.defaultdict(dict)
    tags = []
    for f in sorted(glob.glob(d + "/by_grid_*.txt")):
        tag = f.split("by_grid_")[1][:-4].rsplit("_", 1)[0]
        if tag not in tags:
            tags.append(tag)
        for ln in open(f):
            m = re.match(r"(.+?)\s+grid\s+(\d+) wg\s+(\d+)\s+n\s+(\d+)\s+median\s+([\d.]+) us", ln)
            if not m or (subs and not any(s in m.group(1) for s in subs)):
                continue
            name = re.sub(r"wino_gemm_glds_kernel<2, 16, 6, \d, (\w+)>", r"wino_gemm_glds_kernel<\1>", m.group(1))
            name = re.sub(r"WinoGemmPolicyT<\d>", "WinoGemmPolicyT", name)
            key = (name[:64], m.group(2), m.group(3))
            res[key][tag].append(float(m.group(5)))
            cnt[key][tag] = int(m.group(4))
    print("%-84s %5s" % ("kernel grid wg", "n") + "".join("%9s" % t for t in tags))
    tot = collections.defaultdict(float)
    for k, v in sorted(res.items()):
        n = max(cnt[k].values())
        print("%-84s %5d" % (" ".join(k), n) + "".join("%9.2f" % statistics.mean(v[t]) if v[t] else "%9s" % "-" for t in tags))
        for t in tags:
            if v[t]:
                tot[t] += statistics.mean(v[t]) * n
    print("%-90s" % "launch-weighted sum (ms over the traced run)" + "".join("%9.2f" % (tot[t] / 1e3) for t in tags))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
