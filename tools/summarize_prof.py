#!/usr/bin/env python
"""Summarise rocprofv3 output of tools/profile.sh into a small markdown table (per kernel: calls, avg/total time,
PMC counters averaged per dispatch).  Usage: summarize_prof.py <prof_dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    n = name
    for a, b in (("fhip::", ""), ("void ", ""), ("GemmShape", "Shape"), ("(anonymous namespace)::", "")):
        n = n.replace(a, b)
    return n[:110]


def main(d):
    print(f"# rocprofv3 summary of `{os.path.basename(d.rstrip('/'))}`\n")
    stats = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        print("## kernel-trace --stats (all dispatches of the command, warm-up and per-layer timing passes included)\n")
        print("| kernel | calls | total ms | avg us | min us | max us | % |")
        print("|---|---|---|---|---|---|---|")
        rows = list(csv.DictReader(open(stats[0])))
        for r in rows[:25]:
            print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                  f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
        print()
    # PMC passes
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", ""))
            c = r.get("Counter_Name", "")
            v = float(r.get("Counter_Value", 0) or 0)
            a = agg[k][c]
            a[0] += v
            a[1] += 1
    if agg:
        counters = sorted({c for k in agg for c in agg[k]})
        print("## PMC counters, average per dispatch (separate passes per counter group; FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them)\n")
        print("| kernel | " + " | ".join(counters) + " |")
        print("|---|" + "---|" * len(counters))
        for k in sorted(agg, key=lambda k: -sum(v[0] for v in agg[k].values())):
            print(f"| `{k}` | " + " | ".join(f"{agg[k][c][0] / max(agg[k][c][1], 1):.4g}" if c in agg[k] else "" for c in counters) + " |")


    # derived per-kernel figures: HBM rate from the PMC bytes over the kernel-trace average duration, MFMA utilisation and
    # effective clock from the SQ / GRBM counters (MI355X: 8 XCDs -> GRBM_GUI_ACTIVE / 8 = busy cycles of the dispatch;
    # 1024 SIMDs; SQ_VALU_MFMA_BUSY_CYCLES is summed over SIMDs)
    if agg and stats:
        dur = {short(r["Name"]): float(r["AverageNs"]) for r in rows}
        print("\n## derived per kernel (PMC passes run the same command with --steps 2, so averages are per dispatch)\n")
        print("| kernel | avg us (trace) | HBM bytes / dispatch (2*FETCH + WRITE) | HBM GB/s | frac of 8 TB/s | MFMA util (busy SIMD-cycles / 1024 / cycles) |")
        print("|---|---|---|---|---|---|")
        for k in sorted(agg, key=lambda k: -dur.get(k, 0) * agg[k].get("FETCH_SIZE", [0, 0])[1]):
            if k not in dur or "FETCH_SIZE" not in agg[k] or "WRITE_SIZE" not in agg[k]:
                continue
            f = agg[k]["FETCH_SIZE"][0] / max(agg[k]["FETCH_SIZE"][1], 1)
            w = agg[k]["WRITE_SIZE"][0] / max(agg[k]["WRITE_SIZE"][1], 1)
            by = (2.0 * f + w) * 1024.0
            us = dur[k] / 1e3
            gbs = by / (us * 1e-6) / 1e9 if us > 0 else 0
            cyc = agg[k]["GRBM_GUI_ACTIVE"][0] / max(agg[k]["GRBM_GUI_ACTIVE"][1], 1) / 8.0 if "GRBM_GUI_ACTIVE" in agg[k] else 0
            mf = agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"][0] / max(agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"][1], 1) if "SQ_VALU_MFMA_BUSY_CYCLES" in agg[k] else 0
            util = mf / 1024.0 / cyc if cyc > 0 else 0
            if us < 3:
                continue
            print(f"| `{k}` | {us:.1f} | {by / 1e6:.1f} MB | {gbs:.0f} | {gbs / 8000:.3f} | {util:.3f} |")

    # machine-readable digest for bench.py's roofline.traffic: HBM bytes per launch of the hot kernels.
    # MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
    # stream, so the read side is doubled; WRITE_SIZE is taken as reported; both are KiB.
    import json
    dig = {}
    for k in agg:
        if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
            f = agg[k]["FETCH_SIZE"][0] / max(agg[k]["FETCH_SIZE"][1], 1)
            w = agg[k]["WRITE_SIZE"][0] / max(agg[k]["WRITE_SIZE"][1], 1)
            dig[k] = {"launches_profiled": agg[k]["FETCH_SIZE"][1], "fetch_kib_raw": f, "write_kib_raw": w,
                      "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    # which tree / configuration these counters describe (bench.py --manifest-out, written by the kernel-trace pass of tools/profile.sh):
    # bench.attach_traffic attaches the digest only to a run of the same sources at the same batch and fusion level
    try:
        dig["_meta"] = json.load(open(os.path.join(d, "manifest.json")))
    except (OSError, ValueError):
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from feathercnn_amd import provenance
        dig["_meta"] = provenance.tree_head()
    with open(os.path.join(d, "traffic.json"), "w") as fo:
        json.dump(dig, fo, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
