#!/usr/bin/env python3
"""tools/kernel_resources.py <file.hip> [name filter] [extra hipcc flags...] -- VGPRs / scratch / occupancy / LDS of every kernel of a
translation unit (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles, no GPU needed)."""
import re
import subprocess
import sys

R = "/root/repo"


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    extra = sys.argv[3:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{R}/include", f"-I{R}/feathercnn_amd/csrc", f"-I{R}/tools",
           f"-I{R}/tools/experiments", *extra, "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    pats = {"vgpr": r" VGPRs: (\d+)", "agpr": r"AGPRs: (\d+)", "scratch": r"ScratchSize \[bytes/lane\]: (\d+)", "occ": r"Occupancy \[waves/SIMD\]: (\d+)",
            "lds": r"LDS Size \[bytes/block\]: (\d+)"}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if "error" in line:
            print(line)
        for k, p in pats.items():
            m = re.search(p, line)
            if m and cur is not None:
                cur[k] = m.group(1)
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        if flt and flt not in n:
            continue
        print(f"{n[:120]:120s} vgpr {r.get('vgpr', '?'):>4s} agpr {r.get('agpr', '?'):>4s} scratch {r.get('scratch', '?'):>4s} occ {r.get('occ', '?'):>2s} lds {r.get('lds', '?'):>6s}")


if __name__ == "__main__":
    main()
