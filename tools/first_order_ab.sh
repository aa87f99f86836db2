#!/bin/bash
# tools/first_order_ab.sh -- block order of wino_input_from_first_staged_kernel (FHIP_FIRST_ORDER 0 / 1): time (tools/first_bench.py, interleaved) and
# HBM-side fetch (rocprofv3 --pmc FETCH_SIZE) per build.   Build first:  make -C feathercnn_amd/csrc OBJDIR=$PWD/tools/_build/obj_order1 \
#   OUT=$PWD/tools/_build/libfeather_hip_order1.so EXTRA=-DFHIP_FIRST_ORDER=1
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/first_order_ab
mkdir -p $O
for round in 1 2 3; do
  for v in 0 1; do
    lib=$R/feathercnn_amd/libfeather_hip.so; [ $v = 1 ] && lib=$R/tools/_build/libfeather_hip_order1.so
    echo "== order $v round $round" | tee -a $O/ab.txt
    (cd $R && FEATHER_HIP_LIB=$lib timeout 200 python tools/first_bench.py 32 15 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt)
  done
done
export TMPDIR=/tmp
cd /tmp
for v in 0 1; do
  lib=$R/feathercnn_amd/libfeather_hip.so; [ $v = 1 ] && lib=$R/tools/_build/libfeather_hip_order1.so
  for c in FETCH_SIZE WRITE_SIZE; do
    FEATHER_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${v}_$c -o pmc -- python $R/tools/first_bench.py 32 3 > $O/pmc_${v}_$c.log 2>&1
  done
done
cd $R
python - <<PY | tee -a $O/ab.txt
import csv, glob
for v in (0, 1):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot, n = 0.0, 0
        for f in glob.glob(f"$O/pmc_{v}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "wino_input_from_first_staged" in r.get("Kernel_Name", "") and r["Counter_Name"] == c:
                    tot += float(r["Counter_Value"]); n += 1
        print(f"order {v}: {c} per launch of wino_input_from_first_staged_kernel = {tot / max(n, 1) / 1024:.1f} MiB as reported (n = {n}; FETCH_SIZE: double it, MI355X_MICROARCH.md)")
PY
