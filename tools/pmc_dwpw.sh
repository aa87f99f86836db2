#!/bin/bash
# PMC counters of the fused depthwise + pointwise kernels under tools/dwpw_bench.py (separate passes; no trace domains besides kernel-trace)
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/${1:-pmc_dwpw}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_SALU"; do
  name=$(echo $pass | tr ' ' '+' | cut -c1-40)
  REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_$name -o pmc -- python $R/tools/dwpw_bench.py > $O/pmc_$name.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "dwpw_band" not in k and "ConvGemmPolicy<3" not in k and "ConvGemmPolicy<4" not in k: continue
        key = (k.replace("fhip::", "")[:70], r.get("Grid_Size", ""))
        a = agg[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"] or 0); a[1] += 1
for key, cs in sorted(agg.items()):
    print(key)
    for c, (v, n) in sorted(cs.items()):
        print(f"    {c:34s} {v / max(n, 1):14.4g}  (n={n})")
PY
