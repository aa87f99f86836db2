#!/bin/bash
# round 5, GPU call 4: the GPU suite on the new defaults, A/B of each round-5 change through bench.py, the band kernel's fabric traffic by item order
O=gpurun_out/r5_call4
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
export VARIANTS="base=;k4one=x;band0=x;plainm=x;vnt0=x"
NETS="vgg16 resnet50 mobilenet_v1" ROUNDS=4 timeout 1500 bash tools/variant_ab.sh run r5_ab4 > $O/ab4.txt 2>&1
tail -16 $O/ab4.txt | cut -c1-200
R=$(pwd)
cp feathercnn_amd/libfeather_hip.so /tmp/orig.so
export TMPDIR=/tmp
for v in base band0; do
  cp tools/_build/var_$v/libfeather_hip.so feathercnn_amd/libfeather_hip.so
  REPS=10 timeout 200 python tools/dwpw_bench.py > $O/dwpw_$v.txt 2>&1
  for pass in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/$O/pmc_${v}_$pass -o pmc -- python $R/tools/dwpw_bench.py > $R/$O/pmc_${v}_$pass.log 2>&1)
  done
done
cp /tmp/orig.so feathercnn_amd/libfeather_hip.so
python - <<PY
import csv, glob, collections
for v in ("base", "band0"):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob("$O/pmc_%s_*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "dwpw_band" not in k: continue
            a = agg[k.replace("fhip::", "")[:60]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"] or 0); a[1] += 1
    for k, cs in agg.items():
        print(v, k, {c: (round(x / max(n, 1)), n) for c, (x, n) in cs.items()})
PY
tail -8 $O/dwpw_base.txt; tail -8 $O/dwpw_band0.txt
find $O -name '*kernel_trace.csv' -size +4M -delete
