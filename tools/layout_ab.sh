#!/bin/bash
# tools/layout_ab.sh build | run <out-dir> -- A/B of the column block BP of the Winograd scratch tensors (wino_layout.h).
#   build (here, no GPU): one library per BP under tools/_build/bp_<BP>/ (FHIP_WINO_BP overrides winograd_f63.hip's rule)
#   run   (GPU box):      every variant through bench.py on VGG-16 / ResNet-50, interleaved twice; prints the Winograd stage times
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
BPS="${BPS:-0 256 512 1024 2048 4096}"
if [ "$1" = build ]; then
  for bp in $BPS; do
    mkdir -p $R/tools/_build/bp_$bp
    make -s -j8 -C $R/feathercnn_amd/csrc OBJDIR=/tmp/fhip_obj_bp_$bp OUT=$R/tools/_build/bp_$bp/libfeather_hip.so EXTRA=-DFHIP_WINO_BP=$bp
    echo "built bp_$bp"
  done
  exit 0
fi
O=$R/gpurun_out/${2:-layout_ab}
mkdir -p $O
cp $R/feathercnn_amd/libfeather_hip.so /tmp/libfeather_hip.orig.so
for round in $(seq 1 ${ROUNDS:-2}); do
  for bp in $BPS; do
    cp $R/tools/_build/bp_$bp/libfeather_hip.so $R/feathercnn_amd/libfeather_hip.so
    for net in ${NETS:-vgg16 resnet50}; do
      timeout 300 python $R/bench.py --net $net --steps 30 --warmup 5 --no-cpu-baseline --no-steady > $O/${net}_bp${bp}_r$round.json 2> $O/${net}_bp${bp}_r$round.err || echo "FAILED $net bp $bp"
    done
  done
done
cp /tmp/libfeather_hip.orig.so $R/feathercnn_amd/libfeather_hip.so
python - <<PY
import glob, json, os, re, collections
rows = collections.defaultdict(list)
for f in sorted(glob.glob("$O/*_bp*_r*.json")):
    m = re.match(r"(.*)_bp(\w+?)_r(\d)\.json", os.path.basename(f))
    try:
        d = json.load(open(f))
    except Exception:
        continue
    st = d.get("stage_ms_per_step", {})
    rows[(m.group(1), m.group(2))].append((d["value"], st.get("wino_input"), st.get("wino_gemm"), st.get("wino_chain"), st.get("wino_output")))
for (net, bp), v in sorted(rows.items()):
    print(f"{net:9s} BP {bp:>6s}: " + "   ".join(f"{a:8.0f} img/s  in {b} gemm {c} chain {e} out {o}" for a, b, c, e, o in v))
PY
