#!/bin/bash
# tools/variant_ab.sh build | run <out-dir> -- A/B of library builds that differ by compile-time macros, through bench.py.
#   VARIANTS="name=flags;name=flags..."  (flags may be empty)   NETS="vgg16 resnet50"   ROUNDS=3
#   build (here, no GPU): tools/_build/var_<name>/libfeather_hip.so     run (GPU box): interleaved rounds, prints img/s + Winograd stage times
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
IFS=';' read -ra VARS <<< "${VARIANTS:-base=}"
if [ "$1" = build ]; then
  for v in "${VARS[@]}"; do
    name=${v%%=*}; flags=${v#*=}
    mkdir -p $R/tools/_build/var_$name
    make -s -j8 -C $R/feathercnn_amd/csrc OBJDIR=/tmp/fhip_obj_var_$name OUT=$R/tools/_build/var_$name/libfeather_hip.so EXTRA="$flags"
    echo "built var_$name ($flags)"
  done
  exit 0
fi
O=$R/gpurun_out/${2:-variant_ab}
mkdir -p $O
cp $R/feathercnn_amd/libfeather_hip.so /tmp/libfeather_hip.orig.so
for round in $(seq 1 ${ROUNDS:-3}); do
  for v in "${VARS[@]}"; do
    name=${v%%=*}
    cp $R/tools/_build/var_$name/libfeather_hip.so $R/feathercnn_amd/libfeather_hip.so
    for net in ${NETS:-vgg16}; do
      timeout 300 python $R/bench.py --net $net --steps 30 --warmup 5 --no-cpu-baseline --no-steady > $O/${net}_${name}_r$round.json 2> $O/${net}_${name}_r$round.err || echo "FAILED $net $name"
    done
  done
done
cp /tmp/libfeather_hip.orig.so $R/feathercnn_amd/libfeather_hip.so
python - <<PY
import glob, json, os, re, collections
rows = collections.defaultdict(list)
for f in sorted(glob.glob("$O/*_r*.json")):
    m = re.match(r"([a-z0-9_]+?)_(\w+)_r(\d)\.json", os.path.basename(f))
    for net in ("mobilenet_v1", "resnet50", "vgg16"):
        if os.path.basename(f).startswith(net + "_"):
            name = os.path.basename(f)[len(net) + 1:].rsplit("_r", 1)[0]
            break
    try:
        d = json.load(open(f))
    except Exception:
        continue
    st = d.get("stage_ms_per_step", {})
    rows[(net, name)].append((d["value"], st.get("wino_input"), st.get("wino_gemm"), st.get("wino_chain"), st.get("wino_output"), st.get("igemm")))
for (net, name), v in sorted(rows.items()):
    avg = sum(a[0] for a in v) / len(v)
    print(f"{net:9s} {name:12s} avg {avg:8.0f} img/s | " + "  ".join(f"{a:6.0f} in {b} gemm {c} chain {e} out {o} igemm {g}" for a, b, c, e, o, g in v))
PY
