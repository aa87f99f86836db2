#!/bin/bash
# one GPU call: the GPU suite + smoke + the driver's bench command with the per-layer tables
O=gpurun_out/${1:-r4a}
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -2 $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --layers-out $O/layers.json > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print(d["metric"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_of_sustained"), d["roofline"].get("traffic"), d["roofline"].get("traffic_stale"), d.get("tree"))
for n, v in d["nets"].items():
    print(n, v.get("images_per_s"), v.get("ms_per_step"), v.get("steady_state"))
for n, rs in d["rooflines"].items():
    for r in rs: print("   ", n, r["kernel"][:70], r["frac"], r["ms_per_step"], r.get("layer_frac_min"))
PY
