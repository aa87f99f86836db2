#!/bin/bash
# one GPU call: new depthwise routes (tests + micro-benchmark cold / cache-resident), residual-epilogue A/B, the nets they touch
O=gpurun_out/r3j
mkdir -p $O
timeout 400 python -m pytest tests/test_depthwise_flat_gpu.py tests/test_parity_gpu.py tests/test_baseline_shapes_gpu.py tests/test_net_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 150 tools/_build/dw_bench 10 3 1 > $O/dw_cold.txt 2>&1
timeout 150 tools/_build/dw_bench 10 3 0 > $O/dw_hot.txt 2>&1
grep -E "^conv|product|direct |band|flat cp(36|72|108)" $O/dw_cold.txt | grep -B0 -A6 -E "conv2_dw|conv4_dw|conv14_dw" | cut -c1-100
tail -2 $O/dw_cold.txt; tail -2 $O/dw_hot.txt
PROBE_RESIDUAL=1 timeout 200 tools/_build/r50_probe 20 > $O/probe_res.txt 2>&1
grep -E "residual add|product" $O/probe_res.txt | cut -c1-160
for n in mobilenet_v1 resnet50; do timeout 200 python bench.py --net $n --no-cpu-baseline > $O/$n.json 2> $O/$n.err; done
timeout 200 python bench.py --net mobilenet_v1 --sub-batches 1 --no-cpu-baseline > $O/mobilenet_v1_ss.json 2> $O/mb_ss.err
python - <<PY
import json
for f in ("mobilenet_v1", "mobilenet_v1_ss", "resnet50"):
    try:
        d = json.load(open("$O/%s.json" % f))
    except Exception as e:
        print(f, "failed", e); continue
    print(f, d["value"], d["ms_per_step"], d["nets"][d["config"]["net"]].get("steady_state"))
    for n, rs in d["rooflines"].items():
        for r in rs: print("   ", r["kernel"][:60], r["frac"], r["ms_per_step"], r.get("layer_frac_min"))
PY
