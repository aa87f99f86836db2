#!/bin/bash
O=gpurun_out/r3r
mkdir -p $O
timeout 100 tools/_build/ip_stream_bench > $O/ip_stream_bench.txt 2>&1
cat $O/ip_stream_bench.txt
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_net_gpu.py tests/test_baseline_shapes_gpu.py tests/test_fuzz_gpu.py tests/test_net_fuzz_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 200 python bench.py --net vgg16 --no-cpu-baseline --layers-out $O/vgg_layers.json > $O/vgg16.json 2> $O/vgg16.err
python - <<PY
import json
d = json.load(open("$O/vgg16.json"))
print(d["value"], d["ms_per_step"], d["nets"]["vgg16"].get("steady_state"), d["roofline"]["frac"])
t = json.load(open("$O/vgg_layers.json"))["tables"]["vgg16"]
print([(r["layer"], r["ms"]) for r in t if r["type"] == "InnerProduct"])
PY
