#!/bin/bash
# tools/r5_final.sh -- the round 5: the last GPU call: the whole GPU suite, smoke(), the driver's bench command, then the rocprofv3 profiles of the
# same tree for the three metric nets (kernel trace + PMC passes, tools/profile.sh) -- the traffic digests carry the tree's source fingerprint
# (feathercnn_amd/provenance.py), so bench.py attaches them exactly while the kernels are the ones that were profiled.
# Afterwards (here): python tools/collect_profiles.py r05   copies the summaries from gpurun_out/ into profiles/.
O=gpurun_out/r5_final
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -1 $O/smoke.txt
PROF_TIMEOUT=300 bash tools/profile.sh r05_vgg16 --net vgg16 --steps 20 > $O/prof_vgg16.log 2>&1
PROF_TIMEOUT=300 bash tools/profile.sh r05_resnet50 --net resnet50 --steps 20 > $O/prof_resnet50.log 2>&1
PROF_TIMEOUT=300 bash tools/profile.sh r05_mobilenet_v1 --net mobilenet_v1 --steps 20 > $O/prof_mobilenet_v1.log 2>&1
PROF_TIMEOUT=300 bash tools/profile.sh r05_mobilenet_v1_single_stream --net mobilenet_v1 --sub-batches 1 --steps 20 > $O/prof_mobilenet_v1_single_stream.log 2>&1
# the bench LAST: it finds the digests just written (gpurun_out/prof_* copied into profiles/ on the box for this run)
python tools/collect_profiles.py r05 > $O/collect.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --layers-out $O/layers.json > $O/bench.json 2> $O/bench.err
# the N > 1 line as a driver without a launcher would get it: bench.py starts the two ranks itself (one-GPU rehearsal: numbers mean nothing)
FHIP_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-steady > $O/bench_gpus2.json 2> $O/bench_gpus2.err
python -c "import json; d = json.load(open('$O/bench_gpus2.json')); print('gpus2:', d['n_gpus'], d['shard_check'], d['weight_broadcast'], d['config']['other_nets'])" 
tail -2 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
r = d["roofline"]
print(d["metric"], d["value"], d["ms_per_step"], r["frac"], r.get("frac_of_sustained"), "traffic", r.get("traffic"), r.get("traffic_head"), r.get("traffic_stale"), d.get("tree"), d["cpu_baseline"])
for n, v in d["nets"].items():
    print(n, v.get("images_per_s"), v.get("ms_per_step"), v.get("steady_state"))
for n, rs in d["rooflines"].items():
    for r in rs: print("   ", n, r["kernel"][:70], r["frac"], r["ms_per_step"], r.get("layer_frac_min"), r.get("frac_of_tighter_bound"), r.get("hbm_bound_layers"), r.get("frac_survey_8d_formula"), r.get("traffic"), r.get("traffic_stale"))
print("config.other_nets", d["config"]["other_nets"])
print("roofline.also", d["roofline"]["also"])
PY
