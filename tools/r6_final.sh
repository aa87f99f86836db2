#!/bin/bash
# tools/r6_final.sh -- round 6, the final GPU call on the final tree: the whole GPU suite, smoke(), the rocprofv3 profiles of the DRIVER's bench command
# (--gpus 1 --steps 20 --warmup 5) per metric net (kernel trace + PMC passes, tools/profile.sh; the traffic digests carry the tree's source fingerprint,
# feathercnn_amd/provenance.py, so bench.py attaches them exactly while the kernels are the ones that were profiled), the kernel trace of the whole
# driver command, then the driver's command itself, the 8-rank rehearsal line and the resolution tables.
# Afterwards (here): python tools/collect_profiles.py r06   copies the summaries from gpurun_out/ into profiles/.
O=gpurun_out/r6_final
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -1 $O/smoke.txt
DRV="--gpus 1 --steps 20 --warmup 5"
PROF_TIMEOUT=300 bash tools/profile.sh r06_vgg16 --net vgg16 $DRV > $O/prof_vgg16.log 2>&1
PROF_TIMEOUT=300 bash tools/profile.sh r06_resnet50 --net resnet50 $DRV > $O/prof_resnet50.log 2>&1
PROF_TIMEOUT=300 bash tools/profile.sh r06_mobilenet_v1 --net mobilenet_v1 $DRV > $O/prof_mobilenet_v1.log 2>&1
PROF_TIMEOUT=300 bash tools/profile.sh r06_mobilenet_v1_single_stream --net mobilenet_v1 --sub-batches 1 $DRV > $O/prof_mobilenet_v1_single_stream.log 2>&1
# the whole driver command under the kernel trace (all nets in one process, as the driver runs it)
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_r06_driver_cmd/trace -o trace -- python $OLDPWD/bench.py $DRV --no-cpu-baseline --detail-out $OLDPWD/$O/driver_cmd_detail.json > $OLDPWD/$O/driver_cmd_trace.log 2>&1)
python tools/summarize_prof.py gpurun_out/prof_r06_driver_cmd > gpurun_out/prof_r06_driver_cmd/summary.md 2>&1
find gpurun_out/prof_r06_driver_cmd -name '*kernel_trace.csv' -size +8M -delete
# the bench LAST: it finds the digests just written (gpurun_out/prof_* copied into profiles/ on the box for this run)
python tools/collect_profiles.py r06 > $O/collect.txt 2>&1
timeout 900 python bench.py $DRV --detail-out $O/bench_detail.json --layers-out $O/layers.json > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
# the N = 8 line as a driver without a launcher would get it: bench.py starts the eight ranks itself (one-GPU rehearsal: numbers mean nothing)
FHIP_BENCH_SHARE_GPU=1 timeout 1500 python bench.py --gpus 8 --steps 3 --warmup 1 --no-steady --detail-out $O/bench_gpus8_detail.json > $O/bench_gpus8.json 2> $O/bench_gpus8.err
timeout 600 python tools/resolution_bench.py > $O/resolution_bench.txt 2>&1
timeout 600 python tools/resolution_bench.py iso > $O/resolution_iso.txt 2>&1
python - <<PY
import json
for f in ("$O/bench.json", "$O/bench_gpus8.json"):
    raw = open(f).read().strip()
    d = json.loads(raw)
    print(f, "line bytes", len(raw.encode()), "n_gpus", d["n_gpus"], "value", d["value"], "ms", d["ms_per_step"])
    r = d["roofline"]
    print("  roofline", r["frac"], r.get("frac_min"), r.get("frac_max"), "traffic", r.get("traffic"), r.get("traffic_stale"), "cpu", d.get("cpu_baseline"), "shard", d.get("shard_check"))
    print("  other_nets", json.dumps(d["config"]["other_nets"]))
PY
