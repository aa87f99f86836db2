#!/bin/bash
# tools/r6_check.sh -- round 6 check call: GPU suite, the driver's bench command (line size!), summary of line + detail.
O=gpurun_out/${1:-r6_check}
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_detail.json --layers-out $O/layers.json > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
python - <<PY
import json
raw = open("$O/bench.json").read()
print("line bytes:", len(raw.encode()))
d = json.loads(raw)
print(json.dumps(d, indent=None)[:6000])
PY
