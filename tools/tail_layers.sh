#!/bin/bash
# tools/tail_layers.sh -- per-layer HIP-event tables of two library variants (tools/_build/var_<a|b>), interleaved, for tools/variant_ab-style A/Bs
# that need to know WHICH layers moved.  usage: tail_layers.sh <varA> <varB> "<nets>"
R=$(cd "$(dirname "$0")/.." && pwd)
cp $R/feathercnn_amd/libfeather_hip.so /tmp/o.so
for v in $1 $2 $1 $2; do
  cp $R/tools/_build/var_$v/libfeather_hip.so $R/feathercnn_amd/libfeather_hip.so
  for net in $3; do
    timeout 200 python $R/bench.py --net $net --steps 20 --warmup 5 --no-cpu-baseline --no-steady --layers-out $R/gpurun_out/vl_${v}_${net}_$RANDOM.json > /dev/null 2>&1
  done
done
cp /tmp/o.so $R/feathercnn_amd/libfeather_hip.so
ls $R/gpurun_out | grep -c "^vl_"
