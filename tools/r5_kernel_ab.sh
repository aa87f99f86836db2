#!/bin/bash
# tools/r5_kernel_ab.sh -- per-kernel, per-launch-shape durations (rocprofv3 kernel trace, tools/trace_by_grid.py) of library builds, interleaved:
#   KVARIANTS="base xld" NET=vgg16 bash tools/r5_kernel_ab.sh <out tag>      (builds under tools/_build/var_<name>, tools/variant_ab.sh build)
O=gpurun_out/${1:-kernel_ab}
mkdir -p $O
R=$(pwd)
cp feathercnn_amd/libfeather_hip.so /tmp/orig.so
export TMPDIR=/tmp
for round in $(seq 1 ${ROUNDS:-2}); do
for v in ${KVARIANTS:-base}; do
  cp tools/_build/var_$v/libfeather_hip.so feathercnn_amd/libfeather_hip.so
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_${v}_$round -o t -- python $R/bench.py --net ${NET:-vgg16} --steps 30 --no-cpu-baseline --no-steady > $R/$O/tr_${v}_$round.log 2>&1)
  python tools/trace_by_grid.py $O/tr_${v}_$round wino_ gemm_mfma depthwise dwpw stream_gemm > $O/by_grid_${v}_$round.txt 2>&1
  find $O/tr_${v}_$round -name '*kernel_trace.csv' -delete
done
done
cp /tmp/orig.so feathercnn_amd/libfeather_hip.so
