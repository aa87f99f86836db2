#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace, by grid) of VGG-16 b32 for a few column-block variants built by tools/layout_ab.sh
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/${1:-layout_trace}
mkdir -p $O
export TMPDIR=/tmp
cp $R/feathercnn_amd/libfeather_hip.so /tmp/libfeather_hip.orig.so
for bp in ${BPS:-0 512 1024}; do
  cp $R/tools/_build/bp_$bp/libfeather_hip.so $R/feathercnn_amd/libfeather_hip.so
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$bp -o tr -- python $R/bench.py --net ${NET:-vgg16} --steps 30 --warmup 5 --no-cpu-baseline --no-steady > $O/tr_$bp.log 2>&1)
  python $R/tools/trace_by_grid.py $O/tr_$bp wino_ gemm_mfma > $O/by_grid_$bp.txt 2>&1
  find $O/tr_$bp -name '*.csv' -size +4M -delete
done
cp /tmp/libfeather_hip.orig.so $R/feathercnn_amd/libfeather_hip.so
