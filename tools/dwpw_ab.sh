#!/bin/bash
# tools/dwpw_ab.sh build | run <out> -- ablation builds of the fused depthwise + pointwise route (FHIP_DWPW_ABLATE bits: 1 no depthwise FMAs,
# 2 no halo loads, 4 centre row only) through tools/dwpw_bench.py: where does the kernel's time go?
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
VARS="${VARS:-0 1 2 3 6 7}"
if [ "$1" = build ]; then
  for v in $VARS; do
    mkdir -p $R/tools/_build/dwpw_$v
    make -s -j8 -C $R/feathercnn_amd/csrc OBJDIR=/tmp/fhip_obj_dwpw_$v OUT=$R/tools/_build/dwpw_$v/libfeather_hip.so EXTRA=$([ $v = 0 ] && echo "" || echo "-D${MACRO:-FHIP_DWPW_ABLATE}=$v")
    echo "built dwpw_$v"
  done
  exit 0
fi
O=$R/gpurun_out/${2:-dwpw_ab}
mkdir -p $O
cp $R/feathercnn_amd/libfeather_hip.so /tmp/libfeather_hip.orig.so
for v in $VARS; do
  cp $R/tools/_build/dwpw_$v/libfeather_hip.so $R/feathercnn_amd/libfeather_hip.so
  echo "== ablation $v" | tee -a $O/ab.txt
  timeout 200 python $R/tools/dwpw_bench.py 2>&1 | tee -a $O/ab.txt
done
cp /tmp/libfeather_hip.orig.so $R/feathercnn_amd/libfeather_hip.so
