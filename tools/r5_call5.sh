#!/bin/bash
# round 5, GPU call 5: which round-5 change broke ResNet-50's parity at the bench configuration (call 4)?  The parity tests per library build,
# then the whole suite on the default build, then the A/B of the M-store policies through bench.py
O=gpurun_out/r5_call5
mkdir -p $O
cp feathercnn_amd/libfeather_hip.so /tmp/orig.so
for v in base small0 k4one nosplit; do
  cp tools/_build/var_$v/libfeather_hip.so feathercnn_amd/libfeather_hip.so
  timeout 600 python -m pytest tests/test_baseline_shapes_gpu.py tests/test_k4_persist_gpu.py tests/test_winograd_f43_gpu.py -q -m gpu > $O/pytest_$v.txt 2>&1
  echo "== $v: $(tail -1 $O/pytest_$v.txt)"
done
cp /tmp/orig.so feathercnn_amd/libfeather_hip.so
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
export VARIANTS="base=;small0=x;small2=x;plainm=x"
NETS="vgg16 resnet50" ROUNDS=5 timeout 1500 bash tools/variant_ab.sh run r5_ab5 > $O/ab5.txt 2>&1
tail -10 $O/ab5.txt | cut -c1-150
