#!/bin/bash
# round 5, GPU call 6: the whole GPU suite on the cleaned-up defaults (nt for a large M, plain otherwise), size-ruled nt output stores of the 1x1 GEMMs,
# the nets at other input sizes
O=gpurun_out/r5_call6
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -s > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
grep "of the 224-pixel rate" $O/pytest.txt
export VARIANTS="base=;out150=x;out100=x;out1=x"
NETS="resnet50 mobilenet_v1" ROUNDS=5 timeout 1500 bash tools/variant_ab.sh run r5_ab6 > $O/ab6.txt 2>&1
tail -8 $O/ab6.txt | cut -c1-150
timeout 600 python tools/resolution_bench.py > $O/resolution.txt 2>&1
cat $O/resolution.txt | grep -v "^{"
