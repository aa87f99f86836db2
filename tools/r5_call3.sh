#!/bin/bash
# round 5, GPU call 3: M-store policy by the size of M (5 interleaved rounds), the row split of the last tiles (isolated and in ResNet-50)
O=gpurun_out/r5_call3
mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_winograd_f43_gpu.py tests/test_baseline_shapes_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
GEMM_SPLIT=1 GEMM_BATCHES=36 timeout 300 tools/_build/gemm_bench 20 > $O/gemm_split36.txt 2>&1
GEMM_SPLIT=1 timeout 300 tools/_build/gemm_bench 20 > $O/gemm_split64.txt 2>&1
cat $O/gemm_split36.txt | head -20
export VARIANTS="base=;nosplit=x;ntbig=x;sc1big=x;ntbig100=x;sc1all=x;ntsc1=x"
NETS="vgg16 resnet50" ROUNDS=5 timeout 1500 bash tools/variant_ab.sh run r5_nt_ab3 > $O/nt_ab3.txt 2>&1
tail -16 $O/nt_ab3.txt
