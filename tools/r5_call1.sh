#!/bin/bash
# round 5, GPU call 1: the GPU suite (incl. bench.py --gpus 2 started without a launcher), the non-temporal-hint A/B through bench.py, the
# isolated tile-GEMM experiments (tools/gemm_bench.hip GEMM_R5 / GEMM_TAIL)
O=gpurun_out/r5_call1
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
export VARIANTS="base=;ntld=x;ntst=x;ntall=x;gldsv=x"
NETS="vgg16 resnet50" ROUNDS=2 timeout 1200 bash tools/variant_ab.sh run r5_nt_ab > $O/nt_ab.txt 2>&1
tail -12 $O/nt_ab.txt
GEMM_R5=1 timeout 300 tools/_build/gemm_bench 10 > $O/gemm_r5.txt 2>&1
GEMM_TAIL=1 timeout 300 tools/_build/gemm_bench 10 > $O/gemm_tail.txt 2>&1
tail -30 $O/gemm_tail.txt
