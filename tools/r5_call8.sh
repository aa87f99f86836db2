#!/bin/bash
# round 5, GPU call 8: the LDS-tiled implicit GEMM bounded for 6 blocks per CU (80 registers, 44 - 132 bytes of scratch per lane) against 4 (86 - 103 registers, 5 resident)
O=gpurun_out/r5_call8
mkdir -p $O
export VARIANTS="base=;occ6=x"
NETS="resnet50 mobilenet_v1 vgg16" ROUNDS=4 timeout 1500 bash tools/variant_ab.sh run r5_ab8 > $O/ab8.txt 2>&1
tail -6 $O/ab8.txt | cut -c1-180
