#!/bin/bash
# round 5, GPU call 2: where do non-temporal / write-through stores pay?  (M stores of the tile GEMMs alone, then activation stores per kernel family)
O=gpurun_out/r5_call2
mkdir -p $O
export VARIANTS="base=;gemmst=x;gemmsc1=x;gemmsc1nt=x;act1=x;act2=x;act4=x;act12=x"
NETS="vgg16 resnet50 mobilenet_v1" ROUNDS=3 timeout 1500 bash tools/variant_ab.sh run r5_nt_ab2 > $O/nt_ab2.txt 2>&1
tail -30 $O/nt_ab2.txt
