#!/bin/bash
# r04 helper (GPU box): parity of the weight-stationary bf16 tilings against the kernels they replace, then a timing sweep.
# usage: tools/ws_round.sh TAG "check specs small" "check specs 256" "sweep specs"
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python tools/check_ws.py 2 64 96 $2 > $OUT/check_small.txt 2>&1; tail -4 $OUT/check_small.txt
python tools/check_ws.py 3 256 256 $3 > $OUT/check_256.txt 2>&1; tail -4 $OUT/check_256.txt
MODEL=UNet_ao SHAPE=100,256,256 PREC=bf16 python tools/sweep_convs.py $4 > $OUT/sweep.txt 2>&1; cat $OUT/sweep.txt
