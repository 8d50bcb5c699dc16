#!/bin/bash
# How many wait states does a 16-byte buffer store with an SGPR soffset need before a VALU may overwrite its data registers?  (r06 notes section 10)
# Variant builds of kernels_ws.hip with -DUKBB_STORE_PAD=n linked into tools/_bin/libukbb_fcn_pad<n>.so (0 none, 1 s_nop 0, 5 s_nop 3,
# 2 s_waitcnt expcnt(0), 3 s_nop 7 x 2; the shipped library has both 2 and 3).  Each runs the two-stream check of the bf16 aortic U-Net
# (the victim) and the single-stream forward time.     tools/ab_store_pad.sh > gpurun_out/r06_ab_store_pad.txt
# Building a variant (here, before the gpurun call; tools/_bin/ is git-ignored but travels to the GPU box):
#   cd ukbb_cardiac_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DUKBB_STORE_PAD=$n -c kernels_ws.hip -o /tmp/pad$n.o
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_bin/libukbb_fcn_pad$n.so $(ls build/*.o | grep -v kernels_ws.o) /tmp/pad$n.o
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export UKBB_SPLIT_FROM=0 PREC=bf16
for v in pad0 pad1 pad5 pad2 pad3 shipped; do
  L=$ROOT/tools/_bin/libukbb_fcn_$v.so
  [ $v = shipped ] && L=$ROOT/ukbb_cardiac_amd/libukbb_fcn.so
  [ -f $L ] || { echo "$v: missing $L"; continue; }
  echo "== $v"
  UKBB_FCN_LIB=$L timeout 300 python3 $ROOT/tools/two_stream_check.py UNet_ao 10 304 272 ${ITERS:-240} 2>&1 | tail -2
  UKBB_FCN_LIB=$L timeout 300 python3 $ROOT/tools/bench_unet.py 100 bf16 2>&1 | tail -1
done
