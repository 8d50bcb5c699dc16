#!/bin/bash
# Packed vs scalar fp32 VALU next to the fp32 MFMA streams: builds of the library that differ only in how the
# vector-ALU work beside the MFMAs is spelled, alternated on the headline step.
#   A  libukbb_fcn.so                       the tree as built (hipcc SLP-packs the head's gather into v_pk_fma_f32; Winograd transforms use v_pk_add_f32)
#   B  tools/_bin/libukbb_fcn_nopk_head.so  head only: -fno-slp-vectorize -DUKBB_NO_PACKED_F32 (gather + logits as v_fma_f32)
#   C  tools/_bin/libukbb_fcn_nopk.so       every kernel file that way
# Build here (CPU container):  tools/ab_packed.sh build      Run on the GPU box:  tools/ab_packed.sh run > gpurun_out/ab_packed.txt
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/ukbb_cardiac_amd/csrc
if [ "${1:-run}" = build ]; then
  mkdir -p "$ROOT/tools/_bin"
  make -C "$CS" -j6 BUILD=build_nopk OUT="$ROOT/tools/_bin/libukbb_fcn_nopk.so" OUT_GZ=/dev/null EXTRA="-DUKBB_NO_PACKED_F32 -fno-slp-vectorize" "$ROOT/tools/_bin/libukbb_fcn_nopk.so" || exit 1
  # B: the default objects with only the head replaced
  OBJS=$(ls "$CS"/build/*.o | grep -v kernels_head.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/_bin/libukbb_fcn_nopk_head.so" $OBJS "$CS/build_nopk/kernels_head.o" || exit 1
  exit 0
fi
for r in 1 2 3; do
  for L in ukbb_cardiac_amd/libukbb_fcn.so tools/_bin/libukbb_fcn_nopk_head.so tools/_bin/libukbb_fcn_nopk.so; do
    echo -n "$L: "
    UKBB_FCN_LIB=$ROOT/$L python3 "$ROOT/bench.py" --no-cpu-baseline --no-other-configs --no-f32x3-probe --sustained-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline_detail']['per_kernel_us']
print('ms/step %.4f  head %.1f us  slices/s %.0f  | ' % (d['ms_per_step'], d['roofline']['avg_launch_us'], d['value']) + ' '.join('%s %.1f' % kv for kv in k.items()))"
  done
done
