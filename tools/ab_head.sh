#!/bin/bash
# A/B of two builds of the library on the headline step: tools/ab_head.sh libA.so libB.so  (per-kernel head time + ms per step, 3 rounds alternating)
for r in 1 2 3; do
  for L in "$@"; do
    echo -n "$L: "
    UKBB_FCN_LIB=$PWD/ukbb_cardiac_amd/$L python3 bench.py --no-cpu-baseline --no-other-configs --no-f32x3-probe --sustained-seconds 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['value'])"
  done
done
