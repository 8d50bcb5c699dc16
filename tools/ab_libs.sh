#!/bin/bash
# A/B of builds of the library on the headline step, alternating, 3 rounds: tools/ab_libs.sh libA.so libB.so ...  (paths relative to the repo root)
# prints ms per step, the head's launch time and every kernel's time from the bench line's untimed survey pass
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for r in 1 2 3; do
  for L in "$@"; do
    echo -n "$L: "
    UKBB_FCN_LIB=$ROOT/$L python3 "$ROOT/bench.py" --no-cpu-baseline --no-other-configs --no-f32x3-probe --sustained-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline_detail']['per_kernel_us']
print('ms/step %.4f  head %.1f us  slices/s %.0f  | ' % (d['ms_per_step'], d['roofline']['avg_launch_us'], d['value']) + ' '.join('%s %.1f' % kv for kv in k.items()))"
  done
done
