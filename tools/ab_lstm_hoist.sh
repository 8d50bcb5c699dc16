#!/bin/bash
# bf16 UNet-LSTM cine (100 frames of 256x256): the hoisted-gx time steps (r05, default) against steps that re-multiply x
# (r06 experiment, UKBB_LSTM_BF16_UNHOIST=1): wall time per cine, three alternating rounds, then rocprofv3 kernel stats of each.   tools/ab_lstm_hoist.sh [outdir]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/gpurun_out/ab_lstm_hoist}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
for r in 1 2 3; do
  for m in 0 1; do
    if [ $m = 0 ]; then export UKBB_LSTM_BF16_UNHOIST=1; else unset UKBB_LSTM_BF16_UNHOIST; fi
    echo -n "hoisted=$m: "; python3 "$ROOT/tools/bench_unet_lstm.py" 10 bf16 2>&1 | grep "cine,"
  done
done
for m in 0 1; do
  if [ $m = 0 ]; then export UKBB_LSTM_BF16_UNHOIST=1; else unset UKBB_LSTM_BF16_UNHOIST; fi
  rm -rf "$OUT/trace$m"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace$m" -- python3 "$ROOT/tools/bench_unet_lstm.py" 5 bf16 > "$OUT/under_rocprof$m.txt" 2> "$OUT/trace$m.log"
  f=$(find "$OUT/trace$m" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_hoisted$m.csv"
  echo "hoisted=$m (rocprofv3 kernel stats, 7 cines):"
  python3 - "$OUT/kernel_stats_hoisted$m.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'lstm_' in r['Name']:
        print('   %-80s calls %5s avg %9.1f us  total %8.2f ms' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
  find "$OUT/trace$m" -name '*kernel_trace.csv' -size +20M -delete
done
