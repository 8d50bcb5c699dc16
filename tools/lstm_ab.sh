#!/bin/bash
# per-kernel times of the UNet-LSTM cine under rocprofv3 for the two region shapes of the fused kernel: tools/lstm_ab.sh [fp32|bf16]
PREC=${1:-fp32}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for tc in 32 16; do
  export UKBB_LSTM_TILE_COLS=$tc
  python3 $ROOT/tools/bench_unet_lstm.py 10 $PREC 2>&1 | grep "cine,"
  rm -rf /tmp/lstm_ab; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lstm_ab -- python3 $ROOT/tools/bench_unet_lstm.py 5 $PREC > /dev/null 2>&1
  f=$(find /tmp/lstm_ab -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'wino24_pc_kernel' in r['Name'] and ('4, 1' in r['Name'] or '4, 2' in r['Name']) or 'lstm_' in r['Name']:
        print('   %-70s calls %5s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
