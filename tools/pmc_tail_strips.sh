#!/bin/bash
# HBM traffic of the fused tail (and of the whole bf16 forward) under the two tile walks: counters in passes of their own, proven sets only, each under timeout.
#   tools/pmc_tail_strips.sh [outdir]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/gpurun_out/pmc_tail_strips}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  export UKBB_TAIL_STRIPS=$m
  timeout 240 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmcA$m" -- python3 "$ROOT/tools/bench_unet.py" 100 bf16 3 > "$OUT/pmcA$m.log" 2>&1
  timeout 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmcB$m" -- python3 "$ROOT/tools/bench_unet.py" 100 bf16 3 > "$OUT/pmcB$m.log" 2>&1
  python3 - "$OUT" $m <<'PY'
import csv, glob, sys, collections
out, m = sys.argv[1], sys.argv[2]
tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
allk = collections.defaultdict(float)
for d in ('pmcA', 'pmcB'):
    for f in glob.glob('%s/%s%s/**/*counter_collection.csv' % (out, d, m), recursive=True):
        for row in csv.DictReader(open(f)):
            if 'ukbb::' in row['Kernel_Name']:
                allk[row['Counter_Name']] += float(row['Counter_Value'] or 0)
            if 'unet_tail' in row['Kernel_Name']:
                tot[row['Counter_Name']] += float(row['Counter_Value'] or 0); cnt[row['Counter_Name']] += 1
per = {k: tot[k] / cnt[k] for k in tot}
if per:
    fetch, write = per.get('FETCH_SIZE', 0) * 1024, per.get('WRITE_SIZE', 0) * 1024
    nf = cnt['FETCH_SIZE']                      # tail launches = forwards of the pass
    print('UKBB_TAIL_STRIPS=%s: tail per launch: HBM bytes 2 x FETCH + WRITE = %.1f MB (FETCH_SIZE %.1f MB x 2, WRITE_SIZE %.1f MB), L2 hit rate %.2f; whole forward: %.1f MB (%d forwards)'
          % (m, (2 * fetch + write) / 1e6, fetch / 1e6, write / 1e6, per.get('TCC_HIT_sum', 0) / max(1.0, per.get('TCC_HIT_sum', 0) + per.get('TCC_MISS_sum', 0)),
             (2 * allk['FETCH_SIZE'] + allk['WRITE_SIZE']) * 1024 / 1e6 / max(1, nf), nf))
PY
  find "$OUT" -name '*counter_collection.csv' -size +20M -delete
done
