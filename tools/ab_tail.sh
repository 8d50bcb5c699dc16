#!/bin/bash
# Tail kernel A/B on the config-5 forward (N = 100 x 256x256, bf16): row-major tile walk vs column strips (kernels_tail.hip STRIP), three
# alternating rounds, wall time per forward and -- under rocprofv3 -- the tail kernel's own average.   tools/ab_tail.sh [outdir]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/gpurun_out/ab_tail}
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
for r in 1 2 3; do
  for m in 0 1; do
    echo -n "UKBB_TAIL_STRIPS=$m: "
    UKBB_TAIL_STRIPS=$m python3 "$ROOT/tools/bench_unet.py" 100 bf16 30 | head -1
  done
done
for seg in 4 8 16; do echo -n "UKBB_TAIL_STRIPS=1 UKBB_TAIL_SEG=$seg: "; UKBB_TAIL_STRIPS=1 UKBB_TAIL_SEG=$seg python3 "$ROOT/tools/bench_unet.py" 100 bf16 30 | head -1; done
for m in 0 1; do
  export UKBB_TAIL_STRIPS=$m
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace$m" -- python3 "$ROOT/tools/bench_unet.py" 100 bf16 20 > "$OUT/under_rocprof$m.txt" 2> "$OUT/trace$m.log"
  f=$(find "$OUT/trace$m" -name '*kernel_stats.csv' | head -1)
  cp "$f" "$OUT/kernel_stats_strips$m.csv"
  echo "UKBB_TAIL_STRIPS=$m (rocprofv3 kernel stats):"; grep -E "unet_tail|Name" "$OUT/kernel_stats_strips$m.csv" | cut -c1-200
  find "$OUT/trace$m" -name '*kernel_trace.csv' -size +20M -delete
done
unset UKBB_TAIL_STRIPS
# traffic of the tail under both walks: counters in passes of their own, ONLY the counter sets tools/profile_unet.sh has proven (another
# combination made rocprofv3 abort and then hang in its finaliser for 25 minutes: r06)
for m in 0 1; do
  export UKBB_TAIL_STRIPS=$m
  timeout 600 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmcA$m" -- python3 "$ROOT/tools/bench_unet.py" 100 bf16 3 > "$OUT/pmcA$m.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmcB$m" -- python3 "$ROOT/tools/bench_unet.py" 100 bf16 3 > "$OUT/pmcB$m.log" 2>&1
  python3 - "$OUT" $m <<'PY'
import csv, glob, sys, collections
out, m = sys.argv[1], sys.argv[2]
tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
for d in ('pmcA', 'pmcB'):
    for f in glob.glob('%s/%s%s/**/*counter_collection.csv' % (out, d, m), recursive=True):
        for row in csv.DictReader(open(f)):
            if 'unet_tail' in row['Kernel_Name']:
                tot[row['Counter_Name']] += float(row['Counter_Value'] or 0); cnt[row['Counter_Name']] += 1
per = {k: tot[k] / cnt[k] for k in tot}
if per:
    fetch, write = per.get('FETCH_SIZE', 0) * 1024, per.get('WRITE_SIZE', 0) * 1024
    print('UKBB_TAIL_STRIPS=%s tail per launch: HBM bytes 2 x FETCH + WRITE = %.1f MB (FETCH_SIZE %.1f MB x 2, WRITE_SIZE %.1f MB), L2 hit rate %.2f'
          % (m, (2 * fetch + write) / 1e6, fetch / 1e6, write / 1e6, per.get('TCC_HIT_sum', 0) / max(1.0, per.get('TCC_HIT_sum', 0) + per.get('TCC_MISS_sum', 0))))
PY
  find "$OUT" -name '*counter_collection.csv' -size +20M -delete
done
