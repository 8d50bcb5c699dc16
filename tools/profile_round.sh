#!/bin/bash
# One-shot profile collection on the GPU box (outputs under gpurun_out/$1, copy what is judged into profiles/):
#   1. rocprofv3 --kernel-trace --stats of the bench command, 2. PMC passes (own runs), 3. the plain bench.py line (with that traffic).
# usage: tools/profile_round.sh r01
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --inflight-probe > "$OUT/bench_inflight.json" 2> "$OUT/bench.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-f32x3-probe --no-other-configs > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
find "$OUT/trace" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
find "$OUT/trace" -name '*kernel_trace.csv' -size +20M -delete
cd "$ROOT" && bash tools/run_pmc.sh "gpurun_out/$TAG/pmc"
python3 "$ROOT/tools/pmc_traffic.py" "$OUT/pmc/summary.csv" > "$OUT/pmc_traffic.json"
# the plain bench line last, so that its roofline.traffic comes from the counter passes of this very session
cd /tmp && python3 "$ROOT/bench.py" --steps 20 --warmup 5 --pmc-traffic "$OUT/pmc_traffic.json" > "$OUT/bench.json" 2>> "$OUT/bench.err"
cd "$ROOT"
ls -la "$OUT"
