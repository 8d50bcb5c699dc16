#!/bin/bash
# One counter pass of the bf16 U-Net forward, per-kernel means (a counter set the hardware cannot collect together makes rocprofv3
# abort and then hang in its finaliser -- hence the timeout; known-good sets: tools/profile_unet.sh):  tools/pmc_one.sh "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" [kernel-name filter]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_one
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc $1 --output-format csv -d "$OUT/pass" -- python3 "$ROOT/tools/bench_unet.py" 100 bf16 3 > "$OUT/pass.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$OUT" | grep -E "^kernel|${2:-.}"
