#!/bin/bash
# Hardware-counter passes for the bench workload (GPU box).  Counters are collected in their
# own runs, never together with --kernel-trace/--stats (MI355X guide / gpurun rule).
# usage: tools/run_pmc.sh OUTDIR
set -u
OUT=${1:-gpurun_out/pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-f32x3-probe --no-other-configs --no-kernel-events --steps 3 --warmup 1"
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d "$ROOT/$OUT/pass$i" -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/pass$i.log" 2>&1
done
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT" > "$ROOT/$OUT/summary.csv"
find "$ROOT/$OUT" -name '*counter_collection.csv' -size +20M -delete
