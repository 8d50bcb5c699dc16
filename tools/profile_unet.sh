#!/bin/bash
# BASELINE config 5 (aortic U-Net, N = 100 x 256 x 256): throughput fp32 / bf16 + Dice, per-kernel times, rocprofv3 kernel
# stats and hardware-counter passes of the bf16 path (counters in runs of their own, never with --kernel-trace).
# usage: tools/profile_unet.sh r03   -> gpurun_out/r03_unet/{unet.txt,kernels_bf16.txt,kernel_stats.csv,pmc/summary.csv,roofline.txt}
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_unet
mkdir -p "$OUT/pmc"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/tools/bench_unet.py" 100 > "$OUT/unet.txt" 2> "$OUT/unet.err"
python3 "$ROOT/tools/unet_kernels.py" 100 bf16 > "$OUT/kernels_bf16.txt" 2>> "$OUT/unet.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/bench_unet.py" 100 bf16 20 > "$OUT/under_rocprof.txt" 2> "$OUT/trace.log"
find "$OUT/trace" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
find "$OUT/trace" -name '*kernel_trace.csv' -size +20M -delete
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc/pass$i" -- python3 "$ROOT/tools/bench_unet.py" 100 bf16 3 > "$OUT/pmc/pass$i.log" 2>&1
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc" > "$OUT/pmc/summary.csv"
find "$OUT/pmc" -name '*counter_collection.csv' -size +20M -delete
python3 "$ROOT/tools/unet_roofline.py" "$OUT" > "$OUT/roofline.txt"
cat "$OUT/unet.txt" "$OUT/roofline.txt"
