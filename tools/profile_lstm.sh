#!/bin/bash
# The reference's DEFAULT aortic model (UNet-LSTM, common/network_ao.py:255-319, deploy_network_ao.py:129-183): one 100-frame cine of
# 256 x 256 through ukbb_fcn_forward_cine -- wall time, rocprofv3 kernel stats and hardware-counter passes (counters in runs of their
# own, never with --kernel-trace; the program directly after `--`).
# usage: tools/profile_lstm.sh r05 [fp32|bf16]  -> gpurun_out/r05_unet_lstm[_bf16]/{lstm.txt,kernel_stats.csv,pmc/summary.csv,bytes.txt}
set -u
TAG=${1:-r05}
PREC=${2:-fp32}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SUF=""; [ "$PREC" != fp32 ] && SUF="_$PREC"
OUT=$ROOT/gpurun_out/${TAG}_unet_lstm$SUF
mkdir -p "$OUT/pmc"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/tools/bench_unet_lstm.py" 10 $PREC > "$OUT/lstm.txt" 2> "$OUT/lstm.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/bench_unet_lstm.py" 10 $PREC > "$OUT/under_rocprof.txt" 2> "$OUT/trace.log"
find "$OUT/trace" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
find "$OUT/trace" -name '*kernel_trace.csv' -size +20M -delete
i=0
for set in \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc/pass$i" -- python3 "$ROOT/tools/bench_unet_lstm.py" 1 $PREC > "$OUT/pmc/pass$i.log" 2>&1
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc" > "$OUT/pmc/summary.csv"
find "$OUT/pmc" -name '*counter_collection.csv' -size +20M -delete
python3 "$ROOT/tools/lstm_bytes.py" "$OUT" > "$OUT/bytes.txt" 2>> "$OUT/lstm.err"
cat "$OUT/lstm.txt" "$OUT/bytes.txt"
