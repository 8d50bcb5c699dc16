#!/bin/bash
# Ablation of the fused ConvLSTM epilogue (diagnostic build only; results are garbage under any bit, only the cine time matters):
#   make -C ukbb_cardiac_amd/csrc -j8 BUILD=build_diag OUT=../libukbb_fcn_diag.so EXTRA=-DUKBB_DIAG ../libukbb_fcn_diag.so
#   bash tools/diag_lstm.sh
# UKBB_LSTM_DIAG bits: 1 = no cell arithmetic, 2 = no gx / c loads, 4 = no c / h stores, 8 = no epilogue at all (kernels_wino24.hip, fp32 Winograd forms).
# Stamps of the phases instead: build with EXTRA=-DUKBB_WINO_STAMPS and run with UKBB_STAMPS=1 (prints LSTMSTAMPS lines per launch).
export UKBB_FCN_LIB=$PWD/ukbb_cardiac_amd/libukbb_fcn_diag.so
[ -f "$UKBB_FCN_LIB" ] || { echo "build the diagnostic library first (see the head of this script)"; exit 1; }
for d in 0 1 2 4 3 7 8; do echo -n "diag $d: "; UKBB_LSTM_DIAG=$d timeout 120 python3 tools/bench_unet_lstm.py 5 2>&1 | grep "UNet-LSTM cine"; done
