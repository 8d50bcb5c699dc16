export UKBB_FCN_LIB=$PWD/ukbb_cardiac_amd/libukbb_fcn_diag.so
for d in 0 1 2 4 3 7 8; do echo -n "diag $d: "; UKBB_LSTM_DIAG=$d timeout 120 python tools/bench_unet_lstm.py 5 2>&1 | grep "UNet-LSTM cine"; done
for d in 0 2 8; do echo -n "bf16 diag $d: "; UKBB_LSTM_DIAG=$d timeout 120 python tools/bench_unet_lstm.py 5 bf16 2>&1 | grep "UNet-LSTM cine"; done
