for v in base alt base alt; do
  if [ $v = base ]; then unset UKBB_FCN_LIB; else export UKBB_FCN_LIB=$PWD/ukbb_cardiac_amd/libukbb_fcn_$v.so; fi
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/b_$v.json 2>gpurun_out/b_$v.err
  python -c "
import json;d=json.load(open('gpurun_out/b_$v.json'));k=d['roofline_detail']['per_kernel_us'];print('$v',d['value'],d['ms_per_step'],k['conv0_0+conv0_1'],k['conv1_0'],k['head'])"
done
UKBB_FCN_LIB=$PWD/ukbb_cardiac_amd/libukbb_fcn_alt.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
