"""Per-device launch bookkeeping (ukbb_cardiac_amd/csrc/device_state.h): the header is free of HIP, so its
contract -- once per DEVICE, retry after failure, no caching outside the table, thread-safe -- is checked on the CPU; and no
launch helper may guard a per-device HIP call with a process-wide static any more."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'ukbb_cardiac_amd', 'csrc')


def test_once_per_device_contract(tmp_path):
    exe = str(tmp_path / 'device_state_test')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-pthread', '-o', exe, os.path.join(ROOT, 'tests', 'cpp', 'device_state_test.cpp')])
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert 'device_state ok' in out


def test_no_process_wide_static_guards_a_per_device_call():
    for name in os.listdir(CSRC):
        if not name.endswith(('.hip', '.cpp', '.h')):
            continue
        src = open(os.path.join(CSRC, name)).read()
        # hipFuncSetAttribute is reached only through allow_dynamic_lds (kernels.h)
        for m in re.finditer(r'hipFuncSetAttribute\(reinterpret_cast|= hipFuncSetAttribute', src):
            assert name == 'kernels.h', '%s calls hipFuncSetAttribute directly' % name
        assert not re.search(r'static\s+bool\s+(attr_done|done)\b', src), name
        assert not re.search(r'static\s+const\s+int\s+n_cu\b', src), name
        # device properties are read per device (create() validates the device it was given; kernels.h caches per ordinal)
        if name not in ('kernels.h', 'engine.cpp'):
            assert 'hipGetDeviceProperties' not in src, name
