"""The C-ABI library loads without a GPU and exports every symbol the header
declares; compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'ukbb_fcn.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ukbb_fcn_[a-z_]+)\s*\(', src)))


def test_every_declared_symbol_is_exported():
    from ukbb_cardiac_amd import _lib
    names = header_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(_lib.lib, n), 'libukbb_fcn.so does not export %s' % n
    assert sorted(_lib.EXPORTS) == names
    hdr = open(os.path.join(ROOT, 'include', 'ukbb_fcn.h')).read()
    assert _lib.lib.ukbb_fcn_abi_version() == int(re.search(r'#define UKBB_FCN_ABI_VERSION (\d+)', hdr).group(1)) == _lib.ABI_VERSION


def test_weight_count_matches_python_arch():
    from ukbb_cardiac_amd import _lib
    from ukbb_cardiac_amd.arch import MODELS
    for arch in MODELS.values():
        s = _lib.arch_struct(arch)
        assert _lib.lib.ukbb_fcn_weight_count(C.byref(s)) == arch.n_weight_floats()
    bad = _lib.arch_struct(MODELS['FCN_sa'])
    bad.n_level = 99
    assert _lib.lib.ukbb_fcn_weight_count(C.byref(bad)) == 0


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from ukbb_cardiac_amd import _lib
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['FCN_sa']
    with pytest.raises(_lib.UkbbFcnError, match='no HIP device|CPU fallback'):
        Engine(arch, synthetic_params(arch))


def test_create_rejects_bad_arguments():
    from ukbb_cardiac_amd import _lib
    from ukbb_cardiac_amd.arch import MODELS
    s = _lib.arch_struct(MODELS['FCN_sa'])
    w = np.zeros(10, np.float32)
    h = _lib.lib.ukbb_fcn_create(C.byref(s), _lib.f32ptr(w), 10, 0)
    assert not h and 'expected' in _lib.last_error()
    s.fc = 48
    n = MODELS['FCN_sa'].n_weight_floats()
    assert not _lib.lib.ukbb_fcn_create(C.byref(s), _lib.f32ptr(w), n, 0)
    assert 'unsupported' in _lib.last_error() or 'expected' in _lib.last_error()


def test_weight_blob_roundtrip(tmp_path):
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import load_model
    from ukbb_cardiac_amd.weights import pack_flat, save_blob, synthetic_params
    arch = MODELS['FCN_la_2ch']
    params = synthetic_params(arch, 7)
    p = str(tmp_path / 'FCN_la_2ch')
    save_blob(p + '.ukbbw', arch, params)
    a2, p2 = load_model(p)                                   # reference-style --model_path prefix
    assert a2 == arch and np.array_equal(pack_flat(a2, p2), pack_flat(arch, params))
    with pytest.raises(FileNotFoundError):
        load_model(str(tmp_path / 'missing'))
