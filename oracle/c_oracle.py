"""ctypes driver of oracle/fcn_oracle.c (TEST INFRASTRUCTURE ONLY -- see
oracle/__init__.py).  Builds the library with gcc on first use if the prebuilt
file is missing."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libfcn_oracle.so')


class _Arch(C.Structure):
    _fields_ = [('kind', C.c_int32), ('n_class', C.c_int32), ('n_level', C.c_int32),
                ('n_filter', C.c_int32 * 8), ('n_block', C.c_int32 * 8),
                ('same_dim', C.c_int32), ('fc', C.c_int32)]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, 'fcn_oracle.c')):
        subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        for f in (_lib.oracle_fcn_forward, _lib.oracle_unet_forward):
            f.restype = C.c_int
            f.argtypes = [C.POINTER(_Arch), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int,
                          C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def num_threads():
    return lib().oracle_num_threads()


def forward(arch, flat_weights, image, want_logits=True, want_prob=False):
    """arch: ukbb_cardiac_amd.arch.ModelArch; flat_weights: pack_flat() array;
    image [N,H,W,1] f32.  Returns (logits|None, prob|None, pred int32)."""
    a = _Arch()
    a.kind, a.n_class, a.n_level = arch.kind, arch.n_class, arch.n_level
    for i in range(arch.n_level):
        a.n_filter[i], a.n_block[i] = arch.n_filter[i], arch.n_block[i]
    a.same_dim, a.fc = arch.same_dim, arch.fc
    x = np.ascontiguousarray(image, np.float32)
    n, h, w = x.shape[:3]
    wts = np.ascontiguousarray(flat_weights, np.float32)
    lg = np.empty((n, h, w, arch.n_class), np.float32) if want_logits else None
    pr = np.empty((n, h, w, arch.n_class), np.float32) if want_prob else None
    pd = np.empty((n, h, w), np.int32)
    fp = lambda v: v.ctypes.data_as(C.POINTER(C.c_float)) if v is not None else None
    fn = lib().oracle_fcn_forward if arch.kind == 0 else lib().oracle_unet_forward
    rc = fn(C.byref(a), fp(wts), fp(x), n, h, w, fp(lg), fp(pr), pd.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise RuntimeError('C oracle failed (bad shape or out of memory)')
    return lg, pr, pd
