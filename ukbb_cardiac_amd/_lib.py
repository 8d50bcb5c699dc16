"""ctypes binding of include/ukbb_fcn.h (the only way Python reaches the kernels).

Fails loudly: if ``libukbb_fcn.so`` has not been built
(``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C ukbb_cardiac_amd/csrc``) importing this module raises -- there is no
CPU fallback on the product path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('UKBB_FCN_LIB') or os.path.join(_HERE, 'libukbb_fcn.so')   # override: A/B builds of the kernels
ABI_VERSION = 7
MAX_LEVEL = 8

# every symbol include/ukbb_fcn.h declares
EXPORTS = [
    'ukbb_fcn_abi_version', 'ukbb_fcn_last_error', 'ukbb_fcn_weight_count', 'ukbb_fcn_create',
    'ukbb_fcn_destroy', 'ukbb_fcn_reserve', 'ukbb_fcn_forward', 'ukbb_fcn_forward_host',
    'ukbb_fcn_num_kernels', 'ukbb_fcn_kernel_name', 'ukbb_fcn_kernel_macs', 'ukbb_fcn_set_timing',
    'ukbb_fcn_kernel_times', 'ukbb_fcn_get_activation', 'ukbb_fcn_kernel_config', 'ukbb_fcn_conv_config_name',
    'ukbb_fcn_set_timing_kernel', 'ukbb_fcn_set_precision', 'ukbb_fcn_kernel_mfma_macs',
    'ukbb_fcn_select_kth', 'ukbb_fcn_rescale_pack', 'ukbb_fcn_unpack_labels',
    'ukbb_fcn_roi_compact', 'ukbb_fcn_pairwise_sum', 'ukbb_fcn_zscore_pack',
    'ukbb_fcn_gzip_labels_bound', 'ukbb_fcn_gzip_labels', 'ukbb_fcn_gzip_labels_mode', 'ukbb_fcn_gunzip', 'ukbb_fcn_gzip_crc',
    'ukbb_fcn_forward_seq', 'ukbb_fcn_forward_cine', 'ukbb_fcn_clock_probe', 'ukbb_fcn_kernel_mfma_macs_issued',
    'ukbb_fcn_synth_volume',
]


class ArchStruct(C.Structure):
    _fields_ = [('kind', C.c_int32), ('n_class', C.c_int32), ('n_level', C.c_int32),
                ('n_filter', C.c_int32 * MAX_LEVEL), ('n_block', C.c_int32 * MAX_LEVEL),
                ('same_dim', C.c_int32), ('fc', C.c_int32)]


def arch_struct(arch) -> ArchStruct:
    s = ArchStruct()
    s.kind, s.n_class, s.n_level = arch.kind, arch.n_class, arch.n_level
    for i in range(arch.n_level):
        s.n_filter[i] = arch.n_filter[i]
        s.n_block[i] = arch.n_block[i]
    s.same_dim, s.fc = arch.same_dim, arch.fc
    return s


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'ukbb_cardiac_amd: %s is missing. Build the gfx950 HIP library first '
            '(python -c "import __graft_entry__ as g; g.build()"). There is no CPU fallback.' % LIB_PATH)
    # torch bundles its own libamdhip64; when both it and this library live in one
    # process the HIP runtime must be loaded once, by whoever comes first.  Loading
    # ours first and torch later left the second copy without a visible device on
    # the GPU boxes, so let torch (when installed) bring the runtime in first.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, f32p, i32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
    lib.ukbb_fcn_abi_version.restype = C.c_int
    lib.ukbb_fcn_last_error.restype = C.c_char_p
    lib.ukbb_fcn_weight_count.restype = C.c_size_t
    lib.ukbb_fcn_weight_count.argtypes = [C.POINTER(ArchStruct)]
    lib.ukbb_fcn_create.restype = vp
    lib.ukbb_fcn_create.argtypes = [C.POINTER(ArchStruct), f32p, C.c_size_t, C.c_int]
    lib.ukbb_fcn_destroy.restype = None
    lib.ukbb_fcn_destroy.argtypes = [vp]
    lib.ukbb_fcn_reserve.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.ukbb_fcn_forward.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.ukbb_fcn_forward_host.argtypes = [vp, f32p, C.c_int, C.c_int, C.c_int, f32p, f32p, i32p]
    lib.ukbb_fcn_num_kernels.argtypes = [vp]
    lib.ukbb_fcn_kernel_name.restype = C.c_char_p
    lib.ukbb_fcn_kernel_name.argtypes = [vp, C.c_int]
    lib.ukbb_fcn_kernel_macs.restype = C.c_double
    lib.ukbb_fcn_kernel_macs.argtypes = [vp, C.c_int]
    lib.ukbb_fcn_kernel_mfma_macs.restype = C.c_double
    lib.ukbb_fcn_kernel_mfma_macs.argtypes = [vp, C.c_int]
    lib.ukbb_fcn_kernel_mfma_macs_issued.restype = C.c_double
    lib.ukbb_fcn_kernel_mfma_macs_issued.argtypes = [vp, C.c_int]
    lib.ukbb_fcn_kernel_config.argtypes = [vp, C.c_int]
    lib.ukbb_fcn_conv_config_name.restype = C.c_char_p
    lib.ukbb_fcn_conv_config_name.argtypes = [C.c_int]
    lib.ukbb_fcn_set_timing.argtypes = [vp, C.c_int]
    lib.ukbb_fcn_set_timing_kernel.argtypes = [vp, C.c_int]
    lib.ukbb_fcn_set_precision.argtypes = [vp, C.c_int]
    lib.ukbb_fcn_forward_seq.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.ukbb_fcn_forward_cine.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, vp]
    lib.ukbb_fcn_select_kth.argtypes = [vp, C.c_size_t, C.POINTER(C.c_uint64), C.c_int, f32p, vp]
    lib.ukbb_fcn_rescale_pack.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                          C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.ukbb_fcn_unpack_labels.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           vp, vp, vp]
    lib.ukbb_fcn_roi_compact.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float,
                                         vp, C.POINTER(C.c_uint64), vp]
    lib.ukbb_fcn_pairwise_sum.argtypes = [vp, C.c_uint64, C.c_int, C.c_float, C.POINTER(C.c_float), vp]
    lib.ukbb_fcn_zscore_pack.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                         C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.ukbb_fcn_gzip_labels_bound.restype = C.c_uint64
    lib.ukbb_fcn_gzip_labels_bound.argtypes = [C.c_uint64, C.c_int, C.c_uint64]
    lib.ukbb_fcn_gzip_labels.restype = C.c_int64
    lib.ukbb_fcn_gzip_labels.argtypes = [vp, C.c_uint64, C.c_int, vp, C.c_uint64, vp, C.c_uint64]
    lib.ukbb_fcn_gzip_labels_mode.restype = C.c_int64
    lib.ukbb_fcn_gzip_labels_mode.argtypes = [vp, C.c_uint64, C.c_int, vp, C.c_uint64, vp, C.c_uint64, C.c_int]
    lib.ukbb_fcn_gunzip.restype = C.c_int64
    lib.ukbb_fcn_gunzip.argtypes = [vp, C.c_uint64, vp, C.c_uint64, C.c_int]
    lib.ukbb_fcn_gzip_crc.restype = C.c_uint32
    lib.ukbb_fcn_gzip_crc.argtypes = [C.c_uint32, vp, C.c_uint64]
    lib.ukbb_fcn_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int, C.c_int]
    lib.ukbb_fcn_get_activation.restype = C.c_int64
    lib.ukbb_fcn_get_activation.argtypes = [vp, C.c_char_p, f32p, C.c_int64]
    lib.ukbb_fcn_synth_volume.argtypes = [C.c_uint64, C.c_size_t, vp, vp]
    lib.ukbb_fcn_clock_probe.argtypes = [C.c_int, vp, C.c_int, C.POINTER(C.c_double)]
    if lib.ukbb_fcn_abi_version() != ABI_VERSION:
        raise ImportError('libukbb_fcn.so ABI %d != binding ABI %d: rebuild' % (lib.ukbb_fcn_abi_version(), ABI_VERSION))
    return lib


lib = _load()


class UkbbFcnError(RuntimeError):
    pass


def last_error() -> str:
    return lib.ukbb_fcn_last_error().decode()


def check(rc: int, what: str):
    if rc < 0:
        raise UkbbFcnError('%s failed (%d): %s' % (what, rc, last_error()))
    return rc


def clock_probe_mhz(device: int = 0, stream: int = 0, spin_us: int = 200) -> float:
    """Shader clock the chip holds right now (ukbb_fcn_clock_probe): one wave spinning on `stream` for spin_us microseconds."""
    mhz = C.c_double(0.0)
    check(lib.ukbb_fcn_clock_probe(int(device), C.c_void_p(stream), int(spin_us), C.byref(mhz)), 'ukbb_fcn_clock_probe')
    return float(mhz.value)


def f32ptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def i32ptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))
