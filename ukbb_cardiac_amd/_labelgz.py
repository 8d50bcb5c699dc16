"""ctypes binding of the label-volume gzip writer (``ukbb_fcn_gzip_labels*`` of include/ukbb_fcn.h) and of the whole-file
gzip reader (``ukbb_fcn_gunzip``) from ``libukbb_labelgz.so``: ``csrc/label_gzip.cpp`` + ``csrc/gz_inflate.cpp`` built on
their own, without HIP, so that host-only file I/O (``nifti.save`` / ``nifti.load``) neither loads the GPU runtime nor imports
torch.  The same symbols are also exported by ``libukbb_fcn.so``."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('UKBB_LABELGZ_LIB') or os.path.join(_HERE, 'libukbb_labelgz.so')
FIXED, DYNAMIC = 0, 1

lib = C.CDLL(LIB_PATH)                                       # OSError when it has not been built
lib.ukbb_fcn_gzip_labels_bound.restype = C.c_uint64
lib.ukbb_fcn_gzip_labels_bound.argtypes = [C.c_uint64, C.c_int, C.c_uint64]
lib.ukbb_fcn_gzip_labels_mode.restype = C.c_int64
lib.ukbb_fcn_gzip_labels_mode.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int]
lib.ukbb_fcn_gunzip.restype = C.c_int64
lib.ukbb_fcn_gunzip.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int]
lib.ukbb_fcn_gzip_crc.restype = C.c_uint32
lib.ukbb_fcn_gzip_crc.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64]
