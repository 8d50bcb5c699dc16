"""Minimal NIfTI-1 (.nii / .nii.gz) reader and writer.

The reference uses nibabel (``nib.load`` / ``nim.get_data()`` / ``nim.affine`` /
``nim.header['pixdim']`` / ``nib.Nifti1Image(data, affine)`` / ``nib.save``;
``common/deploy_network.py:80-83,136-151``), which is not installed here and may
not be on the GPU box.  Only what the deployment and evaluation scripts touch is
implemented: single-file NIfTI-1, the common scalar datatypes, scl_slope/inter,
sform/qform affines, pixdim.  Data is returned in (X, Y, Z, T) order, i.e. the
file's Fortran order, like nibabel.
"""
import gzip
import os
import threading
import zlib
import struct

import numpy as np

_DTYPES = {2: 'u1', 4: 'i2', 8: 'i4', 16: 'f4', 64: 'f8', 256: 'i1', 512: 'u2', 768: 'u4', 1024: 'i8', 1280: 'u8'}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}


class NiftiImage:
    """data [X,Y,(Z,(T))] ndarray, affine 4x4 float64, pixdim float32[8]
    (pixdim[1:4] voxel size, pixdim[4] frame time: data/biobank_utils.py:59-63)."""

    def __init__(self, data, affine, pixdim=None, header=None):
        self.data = data
        self.affine = np.asarray(affine, dtype=np.float64)
        self.header = dict(header or {})
        if pixdim is None:
            pixdim = np.ones(8, np.float32)
            vox = np.sqrt((self.affine[:3, :3] ** 2).sum(axis=0))
            pixdim[1:4] = vox
        self.header['pixdim'] = np.asarray(pixdim, dtype=np.float32).copy()

    @property
    def shape(self):
        return self.data.shape

    def get_data(self):
        return self.data


GZIP_LEVEL = 1        # nibabel's Opener default (default_compresslevel = 1): the files the reference writes use it


class _GzWriter:
    """gzip stream without file name and timestamp in its header: the same volume always gives the same bytes, so
    outputs of different workers / reruns can be compared with cmp."""

    def __init__(self, path):
        self._raw = _AtomicFile(path)
        self._gz = gzip.GzipFile(filename='', mode='wb', compresslevel=GZIP_LEVEL, fileobj=self._raw.f, mtime=0)

    def write(self, b):
        return self._gz.write(b)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        try:
            self._gz.close()
        except BaseException as e:                          # a failing flush must not rename a truncated file into place
            self._raw.__exit__(type(e), e, e.__traceback__)
            raise
        self._raw.__exit__(*exc)


def _host_tag():
    """Short tag of this host *instance* (hostname + boot id + pid namespace): a pid only means something inside the namespace
    that issued it.  Two containers of one node share the boot id -- and, with host networking / a shared UTS namespace, the
    hostname -- but not the pid namespace, so its identity (the inode in ``/proc/self/ns/pid``) is part of the tag."""
    global _HOST_TAG
    if _HOST_TAG is None:
        import hashlib
        import socket
        boot, pidns = '', ''
        try:
            with open('/proc/sys/kernel/random/boot_id') as f:
                boot = f.read().strip()
        except OSError:
            pass
        try:
            pidns = os.readlink('/proc/self/ns/pid')        # 'pid:[4026531836]'
        except OSError:
            pass
        _HOST_TAG = 'h' + hashlib.sha1((socket.gethostname() + '|' + boot + '|' + pidns).encode()).hexdigest()[:10]
    return _HOST_TAG


_HOST_TAG = None
_FOREIGN_TMP_MAX_AGE_S = 6 * 3600.0                         # a writer on another host gets this long before its file counts as abandoned


def _tmp_name(path):
    return '%s.tmp.%d.%d.%s' % (path, os.getpid(), threading.get_ident(), _host_tag())


def _sweep_stale_tmp(path):
    """Remove ``<path>.tmp.<pid>.<thread>.<host>`` files whose writing process no longer exists: what a worker killed mid-write (the
    case _AtomicFile exists for) leaves behind; without this, reruns accumulate multi-megabyte partial files in the subject
    directories.  Files of live processes (another worker writing the same target right now) are left alone.  The pid probe is only
    valid for files written on THIS host (same host tag, or the untagged names of earlier versions); a file tagged by another host
    or container sharing the data directory (multi-node shards) is removed by age alone, never by a pid that means nothing here."""
    import time
    d, base = os.path.split(path)
    prefix = base + '.tmp.'
    try:
        names = os.listdir(d or '.')
    except OSError:
        return
    for nm in names:
        if not nm.startswith(prefix):
            continue
        parts = nm[len(prefix):].split('.')
        pid = parts[0]
        if not pid.isdigit():
            continue
        tag = parts[2] if len(parts) > 2 else None
        full = os.path.join(d, nm)
        if tag is not None and tag != _host_tag():
            try:
                if time.time() - os.path.getmtime(full) > _FOREIGN_TMP_MAX_AGE_S:
                    os.remove(full)
            except OSError:
                pass
            continue
        if int(pid) == os.getpid():
            continue
        try:
            os.kill(int(pid), 0)                               # signal 0: existence check only
            continue                                            # alive: not ours to touch
        except ProcessLookupError:
            pass
        except OSError:                                         # e.g. EPERM: exists under another user
            continue
        try:
            os.remove(full)
        except OSError:
            pass


class _AtomicFile:
    """Binary output file that only appears under its final name once it is complete: written as
    ``<path>.tmp.<pid>.<thread>.<host>`` and ``os.replace``d into place on a clean exit, removed otherwise.  A worker
    killed mid-write therefore never leaves a truncated ``seg_*.nii.gz`` behind -- that file is the 'already
    segmented, skip' marker of the deploy loops (common/deploy_network.py:66-67)."""

    def __init__(self, path):
        self.path, self.tmp = str(path), _tmp_name(str(path))
        _sweep_stale_tmp(self.path)
        self.f = open(self.tmp, 'wb')

    def write(self, b):
        return self.f.write(b)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.f.close()
        if exc and exc[0] is not None:
            try:
                os.remove(self.tmp)
            except OSError:
                pass
        else:
            os.replace(self.tmp, self.path)


class _GzReader:
    """Inflates a .gz file (all its members) in pieces of 1 MB of compressed input.  gzip.GzipFile feeds zlib 8 KB at a time
    from Python: thousands of interpreter round trips per cine, all under the GIL -- with several reader threads
    (deploy_network.py --io_threads) that, not inflate itself, was what bounded a cohort run.  Here the interpreter is entered a
    dozen times per file and zlib (which releases the GIL and checks each member's CRC-32 and length) does the rest."""
    PIECE = 1 << 20

    def __init__(self, path):
        with open(path, 'rb') as f:
            self._mv = memoryview(f.read())
        self._pos = 0
        self._pending = b''                                 # input handed to zlib but lying behind a member's trailer
        self._d = zlib.decompressobj(31)                     # 31: gzip container
        self._out = memoryview(b'')

    def _piece(self):
        if self._pending:
            piece, self._pending = self._pending, b''
            return piece
        piece = self._mv[self._pos:self._pos + self.PIECE]
        self._pos += len(piece)
        return piece

    def _more(self):
        """Next run of inflated bytes, b'' at the end of the file."""
        while True:
            if self._d.eof:                                  # a member ended: zero padding, another member, or the end
                rest = self._d.unused_data.lstrip(b'\x00')
                while not rest and self._pos < len(self._mv):
                    rest = bytes(self._piece()).lstrip(b'\x00')
                if not rest:
                    return b''
                if rest[:2] != b'\x1f\x8b' and len(rest) >= 2:
                    raise ValueError('trailing bytes after the gzip stream are not a gzip member')
                self._pending = rest
                self._d = zlib.decompressobj(31)
            piece = self._piece()
            if not len(piece):
                raise EOFError('compressed file ended before the end-of-stream marker was reached')
            out = self._d.decompress(piece)
            if out:
                return out

    def readinto(self, view):
        view = memoryview(view).cast('B')
        got = 0
        while got < len(view):
            if not len(self._out):
                self._out = memoryview(self._more())
                if not len(self._out):
                    break
            k = min(len(view) - got, len(self._out))
            view[got:got + k] = self._out[:k]
            self._out = self._out[k:]
            got += k
        return got

    def read(self, n=-1):
        if n < 0:
            parts = [bytes(self._out)]
            self._out = memoryview(b'')
            while True:
                more = self._more()
                if not more:
                    return b''.join(parts)
                parts.append(more)
        buf = bytearray(n)
        got = self.readinto(buf)
        return buf if got == n else bytes(buf[:got])

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._mv = self._out = None


def _open(path, mode):
    if not str(path).endswith('.gz'):
        return _AtomicFile(path) if 'w' in mode else open(path, mode)
    return _GzWriter(path) if 'w' in mode else _GzReader(path)


def _quat_affine(b, c, d, qx, qy, qz, pixdim):
    a2 = 1.0 - (b * b + c * c + d * d)
    a = np.sqrt(a2) if a2 > 0 else 0.0
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    qfac = -1.0 if pixdim[0] < 0 else 1.0
    S = np.diag([pixdim[1], pixdim[2], pixdim[3] * qfac])
    A = np.eye(4)
    A[:3, :3] = R @ S
    A[:3, 3] = [qx, qy, qz]
    return A


def _parse_header(raw, path):
    if len(raw) < 348:
        raise ValueError('%s: too short for a NIfTI-1 header' % path)
    end = '<'
    if struct.unpack('<i', raw[:4])[0] != 348:
        end = '>'
        if struct.unpack('>i', raw[:4])[0] != 348:
            raise ValueError('%s: not a NIfTI-1 file (sizeof_hdr != 348)' % path)
    if raw[344:347] not in (b'n+1', b'ni1'):
        raise ValueError('%s: bad NIfTI magic %r' % (path, raw[344:348]))
    if raw[344:347] == b'ni1':
        raise ValueError('%s: two-file NIfTI (.hdr/.img) is not supported' % path)
    dim = struct.unpack(end + '8h', raw[40:56])
    datatype, bitpix = struct.unpack(end + '2h', raw[70:74])
    pixdim = np.array(struct.unpack(end + '8f', raw[76:108]), dtype=np.float32)
    vox_offset, slope, inter = struct.unpack(end + '3f', raw[108:120])
    qform_code, sform_code = struct.unpack(end + '2h', raw[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(end + '6f', raw[256:280])
    srow = np.array(struct.unpack(end + '12f', raw[280:328]), dtype=np.float64).reshape(3, 4)
    if datatype not in _DTYPES:
        raise ValueError('%s: unsupported NIfTI datatype %d' % (path, datatype))
    ndim = dim[0]
    if not 1 <= ndim <= 7:
        raise ValueError('%s: bad dim[0] = %d' % (path, ndim))
    shape = tuple(int(d) for d in dim[1:1 + ndim])
    dt = np.dtype(end + _DTYPES[datatype])
    off = int(vox_offset) if vox_offset >= 352 else 352
    if sform_code > 0:                                         # nibabel's get_best_affine order
        affine = np.vstack([srow, [0, 0, 0, 1]])
    elif qform_code > 0:
        affine = _quat_affine(qb, qc, qd, qx, qy, qz, pixdim)
    else:
        affine = np.diag([pixdim[1], pixdim[2], pixdim[3], 1.0]).astype(np.float64)
        affine[:3, 3] = -0.5 * (np.array(shape[:3] + (1,) * (3 - min(3, len(shape))))[:3] - 1) * pixdim[1:4]
    hdr = {'dim': dim, 'datatype': datatype, 'bitpix': bitpix, 'qform_code': qform_code, 'sform_code': sform_code,
           'xyzt_units': raw[123], 'descrip': raw[148:228].rstrip(b'\x00')}
    scaled = slope != 0 and not (slope == 1 and inter == 0) and np.isfinite(slope)
    return shape, dt, off, (slope, inter) if scaled else None, affine, pixdim, hdr


def load_header(path):
    """Header fields only (``pixdim``, ``dim``, ..., plus ``shape``, ``dtype``, ``affine``): the first 352 bytes are inflated,
    not the volume -- what eval scripts take from ``nib.load(image_name).header`` (eval_ventricular_volume.py:40-47)."""
    with _open(path, 'rb') as f:
        shape, dt, off, scale, affine, pixdim, hdr = _parse_header(f.read(352), path)
    hdr = dict(hdr)
    hdr.update(pixdim=pixdim, shape=shape, dtype=dt, affine=affine)
    return hdr


NATIVE_GUNZIP = os.environ.get('UKBB_GUNZIP', 'native') != 'zlib'     # A/B knob (tools/shard_rehearsal.py); tests switch it off to compare with the zlib reader


def _checked_alloc(alloc, shape, dt):
    """alloc(shape, dtype) -> array, or (array, headroom): `headroom` bytes of writable memory lie in front of the array's first
    element (ukbb_cardiac_amd/subject_pipeline.py pads its pinned buffers), which lets the whole-file decoder put the header there
    and the voxels straight into the array."""
    got = alloc(shape, dt)
    data, headroom = got if isinstance(got, tuple) else (got, 0)
    if data.shape != shape or data.dtype != dt or not data.flags.f_contiguous or not data.flags.writeable:
        raise ValueError('alloc() must return a writable Fortran-ordered array of the requested shape and dtype')
    return data, int(headroom)


def _finish(raw_view, shape, dt, scale):
    """voxel bytes (file order) -> the array load() returns when no alloc() buffer is filled in place"""
    n = int(np.prod(shape))
    data = np.frombuffer(raw_view, dtype=dt, count=n).reshape(shape, order='F')
    if not dt.isnative or not data.flags.writeable:
        data = data.astype(dt.newbyteorder('='), copy=True)         # writable, native order
    if scale is not None:
        data = data * np.float64(scale[0]) + np.float64(scale[1])   # nibabel's get_data() applies the scaling
    return data


def _load_gz_whole(path, get_dest):
    """.nii.gz through ukbb_fcn_gunzip (csrc/gz_inflate.cpp): the file is read once and inflated in ONE call -- no window, no
    piece-wise interpreter round trips, 2x zlib 1.2.11's rate on MR image data -- straight into the destination when that has
    room for the 352 header bytes in front of it.  Returns None whenever this path does not apply or the decoder refuses the
    stream (it is strict); the caller then takes the zlib reader, which accepts or raises exactly as it always did."""
    if not NATIVE_GUNZIP:
        return None
    try:
        from . import _labelgz
        gunzip = _labelgz.lib.ukbb_fcn_gunzip
    except Exception:                                           # library missing or stale: the zlib reader
        return None
    with open(path, 'rb') as f:
        blob = f.read()
    try:
        head = zlib.decompressobj(31).decompress(blob[:1 << 16], 352)
    except zlib.error:
        return None
    if len(head) < 352:                                         # a tiny first member, an empty file, ...
        return None
    shape, dt, off, scale, affine, pixdim, hdr = _parse_header(head, path)
    nbytes = int(np.prod(shape)) * dt.itemsize
    total = off + nbytes
    src = np.frombuffer(blob, np.uint8)
    data, headroom = get_dest(shape, dt, scale)
    if data is not None and headroom >= off:
        got = gunzip(src.ctypes.data, len(blob), data.ctypes.data - off, total, 1)
        if got != total:                                        # shorter (truncated), longer (-4) or refused: zlib decides, into the same array
            return None
        return NiftiImage(data, affine, pixdim, hdr)
    buf = np.empty(total, np.uint8)
    got = gunzip(src.ctypes.data, len(blob), buf.ctypes.data, total, 1)
    if got != total:
        return None
    if data is not None:
        memoryview(data.reshape(-1, order='F')).cast('B')[:] = buf[off:]
        return NiftiImage(data, affine, pixdim, hdr)
    return NiftiImage(_finish(buf[off:], shape, dt, scale), affine, pixdim, hdr)


def load(path, alloc=None) -> NiftiImage:
    """``alloc(shape, dtype) -> writable Fortran-ordered ndarray`` (optional; or ``(ndarray, headroom_bytes)``) supplies the memory
    the voxels are decompressed into -- e.g. a view of pinned host memory, so the volume goes file -> staging buffer with no copy
    in between (ukbb_cardiac_amd/subject_pipeline.py).  It is used when the file's voxel type is native-endian and
    unscaled (the float32 cines of the reference, data/biobank_utils.py:314); otherwise the data is a fresh array."""
    dest = {}

    def get_dest(shape, dt, scale):                             # alloc() is called at most once per load, whichever reader ends up filling it
        if 'd' not in dest:
            direct = alloc is not None and scale is None and dt.isnative
            dest['d'] = _checked_alloc(alloc, shape, dt) if direct else (None, 0)
        return dest['d']

    if str(path).endswith('.gz'):
        img = _load_gz_whole(path, get_dest)
        if img is not None:
            return img
    with _open(path, 'rb') as f:
        head = f.read(352)
        shape, dt, off, scale, affine, pixdim, hdr = _parse_header(head, path)
        if off > 352:
            f.read(off - 352)                                   # header extensions
        n = int(np.prod(shape))
        data, _ = get_dest(shape, dt, scale)
        if data is not None:
            view = memoryview(data.reshape(-1, order='F')).cast('B')     # F-order memory, as on disk
            got = 0
            while got < len(view):
                k = f.readinto(view[got:])
                if not k:
                    break
                got += k
            if got < len(view):
                raise ValueError('%s: truncated image data' % path)
        else:
            raw = f.read(n * dt.itemsize)
            if len(raw) < n * dt.itemsize:
                raise ValueError('%s: truncated image data' % path)
            data = np.frombuffer(raw, dtype=dt, count=n).reshape(shape, order='F')
            data = data.astype(dt.newbyteorder('='), copy=True)       # writable, native order
            if scale is not None:
                data = data * np.float64(scale[0]) + np.float64(scale[1])   # nibabel's get_data() applies the scaling
    return NiftiImage(data, affine, pixdim, hdr)


def save(img_or_data, path, affine=None, pixdim=None, as_dtype=None):
    """save(NiftiImage, path) or save(ndarray, path, affine[, pixdim]).  Writes the
    array's own dtype (the reference relies on that: float64 label volumes in
    sequence mode, int32 in ED/ES mode), sform = affine
    (code 2, as nibabel's Nifti1Image(data, affine) does), qform code 0.

    ``as_dtype``: store the values converted to this dtype, slab by slab along the
    last axis while compressing -- the file is byte-identical to ``save(data.astype(as_dtype), ...)``
    without ever holding the converted volume (a 20 MB uint8 label volume becomes the
    160 MB float64 file of deploy_network.py:92,136 this way)."""
    if isinstance(img_or_data, NiftiImage):
        data, affine, pixdim = img_or_data.data, img_or_data.affine, img_or_data.header['pixdim']
    else:
        data = img_or_data
    data = np.asarray(data)
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    out_dtype = np.dtype(as_dtype) if as_dtype is not None else data.dtype
    key = out_dtype.newbyteorder('<').str[1:]
    if key not in _CODES:
        raise ValueError('cannot store dtype %s in NIfTI-1' % data.dtype)
    affine = np.asarray(affine, dtype=np.float64)
    if affine.shape != (4, 4):
        raise ValueError('affine must be 4x4')
    if not 1 <= data.ndim <= 7:
        raise ValueError('NIfTI-1 stores 1 to 7 dimensions')
    dim = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    pd = np.ones(8, np.float32)
    pd[1:4] = np.sqrt((affine[:3, :3] ** 2).sum(axis=0))
    if pixdim is not None:
        pd = np.asarray(pixdim, dtype=np.float32).copy()
    hdr = bytearray(348)
    struct.pack_into('<i', hdr, 0, 348)
    struct.pack_into('<8h', hdr, 40, *dim)
    struct.pack_into('<2h', hdr, 70, _CODES[key], out_dtype.itemsize * 8)
    struct.pack_into('<8f', hdr, 76, *[float(v) for v in pd])
    struct.pack_into('<3f', hdr, 108, 352.0, 1.0, 0.0)
    hdr[123] = 10 if data.ndim >= 4 else 2            # mm (+ seconds): informative only
    struct.pack_into('<2h', hdr, 252, 0, 2)
    struct.pack_into('<12f', hdr, 280, *[float(v) for v in affine[:3].ravel()])
    hdr[344:348] = b'n+1\x00'
    if str(path).endswith('.gz') and _CODES[key] in (2, 4, 8, 16, 64):
        lab = _as_label_volume(data)
        if lab is not None and _save_labels_gz(lab, _CODES[key], bytes(hdr) + b'\x00\x00\x00\x00', path):
            return
    with _open(path, 'wb') as f:
        f.write(bytes(hdr))
        f.write(b'\x00\x00\x00\x00')
        if as_dtype is None:
            f.write(np.asfortranarray(data.astype(data.dtype.newbyteorder('<'), copy=False)).tobytes(order='F'))
        else:
            le = out_dtype.newbyteorder('<')
            if data.ndim < 2:
                f.write(np.asarray(data, dtype=le).tobytes(order='F'))
            else:
                for k in range(data.shape[-1]):               # x fastest on disk: the last axis is the slowest
                    f.write(np.asfortranarray(data[..., k]).astype(le).tobytes(order='F'))


# ---- label volumes: run-length gzip writer of the C library ---------------------------------------------------------
LABEL_FAST_PATH = True     # tests switch it off to compare with the zlib path
# How a label volume's .nii.gz is deflated (deploy scripts: --label_gzip).  All three inflate to the same bytes.
#   'small' (default)  run-length tokens, dynamic Huffman codes from the exact token histogram: typically below the size of zlib
#                      level 1 (checked against zlib only above NOISE_FRACTION, where the zlib stream is kept if smaller),
#                      ~20x less CPU than zlib for the float64 volumes of the sequence loop
#   'fast'             run-length tokens, fixed Huffman codes: no counting pass, files 2-4x larger
#   'zlib'             zlib level 1 over the converted volume, as nibabel writes it
LABEL_GZIP_MODE = 'small'
LABEL_GZIP_MODES = ('small', 'fast', 'zlib')
NOISE_FRACTION = 0.06      # 'small': above this compressed / raw ratio (segmentations: 0.01-0.02) zlib level 1 is tried as well


def set_label_gzip(mode):
    global LABEL_GZIP_MODE
    if mode not in LABEL_GZIP_MODES:
        raise ValueError('label gzip mode %r not in %s' % (mode, LABEL_GZIP_MODES))
    LABEL_GZIP_MODE = mode


def _as_label_volume(data):
    """data as a uint8 array in file (Fortran) order if every voxel is an integer in 0..255 (a segmentation), else None."""
    if data.dtype == np.uint8:
        return np.asfortranarray(data)
    if data.dtype.kind not in 'iuf' or data.size == 0:
        return None
    probe = data.ravel(order='K')[:4096]                      # a view for C- or F-contiguous data
    with np.errstate(invalid='ignore'):
        if not np.array_equal(probe.astype(np.uint8), probe):   # MR intensities fail here at once
            return None
        lab = np.asfortranarray(data).astype(np.uint8, order='F')
        return lab if np.array_equal(lab, data) else None


def _save_labels_gz(lab, datatype_code, prefix, path):
    """The .nii.gz of a label volume through ukbb_fcn_gzip_labels_mode (include/ukbb_fcn.h): same inflated bytes as the zlib
    path of save(), ~20x less time for the float64 volumes of the sequence loop.  False = not available / not wanted
    (the caller falls back to zlib)."""
    if not LABEL_FAST_PATH or LABEL_GZIP_MODE == 'zlib':
        return False
    try:                                                        # missing, unloadable or stale library: the zlib path
        from . import _labelgz
        gz, gz_bound = _labelgz.lib.ukbb_fcn_gzip_labels_mode, _labelgz.lib.ukbb_fcn_gzip_labels_bound
    except Exception:
        return False
    mode = _labelgz.FIXED if LABEL_GZIP_MODE == 'fast' else _labelgz.DYNAMIC
    flat = lab.reshape(-1, order='F')
    n = flat.size
    cap = max(1 << 16, len(prefix) * 2 + n * (np.dtype(_DTYPES[datatype_code]).itemsize) // 24)
    for _ in range(2):
        out = np.empty(cap, np.uint8)
        got = gz(flat.ctypes.data, n, datatype_code, prefix, len(prefix), out.ctypes.data, cap, mode)
        if got >= 0:
            blob = memoryview(out)[:got]
            itemsize = np.dtype(_DTYPES[datatype_code]).itemsize
            if mode == _labelgz.DYNAMIC and got > NOISE_FRACTION * (len(prefix) + n * itemsize):
                # noise-like labels (runs of 1-2 voxels): zlib's cross-row matches can beat run-length tokens there; such a
                # volume is not a segmentation, but 'small' keeps its promise (<= zlib level 1) by taking the smaller stream
                # raw deflate between the SAME 10-byte header the other two writers emit (GzipFile(mtime=0, filename='') and
                # csrc/label_gzip.cpp: no name, no time, OS = 255) and the CRC-32 / ISIZE trailer: files of the three paths
                # stay comparable with cmp
                z = zlib.compressobj(GZIP_LEVEL, zlib.DEFLATED, -15)
                le = np.dtype('<' + _DTYPES[datatype_code])
                xfl = 4 if GZIP_LEVEL == 1 else 2 if GZIP_LEVEL == 9 else 0
                parts = [bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, xfl, 0xff]), z.compress(prefix)]
                crc, size = zlib.crc32(prefix), len(prefix)
                for i in range(0, n, 1 << 20):
                    piece = flat[i:i + (1 << 20)].astype(le).tobytes()
                    crc, size = zlib.crc32(piece, crc), size + len(piece)
                    parts.append(z.compress(piece))
                parts.append(z.flush())
                parts.append(int(crc & 0xffffffff).to_bytes(4, 'little') + int(size & 0xffffffff).to_bytes(4, 'little'))
                zb = b''.join(parts)
                if len(zb) < got:
                    blob = zb
            with _AtomicFile(path) as f:
                f.write(blob)
            return True
        if got != -4:                                           # UKBB_ENOMEM: retry once with the guaranteed bound
            raise RuntimeError('ukbb_fcn_gzip_labels_mode failed (%d)' % got)
        cap = int(gz_bound(n, datatype_code, len(prefix)))
    raise RuntimeError('ukbb_fcn_gzip_labels_mode: output bound exceeded')
