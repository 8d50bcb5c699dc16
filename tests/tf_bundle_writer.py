"""Minimal TensorFlow checkpoint-V2 ("tensor bundle") WRITER, test infrastructure only.

Written independently of ukbb_cardiac_amd/tf_checkpoint.py's reader from the same public format
description [TF-recall: core/lib/io/table_format.txt, tensor_bundle.proto]: sorted table with
prefix-compressed keys, restart points every `restart_interval` entries, per-block type byte +
masked CRC32C, index block, 48-byte footer; BundleHeaderProto under the empty key; raw little-endian
tensor bytes in `<prefix>.data-00000-of-00001`.  It exists so the reader can be exercised without
TensorFlow; it does not make the reader "pinned" against real checkpoints.
"""
import struct

import numpy as np

_DT = {np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3, np.dtype('int64'): 9}


def _bitwise(byte):
    crc = byte
    for _ in range(8):
        crc = (crc >> 1) ^ (0x82F63B78 & -(crc & 1))
    return crc


_T = [_bitwise(i) for i in range(256)]


def _crc32c(data):
    crc = 0xFFFFFFFF
    for b in data:
        crc = _T[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(num, wt, payload):
    return _vi((num << 3) | wt) + payload


def _entry_proto(dtype, shape, offset, size, crc):
    dims = b''.join(_field(2, 2, _vi(len(d)) + d) for d in (_field(1, 0, _vi(s)) for s in shape))
    msg = _field(1, 0, _vi(dtype)) + _field(2, 2, _vi(len(dims)) + dims)
    # shard_id 0 is the proto default and is omitted, as protobuf serialisers do
    if offset:
        msg += _field(4, 0, _vi(offset))
    msg += _field(5, 0, _vi(size))
    if crc is not None:
        msg += _field(6, 5, struct.pack('<I', crc))
    return msg


class _BlockBuilder:
    def __init__(self, restart_interval):
        self.ri = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b''

    def add(self, key, value):
        shared = 0
        if self.count % self.ri == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _vi(shared) + _vi(len(key) - shared) + _vi(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _emit(out, block):
    off = len(out)
    trailer = b'\x00'
    out += block + trailer + struct.pack('<I', _masked(_crc32c(block + trailer)))
    return _vi(off) + _vi(len(block))


def write_checkpoint(prefix, tensors, block_size=256, restart_interval=16, tensor_crc=True):
    """tensors: {name: ndarray}.  Small block_size forces several data blocks.  tensor_crc=False omits the
    per-tensor checksums (pure-Python CRC of 8 MB takes seconds); the index blocks always carry theirs."""
    data = bytearray()
    items = {b'': _field(1, 0, _vi(1)) + _field(3, 2, _vi(2) + _field(1, 0, _vi(1)))}   # num_shards=1, version{producer=1}
    for name in tensors:
        arr = np.ascontiguousarray(tensors[name])
        raw = arr.astype(arr.dtype.newbyteorder('<')).tobytes()
        items[name.encode()] = _entry_proto(_DT[arr.dtype], arr.shape, len(data), len(raw),
                                            _masked(_crc32c(raw)) if tensor_crc else None)
        data += raw
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(bytes(data))
    out = bytearray()
    index = _BlockBuilder(1)
    cur = _BlockBuilder(restart_interval)
    for key in sorted(items):
        cur.add(key, items[key])
        if len(cur.buf) >= block_size:
            index.add(cur.last, _emit(out, cur.finish()))
            cur = _BlockBuilder(restart_interval)
    if cur.count:
        index.add(cur.last, _emit(out, cur.finish()))
    meta_handle = _emit(out, _BlockBuilder(1).finish())
    index_handle = _emit(out, index.finish())
    footer = meta_handle + index_handle
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    out += footer
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out))
