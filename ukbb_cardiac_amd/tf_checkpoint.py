"""Read the reference's trained models: TensorFlow checkpoint-V2 ("tensor bundle") files.

The reference restores ``<model_path>.index`` / ``<model_path>.data-00000-of-00001`` through
``tf.train.import_meta_graph`` + ``saver.restore`` (common/deploy_network.py:44-49,
common/deploy_network_ao.py:53-58; the files are downloaded by demo_pipeline.py:50-54).  This module
parses those two files directly (no TensorFlow) and maps the variables onto the flat weight layout
of ``ukbb_fcn_create`` (include/ukbb_fcn.h).  SURVEY.md section 8(f) row 1.

File formats, restated from TensorFlow's public sources **[TF-recall: tensorflow/core/util/
tensor_bundle/tensor_bundle.cc, core/lib/io/table_format.txt, core/protobuf/tensor_bundle.proto --
not available in this environment, so this reader is pinned only by its own round-trip tests
(tests/test_tf_checkpoint.py), not by a checkpoint TensorFlow wrote]**:

* ``.index`` is a LevelDB-style sorted table: data blocks of prefix-compressed
  (shared, non_shared, value_len, key_delta, value) entries followed by a uint32 restart array and its
  length; every block is followed by a 1-byte compression type (0 none, 1 snappy) and a masked
  CRC32C; an index block maps separator keys to block handles; a 48-byte footer holds the metaindex
  and index handles and the magic 0xdb4775248b80fb57.
* key ``""`` -> ``BundleHeaderProto`` (num_shards, endianness, version); every other key is a
  tensor name -> ``BundleEntryProto`` (dtype, shape, shard_id, offset, size, crc32c).
* ``.data-SSSSS-of-NNNNN`` holds the raw little-endian row-major tensor bytes.

Variable names.  ``build_FCN`` (common/network.py:170-230) only uses ``tf.name_scope``, which does not
scope variables, so ``tf.layers`` numbers them in creation order: ``conv2d/kernel``,
``conv2d_1/kernel``, ..., ``batch_normalization[_k]/{gamma,beta,moving_mean,moving_variance}``, and
the last ``conv2d_K/{kernel,bias}``.  ``UNet`` (common/network_ao.py:18-64) uses
``tf.variable_scope('UNet')`` / ``conv{l}`` / ``conv{l}_up`` / ``conv_out``, and the numbering restarts
in every scope **[TF-recall]**.  Optimizer slots (``.../Adam``, ``.../Adam_1``, ``beta1_power``, ...)
and ``global_step`` are ignored.
"""
import os
import re
import struct
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np

from .arch import KIND_FCN, KIND_UNET, KIND_UNET_LSTM, MODELS, ModelArch

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_LEN = 48
BLOCK_TRAILER = 5

# tensorflow/core/framework/types.proto [TF-recall]
_DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 5: np.dtype('<i2'),
           6: np.dtype('i1'), 9: np.dtype('<i8'), 10: np.dtype('?'), 17: np.dtype('<u2'), 19: np.dtype('<f2'),
           22: np.dtype('<u4'), 23: np.dtype('<u8')}


class CheckpointError(ValueError):
    pass


# ---- CRC32C (Castagnoli), masked as in leveldb / TF -------------------------------------------
def _crc_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC_TABLE = _crc_table()


def crc32c(data: bytes, crc: int = 0) -> int:
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ---- varints / minimal protobuf wire reader ----------------------------------------------------
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = 0
    val = 0
    while True:
        if pos >= len(buf):
            raise CheckpointError('truncated varint')
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 70:
            raise CheckpointError('varint too long')


def _proto_fields(buf: bytes):
    """Yield (field_number, wire_type, value) of one protobuf message."""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]; pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]; pos += 4
        else:
            raise CheckpointError('unsupported protobuf wire type %d' % wt)
        if pos > len(buf):
            raise CheckpointError('truncated protobuf field')
        yield field, wt, v


def _to_int64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


# ---- snappy (raw format) -----------------------------------------------------------------------
def snappy_uncompress(src: bytes) -> bytes:
    n, pos = _varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]; pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], 'little'); pos += nb
            ln += 1
            out += src[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 2], 'little'); pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], 'little'); pos += 4
        if off == 0 or off > len(out):
            raise CheckpointError('corrupt snappy copy')
        for _ in range(ln):                              # may overlap its own output
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointError('snappy length mismatch (%d != %d)' % (len(out), n))
    return bytes(out)


# ---- sorted table ------------------------------------------------------------------------------
def _read_block(buf: bytes, offset: int, size: int, verify: bool) -> bytes:
    if offset + size + BLOCK_TRAILER > len(buf):
        raise CheckpointError('block handle outside the file')
    data = buf[offset:offset + size]
    ctype = buf[offset + size]
    stored = struct.unpack_from('<I', buf, offset + size + 1)[0]
    if verify and stored != mask_crc(crc32c(buf[offset:offset + size + 1])):
        raise CheckpointError('index block at %d: CRC mismatch' % offset)
    if ctype == 0:
        return data
    if ctype == 1:
        return snappy_uncompress(data)
    raise CheckpointError('unknown block compression type %d' % ctype)


def _block_entries(block: bytes):
    if len(block) < 4:
        raise CheckpointError('block too small')
    nrestart = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestart
    if end < 0:
        raise CheckpointError('bad restart array')
    pos = 0
    key = b''
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise CheckpointError('corrupt block entry')
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(buf: bytes, verify: bool = True) -> "OrderedDict[bytes, bytes]":
    if len(buf) < FOOTER_LEN:
        raise CheckpointError('file shorter than a table footer')
    footer = buf[-FOOTER_LEN:]
    if struct.unpack_from('<Q', footer, FOOTER_LEN - 8)[0] != TABLE_MAGIC:
        raise CheckpointError('not a checkpoint-V2 index (bad table magic)')
    pos = 0
    _, pos = _varint(footer, pos)           # metaindex handle (unused)
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isz, pos = _varint(footer, pos)
    out: "OrderedDict[bytes, bytes]" = OrderedDict()
    for _, handle in _block_entries(_read_block(buf, ioff, isz, verify)):
        boff, p = _varint(handle, 0)
        bsz, p = _varint(handle, p)
        for k, v in _block_entries(_read_block(buf, boff, bsz, verify)):
            out[k] = v
    return out


# ---- the bundle --------------------------------------------------------------------------------
class Entry:
    __slots__ = ('dtype', 'shape', 'shard', 'offset', 'size', 'crc')

    def __init__(self):
        self.dtype = 0; self.shape = (); self.shard = 0; self.offset = 0; self.size = 0; self.crc = None


def _parse_entry(buf: bytes) -> Entry:
    e = Entry()
    for f, wt, v in _proto_fields(buf):
        if f == 1: e.dtype = v
        elif f == 2:
            dims = []
            for f2, _, v2 in _proto_fields(v):
                if f2 == 2:                               # TensorShapeProto.dim
                    size = 0
                    for f3, _, v3 in _proto_fields(v2):
                        if f3 == 1: size = _to_int64(v3)
                    dims.append(size)
                elif f2 == 3 and v2:
                    raise CheckpointError('tensor of unknown rank')
            e.shape = tuple(dims)
        elif f == 3: e.shard = v
        elif f == 4: e.offset = _to_int64(v)
        elif f == 5: e.size = _to_int64(v)
        elif f == 6: e.crc = struct.unpack('<I', v)[0]
        elif f == 7:
            raise CheckpointError('partitioned (sliced) variables are not supported')
    return e


class CheckpointReader:
    """``tf.train.load_checkpoint(prefix)``-like access: ``names()``, ``get_tensor(name)``."""

    def __init__(self, prefix: str, verify_index: bool = True):
        self.prefix = prefix
        index = prefix + '.index'
        if not os.path.isfile(index):
            raise FileNotFoundError(index)
        with open(index, 'rb') as f:
            table = read_table(f.read(), verify_index)
        if b'' not in table:
            raise CheckpointError('%s: no bundle header entry' % index)
        self.num_shards, endian = 1, 0
        for f_, _, v in _proto_fields(table[b'']):
            if f_ == 1: self.num_shards = v
            elif f_ == 2: endian = v
        if endian != 0:
            raise CheckpointError('big-endian bundles are not supported')
        self.entries: Dict[str, Entry] = OrderedDict()
        for k, v in table.items():
            if k:
                self.entries[k.decode('utf-8')] = _parse_entry(v)
        self._shards: Dict[int, np.memmap] = {}

    def names(self) -> List[str]:
        return list(self.entries)

    def shape(self, name: str) -> Tuple[int, ...]:
        return self.entries[name].shape

    def _shard(self, i: int):
        if i not in self._shards:
            path = '%s.data-%05d-of-%05d' % (self.prefix, i, self.num_shards)
            if not os.path.isfile(path):
                raise FileNotFoundError(path)
            self._shards[i] = np.memmap(path, dtype=np.uint8, mode='r')
        return self._shards[i]

    def get_tensor(self, name: str, verify_crc: bool = False) -> np.ndarray:
        if name not in self.entries:
            raise KeyError('%s not in checkpoint %s' % (name, self.prefix))
        e = self.entries[name]
        if e.dtype not in _DTYPES:
            raise CheckpointError('%s: unsupported dtype enum %d' % (name, e.dtype))
        dt = _DTYPES[e.dtype]
        n = int(np.prod(e.shape, dtype=np.int64)) if e.shape else 1
        if n * dt.itemsize != e.size:
            raise CheckpointError('%s: %d bytes stored, shape %s needs %d' % (name, e.size, e.shape, n * dt.itemsize))
        data = self._shard(e.shard)
        if e.offset < 0 or e.offset + e.size > data.size:
            raise CheckpointError('%s: data range outside shard %d' % (name, e.shard))
        raw = bytes(data[e.offset:e.offset + e.size])
        if verify_crc and e.crc is not None and mask_crc(crc32c(raw)) != e.crc:
            raise CheckpointError('%s: tensor CRC mismatch' % name)
        return np.frombuffer(raw, dtype=dt).reshape(e.shape).copy()


# ---- variable naming of the reference graphs ---------------------------------------------------
def _numbered(base: str, k: int) -> str:
    return base if k == 0 else '%s_%d' % (base, k)


_BN_KEYS = (('gamma', 'gamma'), ('beta', 'beta'), ('mean', 'moving_mean'), ('var', 'moving_variance'))


_LSTM_VAR = re.compile(r'^LSTM/(forward|backward)/([^/]+)/(kernel|biases|bias)$')
LSTM_CELL_DEFAULT = 'conv_lstm_cell'


def lstm_cell_variables(names, direction: str) -> Dict[str, str]:
    """{'kernel': ..., 'bias': ...} of the ConvLSTM cell created under variable_scope('LSTM') / (direction)
    (network_ao.py:270-295).  The cell's own scope is whatever name TF gave the cell object -- ``conv_lstm_cell`` for
    ConvLSTMCell, ``conv_2d_lstm_cell`` for the Conv2DLSTMCell subclass the reference instantiates (:277,:290), depending
    on the TF 1.x release -- so it is matched, not assumed: the single scope below LSTM/<direction>/ that holds a
    ``kernel`` and a ``biases`` (or ``bias``) variable.  Optimizer slots (``.../kernel/Adam``) have one more path
    component and never match."""
    found: Dict[str, Dict[str, str]] = {}
    for n in names:
        m = _LSTM_VAR.match(n)
        if m and m.group(1) == direction:
            found.setdefault(m.group(2), {})['kernel' if m.group(3) == 'kernel' else 'bias'] = n
    full = {cell: d for cell, d in found.items() if 'kernel' in d and 'bias' in d}
    if len(full) != 1:
        raise CheckpointError('LSTM/%s: expected one cell scope with kernel + biases, found %s' % (direction, sorted(found) or 'none'))
    return next(iter(full.values()))


def variable_names(arch: ModelArch, available=None) -> "OrderedDict[str, Dict[str, str]]":
    """layer name (arch.layer_specs order) -> {'kernel': tf name, 'gamma': ..., ...}.  ``available`` (the names in a
    checkpoint) lets the ConvLSTM cell scope be matched instead of assumed."""
    out: "OrderedDict[str, Dict[str, str]]" = OrderedDict()
    if arch.kind == KIND_FCN:
        nconv = nbn = 0
        for s in arch.layer_specs():
            d = {'kernel': _numbered('conv2d', nconv) + '/kernel'}
            if s.has_bias:
                d['bias'] = _numbered('conv2d', nconv) + '/bias'
            nconv += 1
            if s.has_bn:
                for key, tfk in _BN_KEYS:
                    d[key] = _numbered('batch_normalization', nbn) + '/' + tfk
                nbn += 1
            out[s.name] = d
        return out
    counters: Dict[Tuple[str, str], int] = {}

    def take(scope, base):
        k = counters.get((scope, base), 0)
        counters[(scope, base)] = k + 1
        return '%s/%s' % (scope, _numbered(base, k))

    for s in arch.layer_specs():
        if s.name in ('lstm_fw', 'lstm_bw'):
            # tf.contrib.rnn.Conv2DLSTMCell under variable_scope('LSTM') / ('forward' | 'backward')
            # (network_ao.py:270-295) [TF-recall: the cell's scope name and its `kernel` / `biases` names]
            direction = 'forward' if s.name == 'lstm_fw' else 'backward'
            if available is not None:
                out[s.name] = lstm_cell_variables(available, direction)
            else:
                base = 'LSTM/%s/%s' % (direction, LSTM_CELL_DEFAULT)
                out[s.name] = {'kernel': base + '/kernel', 'bias': base + '/biases'}
            continue
        if s.name == 'lstm_out':
            out[s.name] = {'kernel': 'LSTM/output/conv2d/kernel', 'bias': 'LSTM/output/conv2d/bias'}   # :298-309
            continue
        if s.name == 'logits':
            scope = 'UNet/conv_out'
        elif s.name.startswith('conv'):
            scope = 'UNet/conv%s' % s.name[4:].split('_')[0]
        else:                                               # up{l}_t, up{l}_{i}
            scope = 'UNet/conv%s_up' % s.name[2:].split('_')[0]
        layer = take(scope, 'conv2d_transpose' if s.transposed else 'conv2d')
        d = {'kernel': layer + '/kernel'}
        if s.has_bias:
            d['bias'] = layer + '/bias'
        if s.has_bn:
            bn = take(scope, 'batch_normalization')
            for key, tfk in _BN_KEYS:
                d[key] = bn + '/' + tfk
        out[s.name] = d
    return out


def unidirectional_lstm_variables(available) -> Optional[Dict[str, Dict[str, str]]]:
    """Variable names of the single-direction head Conv_LSTM (network_ao.py:214-252: one Conv2DLSTMCell and the logits conv directly under
    variable_scope('LSTM')), or None when the checkpoint has the bidirectional scopes (or no LSTM at all).
    [TF-recall: LSTM/<cell scope>/{kernel,biases}, LSTM/conv2d/{kernel,bias}]"""
    names = set(available)
    if any(n.startswith(('LSTM/forward/', 'LSTM/backward/', 'LSTM/output/')) for n in names):
        return None
    cell_k = sorted(n for n in names if re.match(r'^LSTM/[^/]+/kernel$', n) and not n.startswith('LSTM/conv2d'))
    if len(cell_k) != 1 or 'LSTM/conv2d/kernel' not in names:
        return None
    scope = cell_k[0][:-len('/kernel')]
    bias = scope + '/biases' if scope + '/biases' in names else scope + '/bias'
    if bias not in names or 'LSTM/conv2d/bias' not in names:
        return None
    return {'lstm': {'kernel': cell_k[0], 'bias': bias}, 'lstm_conv': {'kernel': 'LSTM/conv2d/kernel', 'bias': 'LSTM/conv2d/bias'}}


def infer_arch(reader: CheckpointReader) -> ModelArch:
    """Hyper-parameters from the kernel shapes (the .meta graph is not parsed)."""
    names = set(reader.names())
    unet = any(n.startswith('UNet/') for n in names)
    if unet:
        n_filter, n_block = [], []
        l = 0
        while 'UNet/conv%d/conv2d/kernel' % l in names:
            k = 0
            while 'UNet/conv%d/%s/kernel' % (l, _numbered('conv2d', k)) in names:
                k += 1
            n_filter.append(reader.shape('UNet/conv%d/conv2d/kernel' % l)[3])
            n_block.append(k)
            l += 1
        lstm = any(_LSTM_VAR.match(n) for n in names)
        # the single-direction head Conv_LSTM (network_ao.py:214-252; train_network_ao.py --bidirectional=False) keeps its cell directly
        # under LSTM/ -- LSTM/<cell>/{kernel,biases} + LSTM/conv2d -- with no forward / backward / output scopes.  It is served by the
        # bidirectional engine with a zero backward cell (weights.embed_unidirectional_lstm: exact), see checkpoint_to_params
        if unidirectional_lstm_variables(names) is not None:
            uv = unidirectional_lstm_variables(names)
            n_hidden = reader.shape(uv['lstm']['kernel'])[3] // 4
            cand = ModelArch('UNet-LSTM_custom', KIND_UNET_LSTM, reader.shape(uv['lstm_conv']['kernel'])[3],
                             n_level=len(n_filter), n_filter=tuple(n_filter), n_block=tuple(n_block), same_dim=n_hidden, fc=9)
            for m in MODELS.values():
                if (m.kind, m.n_class, m.n_level, tuple(m.n_filter), tuple(m.n_block), m.same_dim, m.fc) == \
                   (cand.kind, cand.n_class, cand.n_level, tuple(cand.n_filter), tuple(cand.n_block), cand.same_dim, cand.fc):
                    return m
            return cand
        if not n_filter or (not lstm and 'UNet/conv_out/conv2d/kernel' not in names):
            raise CheckpointError('UNet checkpoint without the expected UNet/conv{l}/conv2d variables')
        if lstm:
            # the unrolled step count is not recoverable from the variables: the released model is trained with a
            # 9-frame window (model name ...tw9_h16_bidir..., deploy_network_ao.py:36-38,130)
            n_hidden = reader.shape(lstm_cell_variables(names, 'forward')['kernel'])[3] // 4
            cand = ModelArch('UNet-LSTM_custom', KIND_UNET_LSTM, reader.shape('LSTM/output/conv2d/kernel')[3],
                             n_level=len(n_filter), n_filter=tuple(n_filter), n_block=tuple(n_block),
                             same_dim=n_hidden, fc=9)
        else:
            n_class = reader.shape('UNet/conv_out/conv2d/kernel')[3]
            cand = ModelArch('UNet_custom', KIND_UNET, n_class, n_level=len(n_filter), n_filter=tuple(n_filter),
                             n_block=tuple(n_block))
    else:
        shapes = []
        k = 0
        while _numbered('conv2d', k) + '/kernel' in names:
            shapes.append(reader.shape(_numbered('conv2d', k) + '/kernel'))
            k += 1
        if len(shapes) < 4:
            raise CheckpointError('no conv2d[_k]/kernel variables: not a build_FCN checkpoint')
        n3 = 0
        while n3 < len(shapes) and shapes[n3][0] == 3:
            n3 += 1
        n_filter, n_block = [], []
        for sh in shapes[:n3]:
            if n_filter and sh[2] == sh[3] == n_filter[-1]:
                n_block[-1] += 1
            else:
                n_filter.append(sh[3]); n_block.append(1)
        n_level = len(n_filter)
        rest = shapes[n3:]
        if len(rest) != n_level + 3:
            raise CheckpointError('expected %d 1x1 convs after the encoder, found %d' % (n_level + 3, len(rest)))
        same_dim, fc, n_class = rest[0][3], rest[n_level][3], rest[-1][3]
        cand = ModelArch('FCN_custom', KIND_FCN, n_class, n_level=n_level, n_filter=tuple(n_filter),
                         n_block=tuple(n_block), same_dim=same_dim, fc=fc)
    for m in MODELS.values():                              # reuse the canonical name when it is one of the five
        if (m.kind, m.n_class, m.n_level, tuple(m.n_filter), tuple(m.n_block), m.same_dim, m.fc) == \
           (cand.kind, cand.n_class, cand.n_level, tuple(cand.n_filter), tuple(cand.n_block), cand.same_dim, cand.fc):
            return m
    return cand


def checkpoint_to_params(prefix: str, arch: Optional[ModelArch] = None, verify_crc: bool = False):
    """-> (arch, params) with params[layer] = {'kernel', 'gamma', 'beta', 'mean', 'var'} / {'kernel', 'bias'}."""
    reader = CheckpointReader(prefix)
    if arch is None:
        arch = infer_arch(reader)
    params = {}
    specs = {s.name: s for s in arch.layer_specs()}
    uni = unidirectional_lstm_variables(reader.names()) if arch.kind == KIND_UNET_LSTM else None
    wanted = variable_names(arch, None if uni else reader.names())
    if uni:                                                  # Conv_LSTM head: its two layers are read under their own names and embedded below
        for layer in ('lstm_fw', 'lstm_bw', 'lstm_out'):
            wanted.pop(layer)
        wanted.update(uni)
        nh = arch.same_dim
        shapes = {'lstm': (3, 3, arch.n_filter[0] + nh, 4 * nh), 'lstm_conv': (1, 1, nh, arch.n_class)}
    for layer, names in wanted.items():
        p = {}
        for key, tfname in names.items():
            t = reader.get_tensor(tfname, verify_crc).astype(np.float32)
            p[key] = t
        want = tuple(specs[layer].kernel_shape) if layer in specs else shapes[layer]
        if tuple(p['kernel'].shape) != want:
            raise CheckpointError('%s (%s): kernel shape %s, expected %s' % (layer, names['kernel'], p['kernel'].shape, want))
        params[layer] = p
    if uni:
        from .weights import embed_unidirectional_lstm
        params = embed_unidirectional_lstm(params, arch.same_dim)
    return arch, params


def is_checkpoint(prefix: str) -> bool:
    return os.path.isfile(prefix + '.index')


def main(argv=None):
    import argparse
    from .weights import save_blob
    MODEL_EXT = '.ukbbw'
    ap = argparse.ArgumentParser(description='TF checkpoint-V2 -> %s weight blob' % MODEL_EXT)
    ap.add_argument('model_path', help='checkpoint prefix, e.g. trained_model/FCN_sa')
    ap.add_argument('--model', choices=sorted(MODELS), default=None, help='default: inferred from the variable shapes')
    ap.add_argument('-o', '--output', default=None)
    ap.add_argument('--verify-crc', action='store_true')
    ap.add_argument('--list', action='store_true', help='only list the variables')
    args = ap.parse_args(argv)
    if args.list:
        r = CheckpointReader(args.model_path)
        for n in r.names():
            print('%-60s %s' % (n, r.shape(n)))
        return 0
    arch, params = checkpoint_to_params(args.model_path, MODELS[args.model] if args.model else None, args.verify_crc)
    out = args.output or args.model_path + MODEL_EXT
    save_blob(out, arch, params)
    print('%s: %s, %d floats -> %s' % (args.model_path, arch.name, arch.n_weight_floats(), out))
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
