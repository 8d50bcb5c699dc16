"""Model weights: seeded synthetic generator, flat packing, and the on-disk blob.

The reference restores TF checkpoint-V2 files downloaded at run time
(``demo_pipeline.py:50-54``, ``common/deploy_network.py:48-49``); neither the
checkpoints nor TensorFlow are available here, so models are stored in a small
self-describing blob (``<model_path>.ukbbw``) holding the *unfolded* tensors
(kernel + BN gamma/beta/moving_mean/moving_variance, or kernel + bias).  BN
folding and MFMA fragment packing happen inside ``ukbb_fcn_create``.

Synthetic weights follow SURVEY.md section 8(d) -- they are this repo's choice,
not the reference's.
"""
import json
import struct
from typing import Dict

import numpy as np

from .arch import ModelArch, MODELS, KIND_FCN

MAGIC = b'UKBBW001'
Params = Dict[str, Dict[str, np.ndarray]]

# Logits biases of the default (seed 1234) synthetic models.  A purely random
# bias leaves one class winning almost everywhere, which would make argmax
# parity tests vacuous; these constants (derived once by
# tests/golden/calibrate_bias.py from the CPU oracle on a phantom, then frozen
# here) centre the per-class logits so every class and many class boundaries
# occur.  They are part of the synthetic-weight definition, not of the reference.
SYNTH_LOGITS_BIAS = {
    'FCN_sa': [1.0192, -0.5849, 2.2997, -2.4434],
    'FCN_la_2ch': [2.5910, -1.7769],
    'FCN_la_4ch': [-1.4404, 1.8248, -0.4128],
    'FCN_la_4ch_seg4': [3.6111, 1.1227, -0.1888, 3.4944, 0.8714, -3.1443],
    'UNet_ao': [-0.0696, 1.2976, -0.1832],
}


def synthetic_params(arch: ModelArch, seed: int = 1234) -> Params:
    rng = np.random.default_rng(seed)
    params: Params = {}
    for s in arch.layer_specs():
        kh, kw, a, b = s.kernel_shape
        cin, cout = (b, a) if s.transposed else (a, b)
        fan_in = kh * kw * cin
        if s.transposed:
            # stride-2 transposed conv: each output sees ~ (k/s)^2 taps
            fan_in = max(1, fan_in // 4)
        p = {'kernel': rng.normal(0.0, np.sqrt(2.0 / fan_in), size=s.kernel_shape).astype(np.float32)}
        if s.has_bn:
            p['gamma'] = rng.uniform(0.5, 1.5, size=cout).astype(np.float32)
            p['beta'] = rng.normal(0.0, 0.1, size=cout).astype(np.float32)
            p['mean'] = rng.normal(0.0, 0.1, size=cout).astype(np.float32)
            p['var'] = rng.uniform(0.5, 1.5, size=cout).astype(np.float32)
        if s.has_bias:
            p['bias'] = rng.normal(0.0, 0.1, size=cout).astype(np.float32)
            if seed == 1234 and arch.name in SYNTH_LOGITS_BIAS:
                p['bias'] = np.asarray(SYNTH_LOGITS_BIAS[arch.name], dtype=np.float32)
        params[s.name] = p
    return params


def embed_unidirectional_lstm(params: Params, n_hidden: int) -> Params:
    """The single-direction ConvLSTM head (common/network_ao.py:214-252 Conv_LSTM; train_network_ao.py --bidirectional=False) expressed
    in the bidirectional layer set the engine is built for (BiConv_LSTM :255-319): ``params`` holds the U-Net layers plus 'lstm' (the one
    cell) and 'lstm_conv' (the 1x1 logits conv on its output); returned are the same U-Net layers plus 'lstm_fw' = that cell, 'lstm_bw' = a
    cell of all-zero kernel and bias, 'lstm_out' = [W; 0] over concat([h_fw, h_bw]).
    Exact, not approximate: from the zero state a zero cell gives i = o = 1/2, j = tanh(0) = 0, so c' = sigmoid(f + 1) * 0 + 1/2 * 0 = 0 and
    h' = tanh(0) * 1/2 = 0 at every step, and the output conv adds 16 products 0 * 0 to the single-direction sum.
    (tests/test_unidirectional_lstm.py: the oracle's Conv_LSTM against its BiConv_LSTM on the embedded set -- backward maps exactly zero,
    logits equal to the rounding of the summation order -- and the engine on the embedded set against the oracle's Conv_LSTM.)"""
    out = {k: v for k, v in params.items() if k not in ('lstm', 'lstm_conv')}
    cell, conv = params['lstm'], params['lstm_conv']
    out['lstm_fw'] = {'kernel': np.asarray(cell['kernel'], np.float32), 'bias': np.asarray(cell['bias'], np.float32)}
    out['lstm_bw'] = {'kernel': np.zeros_like(out['lstm_fw']['kernel']), 'bias': np.zeros_like(out['lstm_fw']['bias'])}
    k = np.asarray(conv['kernel'], np.float32)
    if k.shape[:3] != (1, 1, n_hidden):
        raise ValueError('lstm_conv kernel %s, expected (1, 1, %d, n_class)' % (k.shape, n_hidden))
    out['lstm_out'] = {'kernel': np.concatenate([k, np.zeros_like(k)], axis=2), 'bias': np.asarray(conv['bias'], np.float32)}
    return out


def threshold_params(arch: ModelArch, thresholds=None, slope: float = 40.0) -> Params:
    """A parameter set whose label map is a smooth function of the image: label = number of `thresholds` below the intensity after
    two 3x3 means.  Random weights (synthetic_params) give noise-like label maps -- millions of runs per subject, which makes the label
    writer, not the network or the reader, the longest host stage of a cohort run; a trained model segments a few compact regions.
    This set reproduces THAT output statistic at the same arithmetic cost (every kernel does the same work whatever the weights):
    channel 0 carries the image through conv0_0 and conv0_1 (3x3 means), same_dim0, out0 and out1 with identity batch
    norm, every other weight is zero, and the logits are the upper envelope of lines that cross at the thresholds.
    FCN models only (tools/shard_rehearsal.py, tools/bench_subject.py)."""
    from .arch import KIND_FCN
    if arch.kind != KIND_FCN:
        raise ValueError('threshold_params: FCN models only')
    nc = arch.n_class
    if thresholds is None:
        thresholds = [0.3 + 0.5 * c / max(1, nc - 2) for c in range(nc - 1)]     # 0.3 .. 0.8 of the [0, 1] range rescale_intensity produces
    thresholds = [float(t) for t in thresholds]
    if len(thresholds) != nc - 1 or sorted(thresholds) != thresholds:
        raise ValueError('need n_class - 1 increasing thresholds')
    params: Params = {}
    for s in arch.layer_specs():
        kh, kw, a, b = s.kernel_shape
        k = np.zeros(s.kernel_shape, np.float32)
        p = {'kernel': k}
        if s.has_bn:                                                          # scale 1, shift 0 after folding (epsilon 1e-3)
            p['gamma'] = np.full(b, np.sqrt(1.0 + 1e-3), np.float32)
            p['beta'] = np.zeros(b, np.float32)
            p['mean'] = np.zeros(b, np.float32)
            p['var'] = np.ones(b, np.float32)
        if s.name in ('conv0_0', 'conv0_1'):
            k[:, :, 0, 0] = 1.0 / 9.0                                         # two 3x3 means (zero padding at the border)
        elif s.name in ('same_dim0', 'out0', 'out1'):
            k[0, 0, 0, 0] = 1.0                                               # out0: channel 0 of the level-0 block of the concat
        elif s.name == 'logits':
            bias = np.zeros(nc, np.float32)
            for c in range(1, nc):                                            # line c = line c-1 + slope (x - t_c): the upper envelope switches at t_c
                k[0, 0, 0, c] = k[0, 0, 0, c - 1] + slope
                bias[c] = bias[c - 1] - slope * thresholds[c - 1]
            p['bias'] = bias
        elif s.has_bias:
            p['bias'] = np.zeros(b, np.float32)
        params[s.name] = p
    return params


def pack_flat(arch: ModelArch, params: Params) -> np.ndarray:
    """Flatten to the canonical order ``ukbb_fcn_create`` expects: per layer of
    ``arch.layer_specs()``: kernel (C order), then gamma, beta, mean, var -- or
    bias for the logits layer."""
    chunks = []
    for s in arch.layer_specs():
        p = params[s.name]
        k = np.ascontiguousarray(p['kernel'], dtype=np.float32)
        if tuple(k.shape) != tuple(s.kernel_shape):
            raise ValueError('layer %s: kernel shape %s, expected %s' % (s.name, k.shape, s.kernel_shape))
        chunks.append(k.ravel())
        if s.has_bn:
            for key in ('gamma', 'beta', 'mean', 'var'):
                chunks.append(np.ascontiguousarray(p[key], dtype=np.float32).ravel())
        if s.has_bias:
            chunks.append(np.ascontiguousarray(p['bias'], dtype=np.float32).ravel())
    flat = np.concatenate(chunks)
    assert flat.size == arch.n_weight_floats()
    return flat


def unpack_flat(arch: ModelArch, flat: np.ndarray) -> Params:
    params: Params = {}
    off = 0
    for s in arch.layer_specs():
        kh, kw, a, b = s.kernel_shape
        cout = a if s.transposed else b
        n = kh * kw * a * b
        p = {'kernel': flat[off:off + n].reshape(s.kernel_shape).copy()}
        off += n
        if s.has_bn:
            for key in ('gamma', 'beta', 'mean', 'var'):
                p[key] = flat[off:off + cout].copy()
                off += cout
        if s.has_bias:
            p['bias'] = flat[off:off + cout].copy()
            off += cout
        params[s.name] = p
    assert off == flat.size
    return params


def save_blob(path: str, arch: ModelArch, params: Params) -> None:
    flat = pack_flat(arch, params)
    hdr = json.dumps({
        'name': arch.name, 'kind': arch.kind, 'n_class': arch.n_class, 'n_level': arch.n_level,
        'n_filter': list(arch.n_filter), 'n_block': list(arch.n_block),
        'same_dim': arch.same_dim, 'fc': arch.fc, 'n_floats': int(flat.size),
    }).encode()
    with open(path, 'wb') as f:
        f.write(MAGIC)
        f.write(struct.pack('<I', len(hdr)))
        f.write(hdr)
        f.write(flat.astype('<f4').tobytes())


def load_blob(path: str):
    with open(path, 'rb') as f:
        if f.read(8) != MAGIC:
            raise ValueError('%s: not a UKBBW001 weight blob' % path)
        (n,) = struct.unpack('<I', f.read(4))
        h = json.loads(f.read(n).decode())
        flat = np.frombuffer(f.read(), dtype='<f4').astype(np.float32)
    arch = ModelArch(h['name'], h['kind'], h['n_class'], h['n_level'], tuple(h['n_filter']),
                     tuple(h['n_block']), h['same_dim'], h['fc'])
    if flat.size != h['n_floats'] or flat.size != arch.n_weight_floats():
        raise ValueError('%s: truncated or inconsistent weight blob' % path)
    return arch, unpack_flat(arch, flat)


def fold_bn(p, eps: float = 1e-3):
    """Host-side restatement of the fold ``ukbb_fcn_create`` performs
    (fp32, same operation order): scale = gamma / sqrt(var + eps);
    W' = W * scale (per C_out); b' = beta - mean * scale."""
    k = p['kernel'].astype(np.float32)
    if 'gamma' in p:
        scale = (p['gamma'].astype(np.float32) /
                 np.sqrt(p['var'].astype(np.float32) + np.float32(eps))).astype(np.float32)
        shift = (p['beta'].astype(np.float32) - (p['mean'].astype(np.float32) * scale)).astype(np.float32)
        return scale, shift
    return np.ones(k.shape[-1], np.float32), p['bias'].astype(np.float32)
