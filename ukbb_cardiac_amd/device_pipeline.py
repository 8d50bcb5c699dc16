"""Sequence segmentation with the array handling of the deploy loop moved onto the GPU
(SURVEY.md section 8(f) row 3).

``segment_sequence_device`` produces exactly what ``pipeline.segment_sequence`` (the numpy mirror
of common/deploy_network.py:86-116) produces, but only two things cross PCIe per subject: the raw
float32 volume in, a uint8 label volume out.  The percentile sort of common/image_utils.py:72
(~20 M voxels per short-axis subject, the largest host cost of the reference loop) becomes an exact
4-pass radix select on the device, clip / rescale / pad / transpose one fused kernel, and the
label transposes plus the per-frame class counts of the ES pick (:125-130) another.

numpy's ``percentile(..., method='linear')`` is reproduced bit for bit: the device returns the two
neighbouring order statistics of each percentile, and the interpolation between them is done here
by numpy itself (``np.quantile`` on the two-element array with the same fractional index).
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from .pipeline import pad_amounts


def percentile_ranks(n, q):
    """0-based ranks (k, k+1) and the fractional part numpy's linear method uses for percentile q of n
    values when q is given as a tuple/array (as image_utils.py:72 does): float64 quantiles q/100,
    virtual index (n - 1) * quantile.  (A Python-scalar q on a float32 array takes a float32 path in
    numpy 2 and is not what the reference calls.)"""
    quantile = np.true_divide(np.float64(q), 100)
    virtual = (n - 1) * quantile
    k = int(math.floor(virtual))
    return k, min(k + 1, n - 1), np.float64(virtual - k)


def lerp_like_numpy(a_k, a_k1, gamma):
    """np.percentile's interpolation between the neighbouring order statistics a[k], a[k+1] (float32)."""
    # gamma goes in as a float64 ARRAY: numpy demotes scalar quantiles to the data's float32, which is not
    # the path a tuple of percentiles takes
    return np.quantile(np.array([a_k, a_k1], dtype=np.float32), np.array([gamma], dtype=np.float64))[0]


def device_percentiles(vol_t, qs, stream=0):
    """Exact np.percentile(volume, qs) of a dense float32 torch tensor on the GPU."""
    n = vol_t.numel()
    ranks, gammas = [], []
    for q in qs:
        k, k1, g = percentile_ranks(n, q)
        ranks += [k, k1]
        gammas.append(g)
    r = (C.c_uint64 * len(ranks))(*ranks)
    out = np.empty(len(ranks), np.float32)
    _lib.check(_lib.lib.ukbb_fcn_select_kth(vol_t.data_ptr(), n, r, len(ranks), _lib.f32ptr(out), stream), 'ukbb_fcn_select_kth')
    return [lerp_like_numpy(out[2 * i], out[2 * i + 1], gammas[i]) for i in range(len(qs))]


def segment_sequence_device(image, engine, batch_slices=128, thres=(1, 99), return_aux=False):
    """(X,Y,Z,T) float32 volume -> float64 label volume of the same shape, like pipeline.segment_sequence.

    ``image`` is NOT modified (the reference clips it in place); callers that
    save image frames afterwards clip them with the returned bounds (``aux['clip']``).
    ``aux['counts'][t, c]`` = voxels of class c in frame t (input of the ES pick)."""
    import torch
    if image.ndim != 4:
        raise ValueError('expected a 4-D (X,Y,Z,T) sequence, got shape %s' % (image.shape,))
    if image.dtype != np.float32:
        raise TypeError('device pre-processing is exact for float32 volumes only (got %s); use pipeline.segment_sequence'
                        % image.dtype)
    X, Y, Z, T = image.shape
    dev = torch.device('cuda', engine.device)
    stream = torch.cuda.current_stream(dev).cuda_stream
    src = image if (image.flags.f_contiguous or image.flags.c_contiguous) else np.asfortranarray(image)
    vol = torch.from_numpy(src).to(dev)                          # dense copy, strides preserved
    lo, hi = device_percentiles(vol, thres, stream)
    X2, Y2, x_pre, _, y_pre, _ = pad_amounts(X, Y)
    n = T * Z
    batch = torch.empty((n, X2, Y2), dtype=torch.float32, device=dev)
    sx, sy, sz, st = vol.stride()
    _lib.check(_lib.lib.ukbb_fcn_rescale_pack(vol.data_ptr(), X, Y, Z, T, sx, sy, sz, st, float(lo), float(hi),
                                              X2, Y2, x_pre, y_pre, batch.data_ptr(), stream), 'ukbb_fcn_rescale_pack')
    pred = torch.empty((n, X2, Y2), dtype=torch.int32, device=dev)
    engine.reserve(min(batch_slices, n), X2, Y2)
    for i in range(0, n, batch_slices):
        m = min(batch_slices, n - i)
        engine.run_device(batch[i].data_ptr(), m, X2, Y2, pred_ptr=pred[i].data_ptr(), stream=stream)
    n_class = engine.arch.n_class
    lab = torch.empty(X * Y * Z * T, dtype=torch.uint8, device=dev)
    counts = torch.empty((T, n_class), dtype=torch.int64, device=dev)
    _lib.check(_lib.lib.ukbb_fcn_unpack_labels(pred.data_ptr(), X, Y, Z, T, X2, Y2, x_pre, y_pre, n_class,
                                               lab.data_ptr(), counts.data_ptr(), stream), 'ukbb_fcn_unpack_labels')
    lab_h = lab.cpu().numpy().reshape((X, Y, Z, T), order='F')
    out = np.zeros(image.shape)                                 # float64, as deploy_network.py:92
    out[...] = lab_h
    if return_aux:
        return out, {'clip': (lo, hi), 'counts': counts.cpu().numpy()}
    return out


def clip_like_reference(frame, clip):
    """What a frame of ``image`` holds after the reference's in-place clip (image_utils.py:73-74)."""
    lo, hi = clip
    f = np.array(frame, copy=True)
    f[f < lo] = lo
    f[f > hi] = hi
    return f


def pick_ed_es_from_counts(counts, seq_name, seg4=False):
    """pipeline.pick_ed_es on the per-frame class counts (deploy_network.py:125-130)."""
    c1 = counts[:, 1]
    if seq_name == 'sa' or (seq_name == 'la_4ch' and seg4):
        return 0, int(np.argmin(c1))
    return 0, int(np.argmax(c1))
