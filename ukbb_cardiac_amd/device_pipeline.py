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
from .pipeline import pad_amounts, pad_amounts_fixed


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


# ---- aortic cine: z-score + UNet-LSTM on the device (common/deploy_network_ao.py:92-107,129-189) ------------------

def scalar_percentile_ranks(n, q):
    """0-based ranks (k, k+1) and the float32 fractional part np.percentile uses when q is a Python SCALAR and the data
    float32 (image_utils.py:62, ``np.percentile(image, thres_roi)``): numpy 2 then keeps everything in the data's
    dtype -- quantile = q / float32(100), virtual index (n - 1) * quantile in float32, gamma = float32(float64(virtual) - k)."""
    quantile = np.true_divide(q, np.float32(100))
    virtual = np.asanyarray((n - 1) * quantile)
    if virtual >= n - 1:
        return n - 1, n - 1, 0.0
    k = int(np.floor(virtual))
    gamma = np.asanyarray(np.asanyarray(virtual - np.asanyarray(k, dtype=np.intp)), dtype=virtual.dtype)
    return k, k + 1, float(gamma)


def device_scalar_percentile(vol_t, q, stream=0):
    """Exact np.percentile(float32 volume, q) for a Python-scalar q, on the device."""
    n = vol_t.numel()
    k, k1, g = scalar_percentile_ranks(n, q)
    r = (C.c_uint64 * 2)(k, k1)
    out = np.empty(2, np.float32)
    _lib.check(_lib.lib.ukbb_fcn_select_kth(vol_t.data_ptr(), n, r, 2, _lib.f32ptr(out), stream), 'ukbb_fcn_select_kth')
    # numpy's own float32 lerp: a Python-float quantile on a float32 array stays float32, (2 - 1) * g = g exactly
    return np.quantile(out, g)


def device_zscore_stats(vol_t, thres_roi=10.0, stream=0):
    """(mu, sigma + eps, n_roi, val_l) of image_utils.normalise_intensity for a dense float32 (X,Y,Z,T) torch tensor on the
    GPU, bit-identical to numpy: the ROI is compacted in numpy's element order and summed along numpy's pairwise tree
    (``ukbb_fcn_roi_compact`` / ``ukbb_fcn_pairwise_sum``); the few scalar operations around the sums are numpy's own."""
    import torch
    X, Y, Z, T = vol_t.shape
    val_l = device_scalar_percentile(vol_t, thres_roi, stream)
    roi = torch.empty(vol_t.numel(), dtype=torch.float32, device=vol_t.device)
    sx, sy, sz, st = vol_t.stride()
    n_roi = C.c_uint64(0)
    _lib.check(_lib.lib.ukbb_fcn_roi_compact(vol_t.data_ptr(), X, Y, Z, T, sx, sy, sz, st, float(val_l), roi.data_ptr(),
                                             C.byref(n_roi), stream), 'ukbb_fcn_roi_compact')
    n = np.intp(n_roi.value)                                  # _count_reduce_items returns an intp scalar: the divisions below are float64
    s = C.c_float(0)
    _lib.check(_lib.lib.ukbb_fcn_pairwise_sum(roi.data_ptr(), int(n), 0, 0.0, C.byref(s), stream), 'ukbb_fcn_pairwise_sum')
    with np.errstate(all='ignore'):
        mu = np.float32(np.float32(s.value) / n)             # np.mean: ret.dtype.type(ret / rcount)
        _lib.check(_lib.lib.ukbb_fcn_pairwise_sum(roi.data_ptr(), int(n), 1, float(mu), C.byref(s), stream), 'ukbb_fcn_pairwise_sum')
        var = np.float32(np.float32(s.value) / n)            # np.var: sum((x - mean)^2) in float32, / n in float64, back to float32
        sigma = np.float32(np.sqrt(var))
    return mu, sigma + 1e-6, int(n), val_l                   # float32 + Python float stays float32 (image_utils.py:66-67)


_ZSCORE_OK = {}


def device_zscore_matches_numpy(engine, warn=None):
    """Once per process and device: does the device z-score reproduce THIS numpy?  ``device_zscore_stats`` mirrors numpy
    internals -- the 8192-element buffering and the 128-element / 8-accumulator pairwise leaves of ``np.add.reduce``, the float32
    handling of a scalar percentile -- that ``np.setbufsize`` or another numpy release can change without notice.  A
    26 k-voxel probe (more than three reduction buffers, ROI not a multiple of anything) is pushed through both; on any
    difference in (val_l, mu, sigma + eps) the callers keep the host path (deploy_network_ao.py) and say so."""
    key = engine.device
    if key not in _ZSCORE_OK:
        import torch
        ok, why = True, ''
        if np.getbufsize() != 8192:
            ok, why = False, 'np.getbufsize() = %d, the device reproduces the 8192-element default' % np.getbufsize()
        else:
            rng = np.random.default_rng(20261003)
            probe = (1000.0 * rng.gamma(2.0, 1.0, size=(37, 29, 1, 25))).astype(np.float32)
            val_l = np.percentile(probe, 10.0)
            roi = probe[probe >= val_l]
            want = (val_l, np.mean(roi), np.std(roi) + 1e-6)
            dev = torch.device('cuda', engine.device)
            vol = torch.from_numpy(np.asfortranarray(probe)).to(dev)
            mu, den, _, got_l = device_zscore_stats(vol, 10.0, torch.cuda.current_stream(dev).cuda_stream)
            got = (got_l, mu, den)
            if not all(np.float32(a) == np.float32(b) for a, b in zip(want, got)):
                ok, why = False, 'probe statistics differ: numpy %r, device %r (numpy %s)' % (want, got, np.__version__)
        _ZSCORE_OK[key] = (ok, why)
        if not ok and warn is not None:
            warn('  device z-score disabled, host pre-processing used instead: ' + why)
    return _ZSCORE_OK[key][0]


def aortic_lstm_sequence_device(image, engine, z_score=True, weight_R=5, weight_r=0.1, time_step=1, return_aux=False):
    """pipeline.aortic_lstm_prob_sequence + the argmax of deploy_network_ao.py:189 with the array work on the GPU:
    (X,Y,Z,T) float32 aortic cine -> int32 label volume (X,Y,Z,T).  Only the raw volume goes in and uint8 labels come
    back (the host path moves the padded float32 cine in and 3 float32 probability maps per voxel out, and spends more
    time in np.percentile / np.argmax than the network takes).  ``aux['prob']`` (X,Y,Z,T,C) on request; ``aux['counts']``
    = pixels of each class per frame (what eval_aortic_area.py:60-78 turns into areas)."""
    import torch
    if image.ndim != 4 or image.dtype != np.float32:
        raise TypeError('expected a 4-D float32 (X,Y,Z,T) cine; use pipeline.aortic_lstm_prob_sequence otherwise')
    if not z_score:
        raise ValueError('the device path implements the default --z_score pre-processing')
    X, Y, Z, T = image.shape
    dev = torch.device('cuda', engine.device)
    stream = torch.cuda.current_stream(dev).cuda_stream
    src = image if (image.flags.f_contiguous or image.flags.c_contiguous) else np.asfortranarray(image)
    vol = torch.from_numpy(src).to(dev)
    mu, den, n_roi, val_l = device_zscore_stats(vol, 10.0, stream)
    X2, Y2, x_pre, _, y_pre, _ = pad_amounts_fixed(X, Y)
    n_class = engine.arch.n_class
    batch = torch.empty((T, Z, X2, Y2), dtype=torch.float32, device=dev)
    sx, sy, sz, st = vol.stride()
    _lib.check(_lib.lib.ukbb_fcn_zscore_pack(vol.data_ptr(), X, Y, Z, T, sx, sy, sz, st, float(mu), float(den), X2, Y2, x_pre, y_pre,
                                             batch.data_ptr(), stream), 'ukbb_fcn_zscore_pack')
    prob = torch.empty((T, Z, X2, Y2, n_class), dtype=torch.float32, device=dev)
    pred = torch.empty((T, Z, X2, Y2), dtype=torch.int32, device=dev)
    for z in range(Z):                                       # slice positions are independent cines (usually Z = 1)
        fr = batch[:, z] if Z == 1 else batch[:, z].contiguous()
        pr = prob[:, z] if Z == 1 else torch.empty((T, X2, Y2, n_class), dtype=torch.float32, device=dev)
        pd = pred[:, z] if Z == 1 else torch.empty((T, X2, Y2), dtype=torch.int32, device=dev)
        engine.run_cine_device(fr.data_ptr(), T, X2, Y2, pr.data_ptr(), pd.data_ptr(), weight_R, weight_r, time_step, stream)
        if Z > 1:
            prob[:, z].copy_(pr)
            pred[:, z].copy_(pd)
    lab = torch.empty(X * Y * Z * T, dtype=torch.uint8, device=dev)
    counts = torch.empty((T, n_class), dtype=torch.int64, device=dev)
    _lib.check(_lib.lib.ukbb_fcn_unpack_labels(pred.data_ptr(), X, Y, Z, T, X2, Y2, x_pre, y_pre, n_class,
                                               lab.data_ptr(), counts.data_ptr(), stream), 'ukbb_fcn_unpack_labels')
    out = lab.cpu().numpy().reshape((X, Y, Z, T), order='F').astype(np.int32)
    if not return_aux:
        return out
    aux = {'mu': mu, 'den': den, 'n_roi': n_roi, 'val_l': val_l, 'counts': counts.cpu().numpy()}
    if return_aux != 'counts':                                 # 'counts': skip the 78 MB of probabilities
        p = prob[:, :, x_pre:x_pre + X, y_pre:y_pre + Y].permute(2, 3, 1, 0, 4)
        aux['prob'] = p.cpu().numpy()
    return out, aux


def aortic_unet_sequence_device(image, engine, batch_slices=128, return_aux=False):
    """pipeline.aortic_prob_sequence + the argmax of deploy_network_ao.py:189 for the frame-wise 'UNet' model with the array work
    on the GPU: device z-score, pack, batched forward (the engine's label map IS the lowest-index argmax of the float32
    probabilities it would return: ``softmax_argmax``, csrc/kernels.h), labels back as uint8.
    (X,Y,Z,T) float32 -> int32 labels (X,Y,Z,T)."""
    import torch
    if image.ndim != 4 or image.dtype != np.float32:
        raise TypeError('expected a 4-D float32 (X,Y,Z,T) cine; use pipeline.aortic_prob_sequence otherwise')
    X, Y, Z, T = image.shape
    dev = torch.device('cuda', engine.device)
    stream = torch.cuda.current_stream(dev).cuda_stream
    src = image if (image.flags.f_contiguous or image.flags.c_contiguous) else np.asfortranarray(image)
    vol = torch.from_numpy(src).to(dev)
    mu, den, n_roi, val_l = device_zscore_stats(vol, 10.0, stream)
    X2, Y2, x_pre, _, y_pre, _ = pad_amounts_fixed(X, Y)
    n = T * Z
    batch = torch.empty((n, X2, Y2), dtype=torch.float32, device=dev)
    sx, sy, sz, st = vol.stride()
    _lib.check(_lib.lib.ukbb_fcn_zscore_pack(vol.data_ptr(), X, Y, Z, T, sx, sy, sz, st, float(mu), float(den), X2, Y2, x_pre, y_pre,
                                             batch.data_ptr(), stream), 'ukbb_fcn_zscore_pack')
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
    out = lab.cpu().numpy().reshape((X, Y, Z, T), order='F').astype(np.int32)
    if return_aux:
        return out, {'mu': mu, 'den': den, 'n_roi': n_roi, 'val_l': val_l, 'counts': counts.cpu().numpy()}
    return out
