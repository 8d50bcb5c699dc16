"""Host side of the deployment loop, independent of files and of the device.

Each function takes ``forward(batch) -> dict`` -- the stand-in for the
reference's ``sess.run`` (``Engine.run`` in production; a stub in the CPU tests)
-- and reproduces the array handling of ``common/deploy_network.py:86-131,
171-200`` and ``common/deploy_network_ao.py:92-128,188-189``.

Difference from the reference, by design: the reference issues one
``sess.run`` per time frame with batch = Z slices (T = 50 small launches per
subject); slices are independent, so here all Z*T slices of a subject are
flattened into batches of ``batch_slices`` (default 128) to keep the GPU full.
"""
import math

import numpy as np

from .image_utils import normalise_intensity, rescale_intensity


def pad_amounts(X, Y, multiple=16):
    """Centred zero padding up to a multiple of 16 (deploy_network.py:97-99)."""
    X2 = int(math.ceil(X / float(multiple))) * multiple
    Y2 = int(math.ceil(Y / float(multiple))) * multiple
    x_pre, y_pre = (X2 - X) // 2, (Y2 - Y) // 2
    return X2, Y2, x_pre, (X2 - X) - x_pre, y_pre, (Y2 - Y) - y_pre


def pad_amounts_fixed(X, Y, size=256):
    """Aortic images are padded to a fixed 256 x 256 (deploy_network_ao.py:105-107).
    The reference lets np.pad raise on larger inputs; say why instead."""
    if X > size or Y > size:
        raise ValueError('aortic image %dx%d exceeds the fixed %dx%d network input '
                         '(common/deploy_network_ao.py:105 pads to 256, it cannot crop)' % (X, Y, size, size))
    x_pre, y_pre = (size - X) // 2, (size - Y) // 2
    return size, size, x_pre, (size - X) - x_pre, y_pre, (size - Y) - y_pre


def _run_slices(slices, forward, batch_slices, want):
    """slices [N,H,W] float32 -> concatenated outputs of forward over chunks."""
    outs = {k: [] for k in want}
    n = slices.shape[0]
    for i in range(0, n, batch_slices):
        res = forward(np.ascontiguousarray(slices[i:i + batch_slices, :, :, None]))
        for k in want:
            outs[k].append(res[k])
    return {k: np.concatenate(v, axis=0) if len(v) > 1 else v[0] for k, v in outs.items()}


def segment_sequence(image, forward, batch_slices=128, numpy1_casting=False):
    """(X,Y,Z,T) volume -> float64 label volume of the same shape
    (deploy_network.py:86-116; the float64 dtype is the reference's, :92).
    ``image`` is clipped IN PLACE by rescale_intensity, as in the reference."""
    if image.ndim != 4:
        raise ValueError('expected a 4-D (X,Y,Z,T) sequence, got shape %s' % (image.shape,))
    X, Y, Z, T = image.shape
    scaled = rescale_intensity(image, (1, 99), numpy1_casting)
    X2, Y2, x_pre, x_post, y_pre, y_post = pad_amounts(X, Y)
    padded = np.pad(scaled, ((x_pre, x_post), (y_pre, y_post), (0, 0), (0, 0)), 'constant')
    slices = np.transpose(padded, (3, 2, 0, 1)).reshape(T * Z, X2, Y2).astype(np.float32)
    lab = _run_slices(slices, forward, batch_slices, ('pred',))['pred']
    lab = lab.reshape(T, Z, X2, Y2).transpose(2, 3, 1, 0)[x_pre:x_pre + X, y_pre:y_pre + Y]
    pred = np.zeros(image.shape)
    pred[...] = lab
    return pred


def segment_frame(image, forward, batch_slices=128, numpy1_casting=False):
    """(X,Y[,Z]) ED or ES frame -> int32 label volume (deploy_network.py:171-200)."""
    if image.ndim == 2:
        image = np.expand_dims(image, axis=2)
    X, Y = image.shape[:2]
    scaled = rescale_intensity(image, (1, 99), numpy1_casting)
    X2, Y2, x_pre, x_post, y_pre, y_post = pad_amounts(X, Y)
    padded = np.pad(scaled, ((x_pre, x_post), (y_pre, y_post), (0, 0)), 'constant')
    slices = np.transpose(padded, (2, 0, 1)).astype(np.float32)
    lab = _run_slices(slices, forward, batch_slices, ('pred',))['pred']
    return np.transpose(lab, (1, 2, 0))[x_pre:x_pre + X, y_pre:y_pre + Y].astype(np.int32)


def pick_ed_es(pred, seq_name, seg4=False):
    """ED = frame 0; ES = frame of minimal (sa, la_4ch --seg4) or maximal
    (la_2ch, la_4ch) label-1 count (deploy_network.py:125-130)."""
    count = np.sum(pred == 1, axis=(0, 1, 2))
    if seq_name == 'sa' or (seq_name == 'la_4ch' and seg4):
        return 0, int(np.argmin(count))
    return 0, int(np.argmax(count))


def aortic_prob_sequence(image, forward, z_score=True, batch_slices=128, n_class=3):
    """(X,Y,Z,T) aortic cine -> float32 probabilities (X,Y,Z,T,n_class), 'UNet'
    branch of deploy_network_ao.py:92-128."""
    X, Y, Z, T = image.shape
    norm = normalise_intensity(image, 10.0) if z_score else rescale_intensity(image, (1.0, 99.0))
    X2, Y2, x_pre, x_post, y_pre, y_post = pad_amounts_fixed(X, Y)
    padded = np.pad(norm, ((x_pre, x_post), (y_pre, y_post), (0, 0), (0, 0)), 'constant')
    slices = np.transpose(padded, (3, 2, 0, 1)).reshape(T * Z, X2, Y2).astype(np.float32)
    pr = _run_slices(slices, forward, batch_slices, ('prob',))['prob']
    pr = pr.reshape(T, Z, X2, Y2, n_class).transpose(2, 3, 1, 0, 4)[x_pre:x_pre + X, y_pre:y_pre + Y]
    prob = np.zeros((X, Y, Z, T, n_class), dtype=np.float32)
    prob[...] = pr
    return prob


def aortic_segment_frame(image, forward, z_score=True, batch_slices=128):
    """ED/ES mode of deploy_network_ao.py:222-258: pad to a multiple of 16 here
    (not 256), fetch pred directly."""
    X, Y = image.shape[:2]
    if image.ndim == 2:
        image = np.expand_dims(image, axis=2)
    norm = normalise_intensity(image, 10.0) if z_score else rescale_intensity(image, (1.0, 99.0))
    X2, Y2, x_pre, x_post, y_pre, y_post = pad_amounts(X, Y)
    padded = np.pad(norm, ((x_pre, x_post), (y_pre, y_post), (0, 0)), 'constant')
    slices = np.transpose(padded, (2, 0, 1)).astype(np.float32)
    lab = _run_slices(slices, forward, batch_slices, ('pred',))['pred']
    return np.transpose(lab, (1, 2, 0))[x_pre:x_pre + X, y_pre:y_pre + Y].astype(np.int32)


def aortic_lstm_prob_sequence(image, cine_forward, z_score=True, weight_R=5, weight_r=0.1, n_class=3, time_step=1):
    """'UNet-LSTM' branch of deploy_network_ao.py:92-107,129-183: (X,Y,Z,T) aortic cine -> float32 probabilities
    (X,Y,Z,T,n_class).  ``cine_forward(frames[F,256,256], weight_R, weight_r[, time_step]) -> prob[F,256,256,C]``
    performs the circular 9-frame windows centred on frames range(0, T, time_step) (:147) and their weighted tiling
    for one slice position (``Engine.run_cine``: the U-Net features of
    a frame are computed once instead of once per window, results identical)."""
    X, Y, Z, T = image.shape
    norm = normalise_intensity(image, 10.0) if z_score else rescale_intensity(image, (1.0, 99.0))
    X2, Y2, x_pre, x_post, y_pre, y_post = pad_amounts_fixed(X, Y)
    padded = np.pad(norm, ((x_pre, x_post), (y_pre, y_post), (0, 0), (0, 0)), 'constant')
    prob = np.zeros((X, Y, Z, T, n_class), dtype=np.float32)
    for z in range(Z):                                   # the reference batches Z inside sess.run; slices are independent
        frames = np.transpose(padded[:, :, z, :], (2, 0, 1)).astype(np.float32)
        pr = cine_forward(frames, weight_R, weight_r) if time_step == 1 else cine_forward(frames, weight_R, weight_r, time_step)
        prob[:, :, z] = np.transpose(pr, (1, 2, 0, 3))[x_pre:x_pre + X, y_pre:y_pre + Y]
    return prob
