"""Per-subject split of the aortic UNet-LSTM sequence path (deploy_network_ao.py:92-189) on one GPU: host pre/post-processing
around Engine.run_cine against device_pipeline.aortic_lstm_sequence_device.  Prints milliseconds per 100-frame cine."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

from ukbb_cardiac_amd import pipeline
from ukbb_cardiac_amd.arch import MODELS
from ukbb_cardiac_amd.device_pipeline import aortic_lstm_sequence_device
from ukbb_cardiac_amd.engine import Engine
from ukbb_cardiac_amd.image_utils import normalise_intensity
from ukbb_cardiac_amd.weights import synthetic_params


def timed(f, n=3):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    return (time.perf_counter() - t0) / n * 1e3, r


def main():
    arch = MODELS['UNet-LSTM_ao']
    eng = Engine(arch, synthetic_params(arch, 1234))
    rng = np.random.default_rng(0)
    vol = np.asfortranarray(np.round(100 * rng.gamma(2.0, 1.0, size=(240, 196, 1, 100))).astype(np.float32))
    tn, norm = timed(lambda: normalise_intensity(vol, 10.0))
    print('normalise_intensity (host): %.1f ms, dtype %s' % (tn, norm.dtype))
    tp, _ = timed(lambda: np.percentile(vol, 10.0))
    print('  of which percentile %.1f ms' % tp)
    cine = lambda fr, R, r: eng.run_cine(fr, R, r)[0]
    tt, prob = timed(lambda: pipeline.aortic_lstm_prob_sequence(vol, cine), 2)
    print('aortic_lstm_prob_sequence total (host norm + pad + run_cine + crop): %.1f ms' % tt)
    ta, pred = timed(lambda: np.argmax(prob, axis=-1).astype(np.int32))
    print('host argmax: %.1f ms' % ta)
    print('HOST PATH per subject: %.1f ms' % (tt + ta))
    frames = np.zeros((100, 256, 256), np.float32)
    tc, _ = timed(lambda: eng.run_cine(frames), 3)
    print('run_cine alone (incl. H2D 26 MB + D2H 78 MB prob + pred): %.1f ms' % tc)
    td, pred_d = timed(lambda: aortic_lstm_sequence_device(vol, eng), 5)
    print('DEVICE PATH per subject (aortic_lstm_sequence_device): %.1f ms, identical labels: %s' % (td, np.array_equal(pred, pred_d)))
    import torch
    from ukbb_cardiac_amd.device_pipeline import device_zscore_stats
    v = torch.from_numpy(vol).cuda()
    ts, _ = timed(lambda: device_zscore_stats(v, 10.0), 5)
    print('  of which percentile + ROI compaction + mean/std on the device: %.1f ms' % ts)


if __name__ == '__main__':
    main()
