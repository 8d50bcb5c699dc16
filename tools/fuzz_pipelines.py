"""Randomised equality sweep of the device-side subject paths against their numpy mirrors driven by the SAME engine (GPU box):
short-axis sequence (device_pipeline.segment_sequence_device and subject_pipeline.SubjectPipeline vs pipeline.segment_sequence),
aortic UNet and UNet-LSTM sequences (device z-score / pack / windows / argmax vs pipeline.aortic_*), over random volume shapes
(odd sizes, Z = 1, T = 1, short cines, time steps).  Everything must be bit-identical.   python tools/fuzz_pipelines.py [--cases 40]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=40)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    from ukbb_cardiac_amd import device_pipeline as dp, pipeline
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.subject_pipeline import SubjectPipeline
    from ukbb_cardiac_amd.weights import synthetic_params
    rng = np.random.default_rng(args.seed)
    engs = {k: Engine(MODELS[k], synthetic_params(MODELS[k], 1234)) for k in ('FCN_sa', 'UNet_ao', 'UNet-LSTM_ao')}
    bad = 0
    t0 = time.time()

    def volume(shape, integer):
        v = 300.0 * rng.gamma(2.0, 1.0, size=shape)
        if rng.random() < 0.3:
            v[rng.random(shape) < 0.2] = 0.0                           # ties at the bottom of the histogram
        v = np.round(v) if integer else v
        order = 'F' if rng.random() < 0.7 else 'C'
        return np.asarray(v.astype(np.float32), order=order)

    for case in range(args.cases):
        which = ['sa', 'sa_pipe', 'ao_unet', 'ao_lstm'][case % 4]
        if which in ('sa', 'sa_pipe'):
            shape = (int(rng.integers(17, 230)), int(rng.integers(17, 230)), int(rng.integers(1, 5)), int(rng.integers(1, 9)))
            vol = volume(shape, rng.random() < 0.5)
            eng = engs['FCN_sa']
            want = pipeline.segment_sequence(vol.copy(), lambda b: {'pred': eng.run(b, want_prob=False)['pred']}, 7)
            if which == 'sa':
                got, aux = dp.segment_sequence_device(vol, eng, batch_slices=int(rng.integers(1, 20)), return_aux=True)
                ok = np.array_equal(got, want) and np.array_equal(aux['counts'][:, 1], (want == 1).sum(axis=(0, 1, 2)))
            else:
                pipe = SubjectPipeline(eng, shape, batch_slices=int(rng.integers(1, 20)))
                res = list(pipe.run([np.asfortranarray(vol), np.asfortranarray(vol)]))
                ok = all(np.array_equal(r.labels, want.astype(np.uint8)) for r in res)
                del pipe
        elif which == 'ao_unet':
            shape = (int(rng.integers(17, 257)), int(rng.integers(17, 257)), int(rng.integers(1, 3)), int(rng.integers(1, 9)))
            vol = volume(shape, True)
            eng = engs['UNet_ao']
            prob = pipeline.aortic_prob_sequence(vol, lambda b: eng.run(b), batch_slices=5)
            got = dp.aortic_unet_sequence_device(vol, eng, batch_slices=int(rng.integers(1, 9)))
            ok = np.array_equal(got, np.argmax(prob, -1).astype(np.int32))
        else:
            T = int(rng.integers(4, 16))
            shape = (int(rng.integers(17, 257)), int(rng.integers(17, 257)), int(rng.integers(1, 3)), T)
            ts = int(rng.integers(1, 4))
            vol = volume(shape, True)
            eng = engs['UNet-LSTM_ao']
            prob = pipeline.aortic_lstm_prob_sequence(vol, lambda f, R, r, t_=1: eng.run_cine(f, R, r, t_)[0], time_step=ts)
            with np.errstate(invalid='ignore'):
                want = np.argmax(prob, -1).astype(np.int32)
            got, aux = dp.aortic_lstm_sequence_device(vol, eng, time_step=ts, return_aux=True)
            ok = np.array_equal(got, want) and np.array_equal(aux['prob'], prob, equal_nan=True)
        bad += not ok
        print('%-8s %-22s %s' % (which, shape, 'ok' if ok else 'MISMATCH'), flush=True)
    print('cases: %d, mismatches: %d, %.0f s' % (args.cases, bad, time.time() - t0))
    for e in engs.values():
        e.close()
    sys.exit(1 if bad else 0)
