"""bf16 ConvLSTM (kernels_ws.hip, ws_main LS) against the fp32-Winograd-on-bf16-storage form it replaces and against itself:
for several map sizes / cine lengths: (a) two runs of the direct-conv form are bit-identical (no lost store, no race), (b) its probabilities agree with the
Winograd form (UKBB_LSTM_BF16_WINOGRAD=1, same storage rounding, fp32 weights) to bf16-weight precision, labels >= 99 % equal.  GPU box.
    python tools/check_lstm_bf16.py"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(shapes, out):
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['UNet-LSTM_ao']
    res = {}
    with Engine(arch, synthetic_params(arch, 1234)) as eng:
        eng.set_precision('bf16')
        for (F, H, W, ts) in shapes:
            frames = ((cine_phantom(F, H, W, seed=F + H)[..., 0] - 0.3) / 0.25).astype(np.float32)
            p1, l1 = eng.run_cine(frames, time_step=ts)
            p2, l2 = eng.run_cine(frames, time_step=ts)
            res['%d_%d_%d_%d' % (F, H, W, ts)] = p1
            res['%d_%d_%d_%d_same' % (F, H, W, ts)] = np.array([np.array_equal(p1, p2, equal_nan=True) and np.array_equal(l1, l2)])
    np.savez(out, **res)


if __name__ == '__main__':
    shapes = [(13, 32, 48, 1), (20, 64, 96, 1), (11, 48, 80, 2), (30, 128, 128, 1), (100, 256, 256, 1), (9, 16, 16, 1), (25, 80, 272, 3)]
    if len(sys.argv) > 1:
        run(shapes, sys.argv[1])
        sys.exit(0)
    import tempfile
    d = tempfile.mkdtemp()
    outs = {}
    for tag, env in (('direct', {}), ('winograd', {'UKBB_LSTM_BF16_WINOGRAD': '1'}), ('unhoisted', {'UKBB_LSTM_BF16_UNHOIST': '1'})):
        e = dict(os.environ, **env)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), os.path.join(d, tag + '.npz')], env=e)
        outs[tag] = np.load(os.path.join(d, tag + '.npz'))
    bad = 0
    for (F, H, W, ts) in shapes:
        k = '%d_%d_%d_%d' % (F, H, W, ts)
        a, b = outs['direct'][k], outs['winograd'][k]
        same = bool(outs['direct'][k + '_same'][0])
        ok = np.isfinite(a) | ~np.isfinite(b)
        dmax = float(np.nanmax(np.abs(a - b)))
        agree = float((np.argmax(np.nan_to_num(a), -1) == np.argmax(np.nan_to_num(b), -1)).mean())
        # r06 experiment (UKBB_LSTM_BF16_UNHOIST=1): steps that re-multiply x (fp32 gates) against the default, which adds a gx rounded to bf16
        c = outs['unhoisted'][k]
        same_h = bool(outs['unhoisted'][k + '_same'][0])
        dmax_h = float(np.nanmax(np.abs(a - c)))
        agree_h = float((np.argmax(np.nan_to_num(a), -1) == np.argmax(np.nan_to_num(c), -1)).mean())
        fine = same and same_h and ok.all() and dmax < 0.25 and agree > 0.99 and dmax_h < 0.25 and agree_h > 0.99
        bad += not fine
        print('%3d frames %3dx%3d time_step %d: two runs identical %s; vs Winograd form: max |dprob| %.4f, labels equal %.4f; vs the un-hoisted form (UKBB_LSTM_BF16_UNHOIST=1): max |dprob| %.4f, '
              'labels equal %.4f  %s' % (F, H, W, ts, same and same_h, dmax, agree, dmax_h, agree_h, 'ok' if fine else 'FAIL'))
    print('FAIL' if bad else 'OK')
    sys.exit(1 if bad else 0)
