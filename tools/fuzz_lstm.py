"""Randomised parity of the UNet-LSTM path (ukbb_fcn_forward_seq) against the numpy fp64 restatement oracle/fcn_oracle.py
unet_lstm: random weight seeds, N sequences of 9 frames, random map sizes.  GPU box; test infrastructure (uses oracle/).
python tools/fuzz_lstm.py [--cases 24]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def draw_case(rng):
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['UNet-LSTM_ao']
    wseed = int(rng.integers(1, 10 ** 6))
    n, h, w = int(rng.integers(1, 3)), 16 * int(rng.integers(1, 6)), 16 * int(rng.integers(1, 6))
    x = rng.standard_normal((n, 9, h, w, 1)).astype(np.float32)
    return {'arch': arch, 'wseed': wseed, 'params': synthetic_params(arch, wseed), 'x': x, 'n': n, 'h': h, 'w': w}


def grade_case(c):
    """ukbb_fcn_forward_seq vs the numpy fp64 restatement: (ok, relative logits error, label flips, flips away from a tie, pixels)."""
    from oracle import fcn_oracle as O
    from ukbb_cardiac_amd.engine import Engine
    arch = c['arch']
    eng = Engine(arch, c['params'])
    out = eng.run_seq(c['x'], want_logits=True)
    eng.close()
    ref = O.unet_lstm(c['x'], c['params'], arch.n_hidden, n_block=arch.n_block, dtype=np.float64)
    scale = float(np.abs(ref).max())
    err = float(np.abs(out['logits'] - ref).max())
    bad = out['pred'] != O.argmax_pred(ref)
    away = int((bad & (O.top2_margin(ref) > 1e-4)).sum())
    return err <= 1e-3 * scale and away == 0, err / scale, int(bad.sum()), away, int(bad.size)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=24)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--budget_s', type=float, default=600.0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    failed, worst, flips, px = 0, 0.0, 0, 0
    t0 = time.time()
    for case in range(args.cases):
        if time.time() - t0 > args.budget_s:
            print('time budget reached after %d cases' % case)
            break
        c = draw_case(rng)
        ok, rel, fl, away, npx = grade_case(c)
        failed += not ok
        worst = max(worst, rel); flips += fl; px += npx
        print('seed %6d  %dx9x%3dx%3d  rel logits err %.2e  label flips %d (away from a tie: %d)  %s' %
              (c['wseed'], c['n'], c['h'], c['w'], rel, fl, away, 'ok' if ok else 'FAIL'), flush=True)
    print('cases failed: %d; worst relative logits error %.2e; label flips %d of %d pixels; %.0f s' % (failed, worst, flips, px, time.time() - t0))
    sys.exit(1 if failed else 0)
