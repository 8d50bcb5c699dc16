"""Randomised parity sweep on the GPU box: random models, batch sizes and map sizes (multiples of 16, incl. 1x1-tile and
very elongated maps), random weight seeds and inputs; every logit and label of the HIP engine against oracle/fcn_oracle.c
(fp32) with numpy-fp64 arbitration of label disagreements, exactly as tests/test_gpu_parity.py grades.  Test infrastructure
(uses oracle/): prints one line per case and a summary.   python tools/fuzz_parity.py [--cases 60] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


NAMES = ['FCN_sa', 'FCN_la_2ch', 'FCN_la_4ch', 'FCN_la_4ch_seg4', 'UNet_ao']


def draw_case(rng, case):
    """One random (model, weights, batch, size, input) case; the draws are the tool's since r02 plus, for FCN models, one case in
    five on ``weights.threshold_params`` (label maps with the statistics of a trained model: a few compact regions)."""
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import cine_phantom, uniform_slices
    from ukbb_cardiac_amd.weights import synthetic_params, threshold_params
    name = NAMES[int(rng.integers(len(NAMES)))]
    arch = MODELS[name]
    wseed = int(rng.integers(1, 10 ** 6))
    kind = int(rng.integers(4))
    if kind == 0:
        h, w = 16 * int(rng.integers(1, 5)), 16 * int(rng.integers(1, 5))          # a few tiles only
    elif kind == 1:
        h, w = 16, 16 * int(rng.integers(4, 26))                                   # one tile row
        if rng.random() < 0.5:
            h, w = w, h
    else:
        h, w = 16 * int(rng.integers(3, 20)), 16 * int(rng.integers(3, 20))
    n = int(rng.integers(1, 4)) if h * w > 160 * 160 else int(rng.integers(1, 9))
    img = (uniform_slices(n, h, w, seed=case) if rng.random() < 0.5 else cine_phantom(n, h, w, seed=case)).astype(np.float32)
    thr = arch.kind == 0 and rng.random() < 0.2
    if thr:
        t = np.sort(rng.uniform(0.15, 0.85, size=arch.n_class - 1))
        params = threshold_params(arch, thresholds=list(t), slope=float(rng.uniform(10.0, 60.0)))
    else:
        params = synthetic_params(arch, wseed)
    return {'name': name, 'arch': arch, 'wseed': wseed, 'params': params, 'img': img, 'n': n, 'h': h, 'w': w, 'threshold_model': bool(thr)}


def grade_case(c):
    """HIP engine vs oracle/fcn_oracle.c on one case, numpy-fp64 arbitration of label disagreements (tests/test_gpu_parity.py's grade).
    Returns (ok, relative logits error, label flips, flips away from a numerical tie, pixels)."""
    from oracle import c_oracle, fcn_oracle as O
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.weights import pack_flat
    arch, params, img, n = c['arch'], c['params'], c['img'], c['n']
    eng = Engine(arch, params)
    out = eng.run(img, want_logits=True, want_prob=False)
    eng.close()
    lg, _, pd = c_oracle.forward(arch, pack_flat(arch, params), img)
    scale = float(np.abs(lg).max())
    err = float(np.abs(out['logits'] - lg).max())
    bad = out['pred'] != pd
    away = 0
    for i in np.nonzero(bad.reshape(n, -1).any(axis=1))[0]:
        ref64 = O.build_FCN(img[i:i + 1], params, arch.n_class, dtype=np.float64) if arch.kind == 0 else \
            O.UNet(img[i:i + 1], params, arch.n_class, n_block=arch.n_block, dtype=np.float64)
        away += int((bad[i] & (O.top2_margin(ref64)[0] > 1e-4)).sum())
    return err <= 1e-3 * scale and away == 0, err / scale, int(bad.sum()), away, int(bad.size)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=60)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--budget_s', type=float, default=600.0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t_start = time.time()
    worst, flips_total, px_total, away_total, failed = 0.0, 0, 0, 0, 0
    for case in range(args.cases):
        if time.time() - t_start > args.budget_s:
            print('time budget reached after %d cases' % case)
            break
        c = draw_case(rng, case)
        ok, rel, flips, away, px = grade_case(c)
        failed += not ok
        worst = max(worst, rel)
        flips_total += flips; px_total += px; away_total += away
        print('%-16s seed %6d%s  %dx%3dx%3d  rel logits err %.2e  label flips %d (away from a tie: %d)  %s' %
              (c['name'], c['wseed'], ' thr' if c['threshold_model'] else '    ', c['n'], c['h'], c['w'], rel, flips, away, 'ok' if ok else 'FAIL'), flush=True)
    print('cases failed: %d; worst relative logits error %.2e; label flips %d of %d pixels, %d away from a numerical tie; %.0f s' %
          (failed, worst, flips_total, px_total, away_total, time.time() - t_start))
    sys.exit(1 if failed else 0)
