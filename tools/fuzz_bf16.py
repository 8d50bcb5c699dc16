"""Randomised sweep of UKBB_PREC_BF16 on the aortic U-Net (GPU box): random batch sizes, map sizes (multiples of 16, incl. maps
smaller than a tile and very elongated ones), weight seeds and inputs; the bf16 path (bf16 activations in HBM, fused first layer
and logits) against the fp32 path of the same engine: logits deviation relative to the fp32 logits' scale, label disagreement,
Dice of the populated classes, pred == argmax(prob), no NaN.   python tools/fuzz_bf16.py [--cases 200] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--budget_s', type=float, default=300.0)
    args = ap.parse_args()
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.image_utils import np_categorical_dice
    from ukbb_cardiac_amd.phantom import cine_phantom, uniform_slices
    from ukbb_cardiac_amd.weights import synthetic_params
    rng = np.random.default_rng(args.seed)
    arch = MODELS['UNet_ao']
    t0 = time.time()
    worst_rel, worst_dis, worst_dice, failed, done = 0.0, 0.0, 1.0, 0, 0
    for case in range(args.cases):
        if time.time() - t0 > args.budget_s:
            break
        n = int(rng.choice([1, 2, 3, 5, 10, 16, 17, 24]))
        h, w = (int(16 * rng.integers(1, 20)), int(16 * rng.integers(1, 20)))
        if rng.random() < 0.1:
            h, w = rng.choice([(16, 16 * int(rng.integers(1, 40))), (16 * int(rng.integers(1, 40)), 16), (256, 256)])[0:2]
            h, w = int(h), int(w)
        seed = int(rng.integers(0, 1 << 20))
        img = uniform_slices(n, h, w, seed=seed)[..., 0] if rng.random() < 0.3 else cine_phantom(n, h, w, seed=seed)[..., 0]
        img = ((img - 0.3) / 0.25).astype(np.float32)
        with Engine(arch, synthetic_params(arch, seed)) as eng:
            f32 = eng.run(img, want_logits=True)
            eng.set_precision('bf16')
            b16 = eng.run(img, want_logits=True)
        rel = float(np.abs(b16['logits'] - f32['logits']).max() / np.abs(f32['logits']).max())
        dis = float((b16['pred'] != f32['pred']).mean())
        dices = [float(np_categorical_dice(b16['pred'], f32['pred'], k)) for k in (1, 2) if (f32['pred'] == k).sum() > 200]
        ok = bool(np.isfinite(b16['logits']).all()) and rel < 0.08 and dis < 0.08 and np.array_equal(np.argmax(b16['prob'], -1), b16['pred'])
        failed += not ok
        done += 1
        worst_rel, worst_dis = max(worst_rel, rel), max(worst_dis, dis)
        worst_dice = min([worst_dice] + dices)
        print('seed %7d  %2dx%3dx%3d  rel logits dev %.4f  label disagreement %.4f  dice %s  %s' % (
            seed, n, h, w, rel, dis, ' '.join('%.4f' % d for d in dices) or '-', 'ok' if ok else 'FAILED'), flush=True)
    print('cases %d, failed %d; worst relative logits deviation %.4f, worst label disagreement %.4f, worst Dice of a populated class %.4f; %.0f s' % (
        done, failed, worst_rel, worst_dis, worst_dice, time.time() - t0))
    sys.exit(1 if failed else 0)
