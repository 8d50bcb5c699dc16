"""Fused stem of the bf16 U-Net (kernels_stem.hip: conv0_0 + conv0_1 in one launch, image rounded to bf16) against the r03 form
(UKBB_NO_FUSE_STEM=1: conv0_0 in fp32 inside conv0_1's staging) and against the fp32 path, GPU box:  python tools/check_stem.py N H W ..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ukbb_cardiac_amd.arch import MODELS                              # noqa: E402
from ukbb_cardiac_amd.engine import Engine                            # noqa: E402
from ukbb_cardiac_amd.image_utils import np_categorical_dice         # noqa: E402
from ukbb_cardiac_amd.phantom import cine_phantom                     # noqa: E402
from ukbb_cardiac_amd.weights import synthetic_params                 # noqa: E402


def run(arch, params, img, prec, fused):
    if fused:
        os.environ.pop('UKBB_NO_FUSE_STEM', None)
    else:
        os.environ['UKBB_NO_FUSE_STEM'] = '1'
    with Engine(arch, params) as eng:
        eng.set_precision(prec)
        out = eng.run(img, want_logits=True)
        return out, eng.activation('conv0'), eng.kernel_names()


if __name__ == '__main__':
    arch = MODELS['UNet_ao']
    params = synthetic_params(arch, 1234)
    vals = [int(v) for v in sys.argv[1:]] or [2, 64, 96]
    bad = 0
    for i in range(0, len(vals), 3):
        n, H, W = vals[i:i + 3]
        img = ((cine_phantom(n, H, W, seed=5) - 0.3) / 0.25).astype(np.float32)
        f32, c32, _ = run(arch, params, img, 'fp32', True)
        a, ca, na = run(arch, params, img, 'bf16', True)
        b, cb, nb = run(arch, params, img, 'bf16', False)
        scale = float(np.abs(c32).max())
        ea, eb = float(np.abs(ca - c32).max()) / scale, float(np.abs(cb - c32).max()) / scale
        ra, rb = float(np.sqrt(np.mean((ca - c32) ** 2)) / np.sqrt(np.mean(c32 ** 2))), float(np.sqrt(np.mean((cb - c32) ** 2)) / np.sqrt(np.mean(c32 ** 2)))
        da = [float(np_categorical_dice(a['pred'], f32['pred'], k)) for k in (1, 2)]
        db = [float(np_categorical_dice(b['pred'], f32['pred'], k)) for k in (1, 2)]
        ok = ea <= 2.5 * max(eb, 4e-3) and ra <= 2.5 * rb and min(da) >= min(db) - 0.01 and np.isfinite(ca).all()
        print('%dx%dx%d: conv0 vs fp32: fused stem max %.4f rms %.5f | r03 form max %.4f rms %.5f (of scale / rms); Dice vs fp32 fused %.4f %.4f | r03 form %.4f %.4f; launches %d vs %d: %s' % (
            n, H, W, ea, ra, eb, rb, da[0], da[1], db[0], db[1], len(na), len(nb), 'ok' if ok else 'FAIL'))
        bad += not ok
    print('FAIL' if bad else 'OK')
    sys.exit(1 if bad else 0)
