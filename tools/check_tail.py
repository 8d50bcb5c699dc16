"""Fused tail of the bf16 U-Net (kernels_tail.hip: up0_0 -> up0_1 -> logits -> argmax in one launch) against the unfused plan
(UKBB_NO_FUSE_TAIL=1), GPU box:   python tools/check_tail.py N H W [N H W ...]
Both plans form the same bf16 products and round the two intermediate maps to bf16; they differ in fp32 summation order, i.e. by a
bf16 ulp on a few intermediate values: logits agree to ~1e-3 of their scale, labels differ only at near-ties."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ukbb_cardiac_amd.arch import MODELS                              # noqa: E402
from ukbb_cardiac_amd.engine import Engine                            # noqa: E402
from ukbb_cardiac_amd.phantom import cine_phantom                     # noqa: E402
from ukbb_cardiac_amd.weights import synthetic_params                 # noqa: E402


def run(arch, params, img, fused):
    if fused:
        os.environ.pop('UKBB_NO_FUSE_TAIL', None)
    else:
        os.environ['UKBB_NO_FUSE_TAIL'] = '1'
    with Engine(arch, params) as eng:
        eng.set_precision('bf16')
        out = eng.run(img, want_logits=True, want_prob=True)
        only = eng.run(img, want_prob=False)                         # the pred-only path of the kernels
        names = eng.kernel_names()
    assert np.array_equal(only['pred'], out['pred'])
    return out, names


if __name__ == '__main__':
    arch = MODELS['UNet_ao']
    params = synthetic_params(arch, 1234)
    vals = [int(v) for v in sys.argv[1:]] or [2, 64, 96]
    bad = 0
    for i in range(0, len(vals), 3):
        n, H, W = vals[i:i + 3]
        img = ((cine_phantom(n, H, W, seed=5) - 0.3) / 0.25).astype(np.float32)
        a, na = run(arch, params, img, True)
        b, nb = run(arch, params, img, False)
        assert 'up0_0+up0_1+logits' in na and 'up0_0+up0_1+logits' not in nb, (na, nb)
        scale = float(np.abs(b['logits']).max())
        d = float(np.abs(a['logits'] - b['logits']).max())
        dp = float(np.abs(a['prob'] - b['prob']).max())
        lab = float((a['pred'] != b['pred']).mean())
        srt = np.sort(b['logits'], axis=-1)
        margin = (srt[..., -1] - srt[..., -2])[a['pred'] != b['pred']]
        ok = d <= 4e-3 * scale and lab <= 2e-4 and (margin.size == 0 or float(margin.max()) <= 4e-3 * scale) and np.array_equal(np.argmax(a['prob'], -1), a['pred'])
        print('%dx%dx%d: %d launches (unfused %d); logits max |d| %.3g of scale %.3g (%.2e rel), prob max |d| %.2e, labels differ %.5f %% (largest top-2 margin there %.3g): %s' % (
            n, H, W, len(na), len(nb), d, scale, d / scale, dp, 100 * lab, float(margin.max()) if margin.size else 0.0, 'ok' if ok else 'FAIL'))
        bad += not ok
    print('FAIL' if bad else 'OK')
    sys.exit(1 if bad else 0)
