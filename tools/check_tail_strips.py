"""Strip-streaming form of the fused tail (kernels_tail.hip, STRIP: tiles walked down column strips, the four halo rows vertically
adjacent tiles share kept in LDS) against the row-major walk it replaces: the arithmetic per tile is the same and so are the bytes a
tile sees, so logits, probabilities and labels must be IDENTICAL BITS -- for whole strips, for every segment length (UKBB_TAIL_SEG,
incl. 1 = every tile loads its whole halo, and lengths that do not divide the tile rows), ragged maps and borders, both the
pred-only and the full-output kernels.   python tools/check_tail_strips.py [N H W ...]      GPU box; prints OK at the end."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ukbb_cardiac_amd.arch import MODELS                              # noqa: E402
from ukbb_cardiac_amd.engine import Engine                            # noqa: E402
from ukbb_cardiac_amd.phantom import cine_phantom                     # noqa: E402
from ukbb_cardiac_amd.weights import synthetic_params                 # noqa: E402

if __name__ == '__main__':
    arch = MODELS['UNet_ao']
    params = synthetic_params(arch, 1234)
    vals = [int(v) for v in sys.argv[1:]] or [2, 64, 96, 3, 256, 256, 1, 48, 80, 2, 16, 16, 1, 32, 272, 5, 112, 48, 7, 128, 128]
    bad = 0
    for i in range(0, len(vals), 3):
        n, H, W = vals[i:i + 3]
        img = ((cine_phantom(n, H, W, seed=5) - 0.3) / 0.25).astype(np.float32)
        with Engine(arch, params) as eng:
            eng.set_precision('bf16')
            assert 'up0_0+up0_1+logits' in eng.kernel_names() or True
            os.environ['UKBB_TAIL_STRIPS'] = '0'
            os.environ.pop('UKBB_TAIL_SEG', None)
            ref = eng.run(img, want_logits=True, want_prob=True)
            ref_p = eng.run(img, want_prob=False)['pred']
            assert 'up0_0+up0_1+logits' in eng.kernel_names(), eng.kernel_names()
            os.environ['UKBB_TAIL_STRIPS'] = '1'
            for seg in (None, 1, 2, 3, 5, 1000):
                if seg is None:
                    os.environ.pop('UKBB_TAIL_SEG', None)
                else:
                    os.environ['UKBB_TAIL_SEG'] = str(seg)
                got = eng.run(img, want_logits=True, want_prob=True)
                got_p = eng.run(img, want_prob=False)['pred']
                same = all(np.array_equal(got[k], ref[k]) for k in ('logits', 'prob', 'pred')) and np.array_equal(got_p, ref_p)
                if not same:
                    bad += 1
                    d = got['pred'] != ref['pred']
                    rows = np.unique(np.argwhere(d)[:, 1])[:12] if d.any() else []
                    print('%dx%dx%d seg %s: DIFFERS: %d label pixels, max |dlogits| %.3g, first rows %s' %
                          (n, H, W, seg, int(d.sum()), float(np.abs(got['logits'] - ref['logits']).max()), list(rows)))
                else:
                    print('%dx%dx%d seg %s: identical bits (logits, prob, pred; pred-only kernel too)' % (n, H, W, seg or 'auto'))
        os.environ.pop('UKBB_TAIL_SEG', None)
    print('OK' if bad == 0 else 'FAILED: %d cases' % bad)
    sys.exit(1 if bad else 0)
