"""Winograd F(2x4,3x3) (kernels_wino24.hip, tilings 304 / 305) against F(2x2,3x3) (300) and against each other, layer by layer, fp32, on the
FCN and the aortic U-Net (GPU box):   python tools/check_wino24.py
The override changes ONE layer, so everything upstream is bit-identical and the layer's own output shows the kernel's difference.
Expected: 304 and 305 agree bit for bit (same tile grid, transforms and K order, only the region shape differs); both differ from the
F(2x2) kernel by fp32 rounding -- the numpy model of the transform (r04_notes.md) puts the error of F(2x4) at 2.5x that of F(2x2), i.e. at
the level of a direct fp32 sum; bound used here: 2e-6 of the map's scale."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ukbb_cardiac_amd.arch import MODELS                              # noqa: E402
from ukbb_cardiac_amd.engine import Engine                            # noqa: E402
from ukbb_cardiac_amd.phantom import cine_phantom                     # noqa: E402
from ukbb_cardiac_amd.weights import synthetic_params                 # noqa: E402


def run(arch, params, img, override, key):
    if override:
        os.environ['UKBB_CONV_CFG'] = override
    else:
        os.environ.pop('UKBB_CONV_CFG', None)
    with Engine(arch, params) as eng:
        out = eng.run(img, want_logits=True)
        cfgs = dict(zip(eng.kernel_names(), eng.kernel_configs()))
        act = eng.activation(key)
    return out, cfgs, act


if __name__ == '__main__':
    bad = 0
    cases = [('FCN_sa', (3, 192, 208), [('conv2_1', 'conv2_1'), ('conv2_2', 'conv2'), ('conv3_1', 'conv3_1'), ('conv3_2', 'conv3')]),
             ('FCN_sa', (2, 48, 80), [('conv2_2', 'conv2'), ('conv3_2', 'conv3')]),                # ragged regions: 12 x 20 and 6 x 10 maps
             ('UNet_ao', (2, 256, 256), [('conv2_1', 'conv2'), ('conv3_1', 'conv3'), ('up2_0', 'up2_0'), ('up2_1', 'up2')]),
             ('FCN_sa', (3, 192, 208), [('conv1_1', 'conv1')]), ('FCN_sa', (2, 48, 80), [('conv1_1', 'conv1')]),
             ('UNet_ao', (2, 256, 256), [('conv1_1', 'conv1'), ('up1_0', 'up1_0'), ('up1_1', 'up1')]),
             # image pairs with seam regions (tiling 306, maps with Ho % 8 == 4): 12 x 13 maps, even and odd batches, 20 x 16 and 4 x 8 maps
             ('FCN_sa', (4, 192, 208), [('conv4_1', 'conv4_1'), ('conv4_2', 'conv4')]),
             ('FCN_sa', (3, 192, 208), [('conv4_2', 'conv4')]), ('FCN_sa', (1, 192, 208), [('conv4_1', 'conv4_1')]),
             ('FCN_sa', (5, 160, 256), [('conv3_2', 'conv3')]), ('FCN_sa', (3, 64, 128), [('conv4_2', 'conv4')])]
    for model, (n, H, W), layers in cases:
        arch = MODELS[model]
        params = synthetic_params(arch, 1234)
        img = cine_phantom(n, H, W, seed=7).astype(np.float32)
        if model == 'UNet_ao':
            img = (img - 0.3) / 0.25
        for layer, key in layers:
            try:
                ref, rcfg, ract = run(arch, params, img, '%s:%d' % (layer, 301 if layer in ('conv1_1', 'up1_0', 'up1_1') else 300), key)
            except Exception as e:
                print('%s %s: activation %s not available (%s)' % (model, layer, key, e))
                bad += 1
                continue
            res = {}
            pair = (H >> int(layer[4])) % 8 == 4                                     # the layer's map height: tiling 306 applies
            narrow = layer in ('conv1_1', 'up1_0', 'up1_1')                          # 32 output channels: tiling 307 only
            for cfg in ((307,) if narrow else (304, 305, 306) if pair else (304, 305)):
                out, ocfg, oact = run(arch, params, img, '%s:%d' % (layer, cfg), key)
                if ocfg.get(layer) != cfg:
                    print('%s %s cfg %d: NOT TAKEN (ran %s)' % (model, layer, cfg, ocfg.get(layer)))
                    res[cfg] = None
                    continue
                res[cfg] = oact
                scale = float(np.abs(ract).max())
                err = float(np.abs(oact - ract).max()) / scale
                lab = float((out['pred'] != ref['pred']).mean())
                ok = err <= 2e-6 and np.isfinite(oact).all()
                print('%-8s %dx%dx%d %-8s cfg %d vs F(2x2): max |d| %.2e of the scale, labels differ %.5f %%: %s' % (
                    model, n, H, W, layer, cfg, err, 100 * lab, 'ok' if ok else 'FAIL'))
                bad += 0 if ok else 1
            got = [c for c in (304, 305, 306) if res.get(c) is not None]
            if len(got) > 1:
                same = all(np.array_equal(res[got[0]], res[c]) for c in got[1:])
                print('%-8s %dx%dx%d %-8s %s agree bit for bit: %s' % (model, n, H, W, layer, ' == '.join(map(str, got)), same))
                bad += 0 if same else 1
    print('FAIL' if bad else 'OK')
    sys.exit(1 if bad else 0)
