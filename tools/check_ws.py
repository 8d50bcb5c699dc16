"""Weight-stationary bf16 conv tilings (kernels_ws.hip, ids 400-) against the tile-per-workgroup bf16 kernel they replace, layer by
layer on the aortic U-Net in UKBB_PREC_BF16 (GPU box):   python tools/check_ws.py N H W layer:cfg[,cfg...] ...
The override changes ONE layer, so everything upstream is bit-identical and the layer's own output shows the kernel's difference:
both kernels form the same bf16 products and differ only in the fp32 summation order, i.e. by a bf16 ulp on a few elements."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ukbb_cardiac_amd.arch import MODELS                              # noqa: E402
from ukbb_cardiac_amd.engine import Engine                            # noqa: E402
from ukbb_cardiac_amd.phantom import cine_phantom                     # noqa: E402
from ukbb_cardiac_amd.weights import synthetic_params                 # noqa: E402

LEVEL_NAME = {'conv0_1': 'conv0', 'conv1_1': 'conv1', 'conv2_1': 'conv2', 'conv3_1': 'conv3', 'conv4_1': 'conv4',
              'up3_1': 'up3', 'up2_1': 'up2', 'up1_1': 'up1', 'up0_1': 'up0'}


def run(arch, params, img, override=None):
    os.environ['UKBB_NO_FUSE_TAIL'] = '1'                # layer-by-layer plans on both sides: the fused stem / tail have tools of their own
    os.environ['UKBB_NO_FUSE_STEM'] = '1'
    if override:
        os.environ['UKBB_CONV_CFG'] = override
    else:
        os.environ.pop('UKBB_CONV_CFG', None)
    with Engine(arch, params) as eng:
        eng.set_precision('bf16')
        out = eng.run(img, want_logits=True)
        cfgs = dict(zip(eng.kernel_names(), eng.kernel_configs()))
        acts = {}
        for nm in eng.kernel_names():
            key = LEVEL_NAME.get(nm, nm)
            try:
                acts[nm] = eng.activation(key)
            except Exception:
                pass
    return out, cfgs, acts


if __name__ == '__main__':
    n, H, W = (int(v) for v in sys.argv[1:4])
    arch = MODELS['UNet_ao']
    params = synthetic_params(arch, 1234)
    img = ((cine_phantom(n, H, W, seed=5) - 0.3) / 0.25).astype(np.float32)
    base, bcfg, bact = run(arch, params, img)
    bad = 0
    for spec in sys.argv[4:]:
        layer, cfgs = spec.split(':')
        for cfg in cfgs.split(','):
            ov = layer.split('+')[1] if layer.startswith('conv0_0+') else layer.split('+')[0]     # fused launches: conv0_0+conv0_1, up0_1+logits
            out, ocfg, oact = run(arch, params, img, '%s:%s' % (ov, cfg))
            if ocfg.get(layer) != int(cfg):
                print('%-8s cfg %s: NOT TAKEN (ran %s)' % (layer, cfg, ocfg.get(layer)))
                bad += 1
                continue
            if layer.endswith('+logits'):                              # the layer's own output is never stored: compare the logits
                a, b = out['logits'], base['logits']
            else:
                a, b = oact[layer], bact[layer]
            d = np.abs(a - b)
            scale = float(np.abs(b).max())
            nz = int((d > 0).sum())
            # a bf16 ulp is 2^-8 of the value's binade: differences beyond ~1 % of the element (or of the map's scale for tiny ones) are bugs
            tol = np.maximum(np.abs(b) * 2.0 ** -6, scale * 2.0 ** -12)
            wrong = int((d > tol).sum())
            lab = float((out['pred'] != base['pred']).mean())
            print('%-8s cfg %s vs %d: %d of %d elements differ (%.4f %%), max |d| %.4g (scale %.3g), beyond 2 ulp: %d; labels differ %.4f %%' % (
                layer, cfg, bcfg[layer], nz, a.size, 100.0 * nz / a.size, float(d.max()), scale, wrong, 100 * lab))
            if wrong or not np.isfinite(a).all():
                bad += 1
    print('FAIL' if bad else 'OK')
    sys.exit(1 if bad else 0)
