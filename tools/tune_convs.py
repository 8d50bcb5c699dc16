"""Measure every compiled conv tiling on every conv layer of a model at a given
batch shape (GPU box):  python tools/tune_convs.py [model] [N H W]
Prints, per layer, the tilings sorted by measured time; the winners are baked
into engine.cpp's preference table by hand."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ukbb_cardiac_amd import _lib                                     # noqa: E402
from ukbb_cardiac_amd.arch import MODELS                              # noqa: E402
from ukbb_cardiac_amd.engine import Engine                            # noqa: E402
from ukbb_cardiac_amd.weights import synthetic_params                 # noqa: E402

if __name__ == '__main__':
    import torch
    model = sys.argv[1] if len(sys.argv) > 1 else 'FCN_sa'
    n, h, w = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (64, 192, 208)
    arch = MODELS[model]
    params = synthetic_params(arch, 1234)
    x = torch.rand((n, h, w, 1), device='cuda')
    pred = torch.empty((n, h, w), dtype=torch.int32, device='cuda')

    def measure(env):
        os.environ['UKBB_CONV_CFG'] = env
        eng = Engine(arch, params)
        for _ in range(2):
            eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
        eng.set_timing(True)
        for _ in range(5):
            eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
        ms, cnt = eng.kernel_times()
        res = (eng.kernel_names(), eng.kernel_configs(), [m / c for m, c in zip(ms, cnt)], eng.kernel_macs())
        eng.close()
        return res

    names, cfgs, base, macs = measure('')
    ids = [i for i in range(400) if _lib.lib.ukbb_fcn_conv_config_name(i)]
    best = {}
    for li, nm in enumerate(names):
        if cfgs[li] < 0:
            continue
        rows = []
        for cid in ids:
            nm2, cf2, t2, _ = measure('%s:%d' % (nm.split('+')[-1], cid))   # fused kernels are keyed by their last layer
            if cf2[li] != cid:
                continue                      # tiling not valid for this layer
            rows.append((t2[li], cid))
        rows.sort()
        if not rows:
            continue
        best[nm.split('+')[-1]] = rows[0][1]
        peak = 157.3e12
        print('%-10s default cfg %2d %7.1f us | ' % (nm, cfgs[li], base[li] * 1e3) + '  '.join(
            '%d:%.1f(%.0f%%)' % (cid, t * 1e3, 100 * 2 * macs[li] / (t * 1e-3) / peak) for t, cid in rows), flush=True)
    print('UKBB_CONV_CFG=' + ','.join('%s:%d' % kv for kv in best.items()))
