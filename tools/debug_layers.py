"""Per-layer comparison of the HIP engine against the numpy oracle (debugging
aid; run on the GPU box):  python tools/debug_layers.py [model] [N H W]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fcn_oracle as O                                   # noqa: E402
from ukbb_cardiac_amd.arch import MODELS, KIND_FCN                   # noqa: E402
from ukbb_cardiac_amd.engine import Engine                           # noqa: E402
from ukbb_cardiac_amd.phantom import cine_phantom                    # noqa: E402
from ukbb_cardiac_amd.weights import synthetic_params                # noqa: E402

if __name__ == '__main__':
    model = sys.argv[1] if len(sys.argv) > 1 else 'FCN_sa'
    n, h, w = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (2, 32, 48)
    arch = MODELS[model]
    params = synthetic_params(arch, 1234)
    img = cine_phantom(n, h, w, seed=5)
    if arch.kind != KIND_FCN:
        img = ((img - 0.3) / 0.25).astype(np.float32)
    eng = Engine(arch, params)
    out = eng.run(img, want_logits=True)
    print('kernels:', eng.kernel_names())
    if arch.kind == KIND_FCN:
        ref, net = O.build_FCN(img, params, arch.n_class, dtype=np.float64, return_net=True)
        names = [('conv%d' % l, 'conv%d' % l) for l in range(5)] + [('g%d' % l, 'g%d' % l) for l in range(1, 5)]
        p0 = params['out0']
        scale = p0['gamma'].astype(np.float64) / np.sqrt(p0['var'].astype(np.float64) + 1e-3)
        for l in range(1, 5):
            net['g%d' % l] = net['conv%d_same_dim' % l] @ (p0['kernel'][0, 0, 32 * l:32 * (l + 1), :].astype(np.float64) * scale)
    else:
        ref, net = O.UNet(img, params, arch.n_class, n_block=arch.n_block, dtype=np.float64, return_net=True)
        names = [('conv%d' % l, 'conv%d' % l) for l in range(5)] + [('up%d' % l, 'conv%d_up' % l) for l in range(3, -1, -1)]
    for dev_name, ora_name in names:
        try:
            a = eng.activation(dev_name).reshape(net[ora_name].shape)
        except Exception as e:
            print('%-8s unavailable: %s' % (dev_name, e))
            continue
        r = net[ora_name]
        err = np.abs(a - r)
        print('%-8s shape %-20s max|ref| %.4f  max err %.3e  mean err %.3e  worst at %s' % (
            dev_name, r.shape, np.abs(r).max(), err.max(), err.mean(), np.unravel_index(err.argmax(), err.shape)))
    err = np.abs(out['logits'] - ref)
    print('logits   max|ref| %.4f max err %.3e rel %.3e' % (np.abs(ref).max(), err.max(), err.max() / np.abs(ref).max()))
    pred_ref = O.argmax_pred(ref)
    print('pred mismatches: %d of %d' % (int((out['pred'] != pred_ref).sum()), pred_ref.size))
    p_ref = O.softmax(ref)
    print('prob max err %.3e' % np.abs(out['prob'] - p_ref).max())
