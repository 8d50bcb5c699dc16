"""Every convolution launch of the fp32 FCN plan against float64 on the engine's OWN stored input (r06): one kernel's error at a time.

tests/test_gpu_parity.py grades whole forwards (logits, labels, level outputs) against the oracle; here each stored map of the encoder is recomputed from
the map the engine stored in front of it -- numpy float64 through the reference's op (oracle/fcn_oracle.py conv2d_same, reference common/network.py:19-25) with
the BN fold of ukbb_fcn_create -- so a layer's deviation is its kernel's own: the direct stride-2 kernels (exact fp32 MFMA products, fp32 accumulation), the
fused first layer, Winograd F(2x2) and F(2x4) (fp32 transforms: ~1e-6 of the activation scale by the numpy model of DESIGN.md section 4).
Bound asserted per layer: max |engine - float64| <= 1e-5 x the layer's largest activation (measured: see the printed table, 1e-7 .. 2e-6)."""
import numpy as np
import pytest

from oracle import fcn_oracle as O

pytestmark = pytest.mark.gpu
BN_EPS = np.float32(1e-3)


def fold(p):
    sc = (p['gamma'].astype(np.float32) / np.sqrt(p['var'].astype(np.float32) + BN_EPS)).astype(np.float32)
    b = (p['beta'].astype(np.float32) - (p['mean'].astype(np.float32) * sc).astype(np.float32)).astype(np.float32)
    return (p['kernel'].astype(np.float32) * sc[None, None, None, :]).astype(np.float32), b


def layer(x, p, stride):
    w, b = fold(p)
    return np.maximum(O.conv2d_same(np.asarray(x, np.float64), w.astype(np.float64), stride) + b.astype(np.float64), 0.0)


@pytest.mark.parametrize('model,shape', [('FCN_sa', (2, 192, 208)), ('FCN_sa', (3, 80, 112)), ('FCN_la_2ch', (1, 176, 208))])
def test_each_encoder_launch_against_float64_on_its_own_input(model, shape):
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS[model]
    params = synthetic_params(arch, 1234)
    n, H, W = shape
    img = cine_phantom(n, H, W, seed=17)
    with Engine(arch, params) as eng:
        eng.run(img)
        cfgs = dict(zip(eng.kernel_names(), eng.kernel_configs()))
        stored = {}
        for l in range(arch.n_level):
            for i in range(arch.n_block[l]):
                name = 'conv%d' % l if i == arch.n_block[l] - 1 else 'conv%d_%d' % (l, i)
                try:
                    stored['conv%d_%d' % (l, i)] = eng.activation(name).reshape(n, -(-H >> l), -(-W >> l), -1).astype(np.float64)
                except Exception:
                    assert (l, i) == (0, 0)                              # conv0_0 lives inside conv0_1's launch
    report = {}
    x = None
    for l in range(arch.n_level):
        for i in range(arch.n_block[l]):
            nm = 'conv%d_%d' % (l, i)
            stride = 2 if (l > 0 and i == 0) else 1
            if nm not in stored:                                         # fused first layer: conv0_0 in fp32 from the image, then conv0_1
                x = layer(img.astype(np.float64), params[nm], 1)
                continue
            ex = layer(img.astype(np.float64) if x is None else x, params[nm], stride)
            sc = float(np.abs(ex).max())
            err = float(np.abs(stored[nm] - ex).max()) / sc
            report[nm] = err
            assert err <= 1e-5, (nm, err, cfgs)
            x = stored[nm]
    print('%s %s: per-launch max error / layer scale: %s' % (model, shape, {k: float('%.2g' % v) for k, v in report.items()}))
