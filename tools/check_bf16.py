"""UKBB_PREC_BF16 of the aortic U-Net (bf16 operands AND bf16 activations in HBM) against the fp32 path of the same engine:
per-layer activation error, logits error, Dice, label disagreement.   python tools/check_bf16.py [N] [H] [W]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ukbb_cardiac_amd.arch import MODELS                              # noqa: E402
from ukbb_cardiac_amd.engine import Engine                            # noqa: E402
from ukbb_cardiac_amd.image_utils import np_categorical_dice         # noqa: E402
from ukbb_cardiac_amd.phantom import cine_phantom                     # noqa: E402
from ukbb_cardiac_amd.weights import synthetic_params                 # noqa: E402

if __name__ == '__main__':
    n, H, W = (int(v) for v in (sys.argv[1:4] + ['2', '64', '96'][len(sys.argv) - 1:]))
    arch = MODELS['UNet_ao']
    img = ((cine_phantom(n, H, W, seed=5) - 0.3) / 0.25).astype(np.float32)
    with Engine(arch, synthetic_params(arch, 1234)) as eng:
        names = ['conv%d' % l for l in range(5)] + ['up%d' % l for l in (3, 2, 1, 0)]
        f32 = eng.run(img, want_logits=True)
        a32 = {k: eng.activation(k) for k in names}
        eng.set_precision('bf16')
        b16 = eng.run(img, want_logits=True)
        print('kernels:', list(zip(eng.kernel_names(), eng.kernel_configs())))
        for k in names:
            try:
                a = eng.activation(k)
            except Exception as e:                                 # 'up0' is not stored when the logits are fused into up0_1
                print('%-6s not available in this plan' % k)
                continue
            ref = a32[k]
            print('%-6s max|ref| %8.3f  max err %8.4f  rel %.4f  rms rel %.5f' % (
                k, np.abs(ref).max(), np.abs(a - ref).max(), np.abs(a - ref).max() / np.abs(ref).max(),
                np.sqrt(np.mean((a - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))))
    sc = np.abs(f32['logits']).max()
    print('logits: scale %.3f max err %.4f rel %.4f' % (sc, np.abs(b16['logits'] - f32['logits']).max(), np.abs(b16['logits'] - f32['logits']).max() / sc))
    print('Dice bf16 vs fp32: class1 %.4f class2 %.4f; label disagreement %.4f %%' % (
        np_categorical_dice(b16['pred'], f32['pred'], 1), np_categorical_dice(b16['pred'], f32['pred'], 2),
        100.0 * (b16['pred'] != f32['pred']).mean()))
