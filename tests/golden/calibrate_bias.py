"""One-off derivation of ukbb_cardiac_amd.weights.SYNTH_LOGITS_BIAS (test
infrastructure; uses the CPU oracle).  Run from the repo root:
    python tests/golden/calibrate_bias.py
It prints, per model, random_bias(seed 1234) - median(logits over a phantom)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import fcn_oracle as O                                    # noqa: E402
from ukbb_cardiac_amd import weights as Wt                            # noqa: E402
from ukbb_cardiac_amd.arch import MODELS, KIND_FCN                    # noqa: E402
from ukbb_cardiac_amd.phantom import cine_phantom                     # noqa: E402

if __name__ == '__main__':
    saved = dict(Wt.SYNTH_LOGITS_BIAS)
    Wt.SYNTH_LOGITS_BIAS.clear()                                      # get the raw random bias
    for name, arch in MODELS.items():
        params = Wt.synthetic_params(arch, 1234)
        if arch.kind == KIND_FCN:
            lg = O.build_FCN(cine_phantom(2, 192, 208, 0), params, arch.n_class, dtype=np.float32)
        else:
            img = (cine_phantom(1, 256, 256, 3) - 0.3) / 0.25
            lg = O.UNet(img, params, arch.n_class, n_block=arch.n_block, dtype=np.float32)
        med = np.median(lg, axis=(0, 1, 2))
        print(name, np.round(params['logits']['bias'] - med, 4).tolist(), 'frozen:', saved.get(name))
