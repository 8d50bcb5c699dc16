"""UNet-LSTM graphs other than the reference's (9 steps, 3 classes): T in {1, 3, 5, 13}, n_class in {2, 3, 4} -- fp32 logits against the fp64 restatement
(oracle/fcn_oracle.py unet_lstm), the cine path (finite, pred = argmax(prob)), and label agreement of the bf16 form.  GPU box; test infrastructure (uses oracle/).
    python tools/check_lstm_variants.py"""
import dataclasses, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fcn_oracle as O
from ukbb_cardiac_amd.arch import MODELS
from ukbb_cardiac_amd.engine import Engine
from ukbb_cardiac_amd.weights import synthetic_params
base = MODELS['UNet-LSTM_ao']
bad = 0
for T, ncls in ((5, 2), (3, 4), (1, 3), (13, 3)):
    arch = dataclasses.replace(base, name='v', fc=T, n_class=ncls)
    params = synthetic_params(arch, 77 + T)
    x = np.random.default_rng(T).standard_normal((2, T, 32, 48, 1)).astype(np.float32)
    with Engine(arch, params) as eng:
        out = eng.run_seq(x, want_logits=True)
        ref = O.unet_lstm(x, params, arch.n_hidden, n_block=arch.n_block, dtype=np.float64)
        err = np.abs(out['logits'] - ref).max() / np.abs(ref).max()
        F = max(T, 6)
        frames = np.random.default_rng(9).standard_normal((F + 3, 32, 48)).astype(np.float32)
        ok_cine = True
        try:
            prob, pred = eng.run_cine(frames, weight_R=(T + 1) // 2)
            ok_cine = np.isfinite(prob).all() and np.array_equal(pred, np.argmax(prob, -1))
        except Exception as e:
            ok_cine = 'exc: %s' % e
        eng.set_precision('bf16')
        o16 = eng.run_seq(x, want_logits=True)
        agree = (o16['pred'] == out['pred']).mean()
    print('T=%d n_class=%d: rel logits err %.2e, cine ok %s, bf16 label agreement %.4f' % (T, ncls, err, ok_cine, agree))
    bad += not (err < 1e-3 and ok_cine is True and agree > 0.97)
print('FAIL' if bad else 'OK')
