"""Stage-by-stage check of the fused ConvLSTM kernels (kernels_wino24.hip, ConvArgs::ls_mode) against numpy on one small window:
the x pass's per-frame first step h1 (both directions), then every step's hidden map.  GPU box; test infrastructure (uses oracle/).
    python tools/debug_lstm.py [H W]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fcn_oracle as O                                    # noqa: E402
from ukbb_cardiac_amd.arch import MODELS                              # noqa: E402
from ukbb_cardiac_amd.engine import Engine                            # noqa: E402
from ukbb_cardiac_amd.weights import synthetic_params                 # noqa: E402

if __name__ == '__main__':
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 48)
    arch = MODELS['UNet-LSTM_ao']
    params = synthetic_params(arch, 1234)
    T, NH = 9, 16
    x = np.random.default_rng(1).standard_normal((1, T, H, W, 1)).astype(np.float32)
    with Engine(arch, params) as eng:
        out = eng.run_seq(x, want_logits=True)
        feat = eng.activation('up0').reshape(T, H, W, 16)
        h1 = eng.activation('lstm:h1')[:2 * T * H * W * NH].reshape(2, T, H, W, NH)
        hall = eng.activation('lstm:hall')[:2 * T * H * W * NH].reshape(2, T, H, W, NH)
    f64 = feat.astype(np.float64)
    zeros = np.zeros((1, H, W, NH))
    bad = 0
    for d, name in enumerate(('lstm_fw', 'lstm_bw')):
        p = {k: np.asarray(v, np.float64) for k, v in params[name].items()}
        # x pass: per frame, first step from the zero state
        for f in range(T):
            hh, _ = O.conv_lstm_cell(f64[f:f + 1], zeros, zeros, p)
            err = np.abs(h1[d, f] - hh[0]).max()
            if err > 1e-4:
                bad += 1
                idx = np.unravel_index(np.argmax(np.abs(h1[d, f] - hh[0])), hh[0].shape)
                print('%s h1 frame %d: max err %.3e at (y, x, c) = %s  got %.5f want %.5f' % (name, f, err, idx, h1[d, f][idx], hh[0][idx]))
        if d == 0 and os.environ.get('UKBB_DEBUG_LSTM_MATCH'):
            hh, _ = O.conv_lstm_cell(f64[0:1], zeros, zeros, p)
            want, got = hh[0], h1[0, 0]
            print('per-channel max err:', ' '.join('%.1e' % np.abs(got[..., c] - want[..., c]).max() for c in range(NH)))
            for c in range(NH):                           # which (channel, x offset within the 4-pixel tile) of the oracle does channel c of pixel column j hold?
                for j in range(4):
                    g = got[:, j::4, c]
                    best = min(((np.abs(g - want[:, jj::4, cc]).max(), cc, jj) for cc in range(NH) for jj in range(4)))
                    print('got ch %2d col %d  <-  want ch %2d col %d (err %.1e)' % (c, j, best[1], best[2], best[0]))
        hprev, cprev = zeros, zeros
        order = range(T) if d == 0 else range(T - 1, -1, -1)
        for n, t in enumerate(order):
            hprev, cprev = O.conv_lstm_cell(f64[t:t + 1], hprev, cprev, p)
            got = h1[d, t] if n == 0 else hall[d, t]
            err = np.abs(got - hprev[0]).max()
            print('%s step %d (frame %d): max |h - oracle| %.3e' % (name, n, t, err))
            bad += err > 1e-4
    ref = O.unet_lstm(x, params, arch.n_hidden, n_block=arch.n_block, dtype=np.float64)
    print('logits: max err %.3e (scale %.3f)' % (np.abs(out['logits'] - ref).max(), np.abs(ref).max()))
    print('FAIL' if bad else 'OK')
    sys.exit(1 if bad else 0)
