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
    prec = sys.argv[3] if len(sys.argv) > 3 else 'fp32'
    tol = 1e-4 if prec == 'fp32' else 6e-2                       # bf16: storage rounding of features / gx / h accumulates over the steps
    arch = MODELS['UNet-LSTM_ao']
    params = synthetic_params(arch, 1234)
    T, NH = 9, 16
    x = np.random.default_rng(1).standard_normal((1, T, H, W, 1)).astype(np.float32)
    def widen(buf, n):                                           # bf16 plans keep the hidden maps as bf16 in the same buffers
        if prec == 'fp32':
            return buf[:n]
        u = buf.view(np.uint16)[:n].astype(np.uint32) << 16
        return u.view(np.float32)
    with Engine(arch, params) as eng:
        if prec != 'fp32':
            eng.set_precision(prec)
        out = eng.run_seq(x, want_logits=True)
        feat = eng.activation('up0').reshape(T, H, W, 16)
        h1 = widen(eng.activation('lstm:h1'), 2 * T * H * W * NH).reshape(2, T, H, W, NH)
        hall = widen(eng.activation('lstm:hall'), 2 * T * H * W * NH).reshape(2, T, H, W, NH)
        gx_all = eng.activation('lstm:gx')
    f64 = feat.astype(np.float64)
    zeros = np.zeros((1, H, W, NH))
    bad = 0
    for d, name in enumerate(('lstm_fw', 'lstm_bw')):
        p = {k: np.asarray(v, np.float64) for k, v in params[name].items()}
        # x pass: per frame, first step from the zero state
        for f in range(T):
            hh, _ = O.conv_lstm_cell(f64[f:f + 1], zeros, zeros, p)
            err = np.abs(h1[d, f] - hh[0]).max()
            if err > tol:
                bad += 1
                idx = np.unravel_index(np.argmax(np.abs(h1[d, f] - hh[0])), hh[0].shape)
                print('%s h1 frame %d: max err %.3e at (y, x, c) = %s  got %.5f want %.5f' % (name, f, err, idx, h1[d, f][idx], hh[0][idx]))
        if os.environ.get('UKBB_DEBUG_LSTM_TILES'):             # which 2 x 32-pixel tiles of each frame's h1 are off (bf16 form: one tile per wave and round)
            for f in range(T):
                hh, _ = O.conv_lstm_cell(f64[f:f + 1], zeros, zeros, p)
                e = np.abs(h1[d, f] - hh[0]).max(axis=-1)
                rows = [''.join('x' if e[y:y + 2, x0:x0 + 32].max() > tol else '.' for x0 in range(0, W, 32)) for y in range(0, H, 2)]
                print('%s frame %d tiles (rows of 2 px, cols of 32 px): %s' % (name, f, ' '.join(rows)))
        if os.environ.get('UKBB_DEBUG_LSTM_GX') and prec != 'fp32':     # bf16 form: the x pass's gx (lane-native) against W_x * x + b
            ty_n, tx_n = (H + 1) // 2, (W + 31) // 32
            per_img = ty_n * tx_n * 2 * 2048
            gx_raw = gx_all.view(np.uint16)
            kern = p['kernel'][:, :, :16, :]
            for f in range(T):
                z = O.conv2d_same(f64[f:f + 1], kern, 1)[0] + p['bias']           # [H][W][64], gates i | j | f | o x 16
                blk = (gx_raw[(d * T + f) * per_img:(d * T + f + 1) * per_img].astype(np.uint32) << 16).view(np.float32)
                blk = blk.reshape(ty_n, tx_n, 2, 4, 2, 32, 8)                     # tile y, tile x, row, gate, lane half, pixel, hidden channel of the half
                got = np.zeros((H, W, 64), np.float32)
                for ty in range(ty_n):
                    for tx in range(tx_n):
                        for r in range(2):
                            y = ty * 2 + r
                            x1 = min(W, tx * 32 + 32)
                            for q in range(4):
                                for g in range(2):
                                    got[y, tx * 32:x1, q * 16 + 8 * g:q * 16 + 8 * g + 8] = blk[ty, tx, r, q, g, :x1 - tx * 32, :]
                e = np.abs(got - z).max(axis=-1)
                rows = [''.join('x' if e[y:y + 2, x0:x0 + 32].max() > 0.05 else '.' for x0 in range(0, W, 32)) for y in range(0, H, 2)]
                print('%s frame %d gx tiles: %s   (max err %.3g)' % (name, f, ' '.join(rows), e.max()))
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
            bad += err > tol
    ref = O.unet_lstm(x, params, arch.n_hidden, n_block=arch.n_block, dtype=np.float64)
    print('logits: max err %.3e (scale %.3f)' % (np.abs(out['logits'] - ref).max(), np.abs(ref).max()))
    print('FAIL' if bad else 'OK')
    sys.exit(1 if bad else 0)
