"""Aortic UNet-LSTM (the reference's default aortic model): one slice position, T = 100 frames of 256x256,
circular 9-frame windows (common/deploy_network_ao.py:129-183).  GPU box only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    import ctypes as C
    import torch
    from ukbb_cardiac_amd import _lib
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['UNet-LSTM_ao']
    params = synthetic_params(arch, 1234)
    if os.environ.get('UNI'):                                          # the single-direction head (network_ao.py:214-252) through its zero-backward-cell embedding
        from ukbb_cardiac_amd.weights import embed_unidirectional_lstm
        uni = {k: v for k, v in params.items() if not k.startswith('lstm')}
        uni['lstm'] = params['lstm_fw']
        uni['lstm_conv'] = {'kernel': params['lstm_out']['kernel'][:, :, :arch.same_dim], 'bias': params['lstm_out']['bias']}
        params = embed_unidirectional_lstm(uni, arch.same_dim)
    eng = Engine(arch, params)
    F, H, W = 100, 256, 256
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5                  # timed cines (profilers pass a small count)
    prec = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
    if prec != 'fp32':
        eng.set_precision(prec)
    x = torch.randn((F, H, W), device='cuda')
    prob = torch.empty((F, H, W, 3), device='cuda')
    pred = torch.empty((F, H, W), dtype=torch.int32, device='cuda')

    def step():
        _lib.check(_lib.lib.ukbb_fcn_forward_cine(eng._h, C.c_void_p(x.data_ptr()), F, H, W, 5, 0.1, 1, C.c_void_p(prob.data_ptr()),
                                                  C.c_void_p(pred.data_ptr()), None), 'forward_cine')
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print('cines_total=%d' % (n + 2))
    unet = 3183.5e6 * 2                      # FLOP per 256x256 frame through the U-Net
    lstm = 2 * 9 * 256 * 256 * (9 * 32 * 64) * 2 + 9 * 256 * 256 * 32 * 3 * 2     # per window: 18 gate convs + 9 output convs
    ref_flop = F * (9 * unet + lstm)         # the reference recomputes the U-Net for each of the 9 window positions
    our_flop = F * (unet + lstm)
    print('UNet-LSTM cine, %d frames of %dx%d, %s: %.2f ms per slice position = %.0f frames/s' % (F, H, W, prec, dt * 1e3, F / dt))
    print('   work as the reference executes it: %.2f TFLOP (U-Net 9x per frame) -> %.0f TFLOP/s equivalent; '
          'as executed here (features once): %.2f TFLOP -> %.0f TFLOP/s' % (ref_flop / 1e12, ref_flop / dt / 1e12, our_flop / 1e12, our_flop / dt / 1e12))
