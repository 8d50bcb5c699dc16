"""Aortic U-Net throughput, fp32 vs bf16 MFMA operands (BASELINE config 5; GPU box).
    python tools/bench_unet.py [batch] [fp32,bf16 | bf16 | fp32] [timed steps]
With a single precision only that path runs (what the rocprofv3 counter passes of tools/profile_unet.sh wrap)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ukbb_cardiac_amd.arch import MODELS, fcn_macs_per_slice           # noqa: E402
from ukbb_cardiac_amd.engine import Engine                            # noqa: E402
from ukbb_cardiac_amd.image_utils import np_categorical_dice         # noqa: E402
from ukbb_cardiac_amd.phantom import cine_phantom                     # noqa: E402
from ukbb_cardiac_amd.weights import synthetic_params                 # noqa: E402

if __name__ == '__main__':
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    arch = MODELS['UNet_ao']
    eng = Engine(arch, synthetic_params(arch, 1234))
    img = ((cine_phantom(n, 256, 256, seed=5) - 0.3) / 0.25).astype(np.float32)
    x = torch.from_numpy(img).cuda()
    pred = torch.empty((n, 256, 256), dtype=torch.int32, device='cuda')
    m3, m1 = fcn_macs_per_slice(arch, 256, 256)
    precs = tuple(sys.argv[2].split(',')) if len(sys.argv) > 2 else ('fp32', 'bf16')
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50          # a timed region of >= 50 ms (one host synchronisation costs 0.3-0.5 ms)
    res = {}
    for prec in precs:
        eng.set_precision(prec)
        for _ in range(3):
            eng.run_device(x.data_ptr(), n, 256, 256, pred_ptr=pred.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = steps
        for _ in range(k):
            eng.run_device(x.data_ptr(), n, 256, 256, pred_ptr=pred.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        res[prec] = pred.cpu().numpy().copy()
        print('UNet_ao %s: N=%d 256x256: %.2f ms/step  %.0f slices/s  %.1f TFLOP/s algorithmic'
              % (prec, n, dt * 1e3, n / dt, 2.0 * (m3 + m1) * n / dt / 1e12))
    print('forwards_total=%d (3 warm-up + %d timed per precision)' % ((3 + steps) * len(precs), steps))
    if 'bf16' in res and 'fp32' in res:
      print('Dice bf16 vs fp32: class1 %.4f class2 %.4f; label disagreement %.4f %%' % (
        np_categorical_dice(res['bf16'], res['fp32'], 1), np_categorical_dice(res['bf16'], res['fp32'], 2),
        100.0 * (res['bf16'] != res['fp32']).mean()))
