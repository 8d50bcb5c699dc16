"""Config 5 experiment: the aortic U-Net's deep levels as two half-batch chains on two streams inside one forward
(UKBB_SPLIT_FROM=k at plan build: conv{k}_0 .. up{k}_1).  Prints ms per forward for k = off, 4, 3, 2, 1 and checks that the labels are
bit-identical to the unsplit plan.   python tools/split_levels.py [batch] [fp32|bf16] [rounds]      MODEL=FCN_sa HW=192,208 python tools/split_levels.py 64 fp32"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    import torch
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    prec = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    arch = MODELS[os.environ.get('MODEL', 'UNet_ao')]
    params = synthetic_params(arch, 1234)
    H, W = (int(v) for v in os.environ.get('HW', '256,256').split(','))
    img = ((cine_phantom(n, H, W, seed=5) - 0.3) / 0.25).astype(np.float32)
    x = torch.from_numpy(img).cuda()
    pred = torch.empty((n, H, W), dtype=torch.int32, device='cuda')
    ref = None
    for r in range(rounds):
        for k in (0, 4, 3, 2, 1):
            os.environ['UKBB_SPLIT_FROM'] = str(k)                     # 0 = off (the fp32 U-Net's default is 1)
            eng = Engine(arch, params)
            eng.set_precision(prec)
            for _ in range(3):
                eng.run_device(x.data_ptr(), n, H, W, pred_ptr=pred.data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                eng.run_device(x.data_ptr(), n, H, W, pred_ptr=pred.data_ptr())
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            p = pred.cpu().numpy().copy()
            if ref is None:
                ref = p
            same = bool(np.array_equal(p, ref))
            print('round %d  split from level %s: %.4f ms per forward  %.0f slices/s  labels %s' %
                  (r, k or 'off', dt * 1e3, n / dt, 'identical to the unsplit plan' if same else 'DIFFER (%d px)' % int((p != ref).sum())), flush=True)
            eng.close()
