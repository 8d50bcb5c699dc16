"""Ablation timing of fcn_head_pc_kernel (diagnostic build, make EXTRA=-DUKBB_DIAG): UKBB_HEAD_DIAG bits
1 = producers skip the gather FMAs, 2 = consumers skip out0/out1/logits (same_dim0 only), 4 = consumers skip the logits/softmax VALU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ukbb_cardiac_amd.arch import MODELS
from ukbb_cardiac_amd.engine import Engine
from ukbb_cardiac_amd.weights import synthetic_params
arch = MODELS['FCN_sa']; params = synthetic_params(arch, 1234)
n, h, w = 64, 192, 208
x = torch.rand((n, h, w, 1), device='cuda'); pred = torch.empty((n, h, w), dtype=torch.int32, device='cuda')
for d in (0, 1, 4, 5, 2, 3):
    os.environ['UKBB_HEAD_DIAG'] = str(d)
    eng = Engine(arch, params)
    for _ in range(3): eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
    eng.set_timing(True)
    for _ in range(8): eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
    ms, cnt = eng.kernel_times()
    i = eng.kernel_names().index('head')
    print('head diag %d: %.1f us' % (d, ms[i] / cnt[i] * 1e3), flush=True)
    eng.close()
