"""Winograd kernel phase stamps (diagnostic build: make EXTRA=-DUKBB_WINO_STAMPS): prints, per Winograd layer of one forward,
producer / consumer cycles per stage; UKBB_CONV_DIAG=32 additionally drops the output stores.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ukbb_cardiac_amd.arch import MODELS
from ukbb_cardiac_amd.engine import Engine
from ukbb_cardiac_amd.weights import synthetic_params
arch = MODELS['FCN_sa']; params = synthetic_params(arch, 1234)
n, h, w = 64, 192, 208
x = torch.rand((n, h, w, 1), device='cuda'); pred = torch.empty((n, h, w), dtype=torch.int32, device='cuda')
eng = Engine(arch, params)
for _ in range(3): eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
torch.cuda.synchronize()
os.environ['UKBB_STAMPS'] = '1'
eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
torch.cuda.synchronize()
del os.environ['UKBB_STAMPS']
eng.set_timing(True)
for _ in range(5): eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
ms, cnt = eng.kernel_times()
print({k: round(m / c * 1e3, 1) for k, m, c in zip(eng.kernel_names(), ms, cnt)})
