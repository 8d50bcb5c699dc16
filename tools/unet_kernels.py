"""Per-kernel times of the aortic U-Net (and UNet-LSTM features) at 256x256, batch N:  python tools/unet_kernels.py [N] [fp32|bf16]
MODEL=FCN_sa HW=192,208 python tools/unet_kernels.py 10   gives the same table for another model / slice size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ukbb_cardiac_amd import _lib
from ukbb_cardiac_amd.arch import MODELS
from ukbb_cardiac_amd.engine import Engine
from ukbb_cardiac_amd.weights import synthetic_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
arch = MODELS[os.environ.get('MODEL', 'UNet_ao')]
H, W = (int(v) for v in os.environ.get('HW', '256,256').split(','))
eng = Engine(arch, synthetic_params(arch, 1234))
if len(sys.argv) > 2:
    eng.set_precision(sys.argv[2])
x = torch.rand((n, H, W, 1), device='cuda'); pred = torch.empty((n, H, W), dtype=torch.int32, device='cuda')
for _ in range(3): eng.run_device(x.data_ptr(), n, H, W, pred_ptr=pred.data_ptr())
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20): eng.run_device(x.data_ptr(), n, H, W, pred_ptr=pred.data_ptr())
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20
eng.set_timing(True)
for _ in range(5): eng.run_device(x.data_ptr(), n, H, W, pred_ptr=pred.data_ptr())
ms, cnt = eng.kernel_times()
macs = eng.kernel_macs()
tot = 0
bf = len(sys.argv) > 2 and sys.argv[2] == 'bf16'
peak = 2500e12 if bf else 157.3e12                                 # dense bf16 / fp32 MFMA peak (MI355X_MICROARCH.md)
for nm, cfg, m, c, mac in zip(eng.kernel_names(), eng.kernel_configs(), ms, cnt, macs):
    t = m / c
    tot += t
    print('%-10s cfg %4d %-48s %7.1f us  %.3f of the %s MFMA peak (reference-graph FLOPs)' % (nm, cfg, _lib.lib.ukbb_fcn_conv_config_name(cfg).decode() if cfg >= 0 else '', t * 1e3, 2 * mac / (t * 1e-3) / peak, 'bf16' if bf else 'fp32'))
print('sum %.1f us -> %.0f slices/s ; untimed stream: %.1f us per forward -> %.0f slices/s' % (tot * 1e3, n / (tot * 1e-3), wall * 1e6, n / wall))
