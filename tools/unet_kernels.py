"""Per-kernel times of the aortic U-Net (and UNet-LSTM features) at 256x256, batch N:  python tools/unet_kernels.py [N] [fp32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ukbb_cardiac_amd import _lib
from ukbb_cardiac_amd.arch import MODELS
from ukbb_cardiac_amd.engine import Engine
from ukbb_cardiac_amd.weights import synthetic_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
arch = MODELS['UNet_ao']
eng = Engine(arch, synthetic_params(arch, 1234))
if len(sys.argv) > 2:
    eng.set_precision(sys.argv[2])
x = torch.rand((n, 256, 256, 1), device='cuda'); pred = torch.empty((n, 256, 256), dtype=torch.int32, device='cuda')
for _ in range(3): eng.run_device(x.data_ptr(), n, 256, 256, pred_ptr=pred.data_ptr())
eng.set_timing(True)
for _ in range(5): eng.run_device(x.data_ptr(), n, 256, 256, pred_ptr=pred.data_ptr())
ms, cnt = eng.kernel_times()
macs = eng.kernel_macs()
tot = 0
for nm, cfg, m, c, mac in zip(eng.kernel_names(), eng.kernel_configs(), ms, cnt, macs):
    t = m / c
    tot += t
    print('%-10s cfg %4d %-48s %7.1f us  %.2f of peak (reference-graph FLOPs)' % (nm, cfg, _lib.lib.ukbb_fcn_conv_config_name(cfg).decode() if cfg >= 0 else '', t * 1e3, 2 * mac / (t * 1e-3) / 157.3e12))
print('sum %.1f us -> %.0f slices/s' % (tot * 1e3, n / (tot * 1e-3)))
