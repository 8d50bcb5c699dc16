"""Ablation timing of the producer/consumer conv kernel (diagnostic build only):

    make -C ukbb_cardiac_amd/csrc clean all EXTRA=-DUKBB_DIAG -j8
    python tools/diag_conv.py conv2_0:124 conv3_0:120 ...

UKBB_CONV_DIAG bits: 1 = every item reads image 0 (inputs L2-hot), 2 = consumers skip LDS reads + MFMAs,
4 = producers issue no global loads, 8 = no epilogue.  Results are garbage under any bit; only the layer time matters."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    import torch
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['FCN_sa']
    params = synthetic_params(arch, 1234)
    n, h, w = 64, 192, 208
    x = torch.rand((n, h, w, 1), device='cuda')
    pred = torch.empty((n, h, w), dtype=torch.int32, device='cuda')
    for spec in sys.argv[1:]:
        layer, cfg = spec.split(':')
        row = []
        for diag in [int(v) for v in os.environ.get('DIAGS', '0,1,2,4,6,8').split(',')]:
            os.environ['UKBB_CONV_CFG'] = spec
            os.environ['UKBB_CONV_DIAG'] = str(diag)
            eng = Engine(arch, params)
            for _ in range(2):
                eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
            eng.set_timing(True)
            for _ in range(5):
                eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
            ms, cnt = eng.kernel_times()
            names, cfgs = eng.kernel_names(), eng.kernel_configs()
            i = names.index(layer)
            assert cfgs[i] == int(cfg), (cfgs[i], cfg)
            row.append('diag%d %.1f' % (diag, ms[i] / cnt[i] * 1e3))
            print('   ', spec, row[-1], file=sys.stderr, flush=True)
            eng.close()
        print('%-8s cfg %-4s us: ' % (layer, cfg) + '  '.join(row), flush=True)
