"""Time given (layer:cfg) pairs on the bench workload (GPU box):  python tools/sweep_convs.py conv2_0:124,127 conv3_0:124,126 ...
   MODEL=UNet_ao SHAPE=100,256,256 selects another model / batch shape, PREC=bf16 the precision mode."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    import torch
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS[os.environ.get('MODEL', 'FCN_sa')]
    params = synthetic_params(arch, 1234)
    n, h, w = (int(v) for v in os.environ.get('SHAPE', '64,192,208').split(','))
    x = torch.rand((n, h, w, 1), device='cuda')
    pred = torch.empty((n, h, w), dtype=torch.int32, device='cuda')
    for spec in sys.argv[1:]:
        layer, cfgs = spec.split(':')
        row = []
        for cfg in cfgs.split(','):
            parts = layer.split('+')                                   # plan names of fused launches: conv0_0+conv0_1, up0_1+logits
            ov = parts[1] if parts[0] == 'conv0_0' and len(parts) > 1 else parts[0]
            os.environ['UKBB_CONV_CFG'] = '%s:%s' % (ov, cfg)
            eng = Engine(arch, params)
            if os.environ.get('PREC'):
                eng.set_precision(os.environ['PREC'])
            for _ in range(3):
                eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
            eng.set_timing(True)
            for _ in range(8):
                eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
            ms, cnt = eng.kernel_times()
            names, got = eng.kernel_names(), eng.kernel_configs()
            i = names.index(layer)
            row.append('%s:%.1f%s' % (cfg, ms[i] / cnt[i] * 1e3, '' if got[i] == int(cfg) else '(ran %d)' % got[i]))
            eng.close()
        print('%-8s us  ' % layer + '  '.join(row), flush=True)
