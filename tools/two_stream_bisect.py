"""Which kernel of engine B disturbs engine A when both run from two streams of one process?  (r06: the bf16 U-Net fails tools/two_stream_check.py.)
Engine A (checked against its single-stream result) runs full forwards on stream 0 while engine B loops over ops [i, i] of its plan on stream 1
(UKBB_DEBUG_OPS, reading whatever an earlier full forward left in its buffers).   python tools/two_stream_bisect.py [N H W] [prec] [iterations]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    import torch
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.phantom import uniform_slices
    from ukbb_cardiac_amd.weights import synthetic_params
    n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (10, 304, 272)
    prec = sys.argv[4] if len(sys.argv) > 4 else 'bf16'
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 40
    arch = MODELS[os.environ.get('MODEL', 'UNet_ao')]
    params = synthetic_params(arch, 1234)
    dev = torch.device('cuda', 0)
    os.environ['UKBB_SPLIT_FROM'] = '0'
    xa = torch.from_numpy(uniform_slices(n, h, w, seed=10)).to(dev)
    xb = torch.from_numpy(uniform_slices(n, h, w, seed=11)).to(dev)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ea = Engine(arch, params)
    if prec != 'fp32':
        ea.set_precision(prec)
    ref = torch.empty((n, h, w), dtype=torch.int32, device=dev)
    ea.run_device(xa.data_ptr(), n, h, w, pred_ptr=ref.data_ptr())
    torch.cuda.synchronize()
    names = ea.kernel_names()
    cfgs = ea.kernel_configs()
    pa = torch.empty((n, h, w), dtype=torch.int32, device=dev)
    pb = torch.empty((n, h, w), dtype=torch.int32, device=dev)
    for i, nm in enumerate(names):
        os.environ.pop('UKBB_DEBUG_OPS', None)
        eb = Engine(arch, params)
        if prec != 'fp32':
            eb.set_precision(prec)
        eb.run_device(xb.data_ptr(), n, h, w, pred_ptr=pb.data_ptr())       # full forward: every buffer holds sane data
        torch.cuda.synchronize()
        os.environ['UKBB_DEBUG_OPS'] = '%d,%d' % (i, i)
        eb2 = Engine(arch, params)                                           # plan with only op i
        if prec != 'fp32':
            eb2.set_precision(prec)
        # eb2 needs its own buffers filled: one full forward is impossible with the restriction, so loop op i of `eb` instead: rebuild eb's plan
        eb.close(); eb = eb2
        os.environ.pop('UKBB_DEBUG_OPS', None)
        bad = 0
        for it in range(iters):
            for _ in range(6):
                eb.run_device(xb.data_ptr(), n, h, w, pred_ptr=pb.data_ptr(), stream=sb.cuda_stream)
            ea.run_device(xa.data_ptr(), n, h, w, pred_ptr=pa.data_ptr(), stream=sa.cuda_stream)
            for _ in range(6):
                eb.run_device(xb.data_ptr(), n, h, w, pred_ptr=pb.data_ptr(), stream=sb.cuda_stream)
            torch.cuda.synchronize()
            bad += int((pa != ref).any())
        print('B loops op %2d %-22s cfg %4d: A wrong in %d of %d forwards %s' % (i, nm, cfgs[i], bad, iters, 'DISTURBS' if bad else ''), flush=True)
        eb.close()
