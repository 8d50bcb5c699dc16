"""Do kernels that use scratch (private-segment) memory survive running CONCURRENTLY from two HIP streams of one process?
r06: the half-batch chains of the bf16 U-Net produced sporadic garbage exactly when `conv_wr_kernel<2,2,8>` (24 bytes of scratch: the
only spilling kernel in the range) ran from two streams at once; with the scratch-free tiling of the same layer: never.  This tool runs TWO
engines of one model on two streams, batches in flight on both, and compares every label map with the single-stream reference.
    python tools/two_stream_check.py [model] [N H W] [iterations]        GPU box; prints OK / FAILED."""
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
    model = sys.argv[1] if len(sys.argv) > 1 else 'FCN_sa'
    n, h, w = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (64, 192, 208)
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 300
    prec = os.environ.get('PREC', 'fp32')
    arch = MODELS[model]
    params = synthetic_params(arch, 1234)
    dev = torch.device('cuda', 0)
    xs = [torch.from_numpy(uniform_slices(n, h, w, seed=10 + i)).to(dev) for i in range(2)]
    engs = [Engine(arch, params) for _ in range(2)]
    for e in engs:
        if prec != 'fp32':
            e.set_precision(prec)
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    refs = []
    for i in range(2):                                           # single-stream references
        p = torch.empty((n, h, w), dtype=torch.int32, device=dev)
        engs[i].run_device(xs[i].data_ptr(), n, h, w, pred_ptr=p.data_ptr())
        torch.cuda.synchronize()
        refs.append(p.clone())
    preds = [[torch.empty((n, h, w), dtype=torch.int32, device=dev) for _ in range(4)] for _ in range(2)]
    bad = 0
    for it in range(0, iters, 4):
        for k in range(4):                                       # 4 batches in flight per stream before anything is checked
            for i in range(2):
                engs[i].run_device(xs[i].data_ptr(), n, h, w, pred_ptr=preds[i][k].data_ptr(), stream=streams[i].cuda_stream)
        torch.cuda.synchronize()
        for i in range(2):
            for k in range(4):
                d = int((preds[i][k] != refs[i]).sum())
                if d:
                    bad += 1
                    if bad <= 5:
                        print('iteration %d stream %d: %d label pixels differ from the single-stream result' % (it + k, i, d), flush=True)
    print('%s %s %dx%dx%d: %d forwards per stream on two streams, %d with differing labels' % (model, prec, n, h, w, iters, bad))
    print('OK' if bad == 0 else 'FAILED')
    sys.exit(1 if bad else 0)
