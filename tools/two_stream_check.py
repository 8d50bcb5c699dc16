"""Is a forward right while kernels of ANOTHER stream of the same process run beside it?
Two engines of one model on two streams, four batches in flight on each, every label map compared with the single-stream result.
r06: this is the check that exposed (and now guards) the wide-store hazard of kernels_ws.hip -- the bf16-storage U-Net failed 85-98 % of
its forwards here until its 16-byte buffer stores were padded (profiles/r06_notes.md section 10).  A 'UNet-LSTM...' model runs whole cines
(N = frames) through ukbb_fcn_forward_cine.
    PREC=fp32|bf16|f32x3 python tools/two_stream_check.py [model] [N H W] [iterations]        GPU box; prints OK / FAILED."""
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
    lstm = model.startswith('UNet-LSTM')                        # a cine of n frames per "forward" (ukbb_fcn_forward_cine), frames [n, h, w]
    xs = [torch.from_numpy(uniform_slices(n, h, w, seed=10 + i)[..., 0].copy() if lstm else uniform_slices(n, h, w, seed=10 + i)).to(dev) for i in range(2)]
    engs = [Engine(arch, params) for _ in range(2)]
    for e in engs:
        if prec != 'fp32':
            e.set_precision(prec)
    streams = [torch.cuda.Stream(dev) for _ in range(2)]

    def forward(i, pred, stream=0):
        if lstm:
            engs[i].run_cine_device(xs[i].data_ptr(), n, h, w, probs[i].data_ptr(), pred.data_ptr(), stream=stream)
        else:
            engs[i].run_device(xs[i].data_ptr(), n, h, w, pred_ptr=pred.data_ptr(), stream=stream)
    probs = [torch.empty((n, h, w, arch.n_class), dtype=torch.float32, device=dev) for _ in range(2)] if lstm else None   # (one per stream: graded on the labels)
    refs = []
    for i in range(2):                                           # single-stream references
        p = torch.empty((n, h, w), dtype=torch.int32, device=dev)
        forward(i, p)
        torch.cuda.synchronize()
        refs.append(p.clone())
    preds = [[torch.empty((n, h, w), dtype=torch.int32, device=dev) for _ in range(4)] for _ in range(2)]
    bad = 0
    for it in range(0, iters, 4):
        for k in range(4):                                       # 4 batches in flight per stream before anything is checked
            for i in range(2):
                forward(i, preds[i][k], streams[i].cuda_stream)
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
