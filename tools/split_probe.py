"""Does a small batch run faster as S image ranges on S streams?  (FCN_sa, N = 10 x 192x208: every launch pays ~10 us of fill / drain that
another range's kernels could hide.)  Prototype with S engines = S workspaces; per forward: fork from the main stream, S ranges, join.
GPU box.   python tools/split_probe.py [N] [S ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    import torch
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.weights import synthetic_params
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    splits = [int(v) for v in sys.argv[2:]] or [1, 2, 3, 5]
    H, W = 192, 208
    arch = MODELS['FCN_sa']
    params = synthetic_params(arch, 1234)
    x = torch.rand((n, H, W), device='cuda')
    pred = torch.empty((n, H, W), dtype=torch.int32, device='cuda')
    ref = None
    for S in splits:
        engs = [Engine(arch, params) for _ in range(S)]
        streams = [torch.cuda.Stream() for _ in range(S)]
        main = torch.cuda.current_stream()
        bounds = [n * i // S for i in range(S + 1)]

        def forward():
            if S == 1:
                engs[0].run_device(x.data_ptr(), n, H, W, pred_ptr=pred.data_ptr(), stream=main.cuda_stream)
                return
            ev = torch.cuda.Event()
            ev.record(main)
            for i in range(S):
                a, b = bounds[i], bounds[i + 1]
                streams[i].wait_event(ev)
                engs[i].run_device(x.data_ptr() + a * H * W * 4, b - a, H, W, pred_ptr=pred.data_ptr() + a * H * W * 4, stream=streams[i].cuda_stream)
                e2 = torch.cuda.Event()
                e2.record(streams[i])
                main.wait_event(e2)
        for _ in range(5):
            forward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 50
        for _ in range(k):
            forward()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        p = pred.cpu()
        same = True if ref is None else bool((p == ref).all())
        ref = p if ref is None else ref
        print('N = %d as %d range(s): %.1f us per forward = %.0f slices/s; labels identical to S = %d: %s' % (n, S, dt * 1e6, n / dt, splits[0], same))
        for e in engs:
            e.close()
