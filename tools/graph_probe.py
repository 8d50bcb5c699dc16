"""Does replaying the forward as a captured hipGraph shorten the gaps between its dependent launches?  (GPU box)
    python tools/graph_probe.py            FCN_sa N = 64 and N = 10 at 192x208, UNet_ao bf16 N = 100 at 256x256
The engine's launches are captured through torch's stream capture (any hipLaunchKernel on the capturing stream lands in the graph)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                              # noqa: E402
from ukbb_cardiac_amd.arch import MODELS                                  # noqa: E402
from ukbb_cardiac_amd.engine import Engine                                # noqa: E402
from ukbb_cardiac_amd.weights import synthetic_params                     # noqa: E402


def probe(model, n, H, W, prec=None, iters=50):
    arch = MODELS[model]
    eng = Engine(arch, synthetic_params(arch, 1234))
    if prec:
        eng.set_precision(prec)
    x = torch.rand((n, H, W), device='cuda')
    pred = torch.empty((n, H, W), dtype=torch.int32, device='cuda')
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            eng.run_device(x.data_ptr(), n, H, W, pred_ptr=pred.data_ptr(), stream=s.cuda_stream)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            eng.run_device(x.data_ptr(), n, H, W, pred_ptr=pred.data_ptr(), stream=s.cuda_stream)
        s.synchronize()
        direct = (time.perf_counter() - t0) / iters
        ref = pred.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        eng.run_device(x.data_ptr(), n, H, W, pred_ptr=pred.data_ptr(), stream=s.cuda_stream)
    pred.zero_()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / iters
    print('%-8s %-5s N=%3d %dx%d: stream launches %.1f us per forward (%.0f slices/s); captured graph %.1f us (%.0f slices/s); same labels: %s' % (
        model, prec or 'fp32', n, H, W, direct * 1e6, n / direct, graph * 1e6, n / graph, bool(torch.equal(pred, ref))), flush=True)
    eng.close()


if __name__ == '__main__':
    probe('FCN_sa', 64, 192, 208)
    probe('FCN_sa', 10, 192, 208)
    probe('FCN_sa', 1, 192, 208)
    probe('UNet_ao', 100, 256, 256, 'bf16')
    probe('UNet_ao', 10, 256, 256, 'bf16')
