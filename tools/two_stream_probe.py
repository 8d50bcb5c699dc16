"""Throughput of the bench workload with 1, 2 and 3 batches in flight (one engine + one HIP stream each).  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ukbb_cardiac_amd.arch import MODELS
from ukbb_cardiac_amd.engine import Engine
from ukbb_cardiac_amd.weights import synthetic_params
from ukbb_cardiac_amd.phantom import uniform_slices
arch = MODELS['FCN_sa']; params = synthetic_params(arch, 1234)
n, h, w = 64, 192, 208
x = torch.from_numpy(uniform_slices(n, h, w, seed=1)).cuda()
for S in (1, 2, 3):
    engs = [Engine(arch, params) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    preds = [torch.empty((n, h, w), dtype=torch.int32, device='cuda') for _ in range(S)]
    for e in engs: e.reserve(n, h, w)
    def step(i):
        k = i % S
        engs[k].run_device(x.data_ptr(), n, h, w, pred_ptr=preds[k].data_ptr(), stream=streams[k].cuda_stream)
    for i in range(6): step(i)
    torch.cuda.synchronize()
    K = 40
    t0 = time.perf_counter()
    for i in range(K): step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('streams %d: %.1f slices/s, %.4f ms per step' % (S, n * K / dt, dt / K * 1e3), flush=True)
    assert all(torch.equal(preds[0], p) for p in preds)
    for e in engs: e.close()
