"""Throughput of the other BASELINE.json configurations (parity-test cases, not the bench line):
config 3 (mixed models / shapes) and config 4 (subject-granular scale-out, run here with G = 1; under
torchrun each rank takes subjects i % G == rank with no collective).  GPU box only.

    python tools/bench_configs.py            # config 3 + config 4 (G = 1, 40 subjects)
    python -m torch.distributed.run --nproc-per-node 8 ... tools/bench_configs.py --subjects 1000
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    import torch
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.shard import subjects_for_shard
    from ukbb_cardiac_amd.weights import synthetic_params
    ap = argparse.ArgumentParser()
    ap.add_argument('--subjects', type=int, default=40)
    args = ap.parse_args()
    rank = int(os.environ.get('RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    def rate(model, n, h, w, steps=10):
        arch = MODELS[model]
        eng = Engine(arch, synthetic_params(arch, 1234), device=local)
        x = torch.rand((n, h, w), device=dev)
        pred = torch.empty((n, h, w), dtype=torch.int32, device=dev)
        for _ in range(3):
            eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        eng.close()
        return n / dt, dt * 1e3

    if rank == 0:
        print('config 3 (mixed):')
        for model, n, h, w in [('FCN_sa', 10, 192, 208), ('FCN_la_2ch', 50, 176, 208), ('FCN_la_4ch', 50, 176, 208),
                               ('FCN_la_4ch_seg4', 50, 176, 208), ('FCN_sa', 64, 208, 256)]:
            r, ms = rate(model, n, h, w)
            print('  %-16s N=%-3d %dx%d: %8.0f slices/s (%.3f ms per call)' % (model, n, h, w, r, ms))
    # config 4: subjects of 500 slices generated on device from seed = subject id
    arch = MODELS['FCN_sa']
    eng = Engine(arch, synthetic_params(arch, 1234), device=local)
    mine = subjects_for_shard(list(range(args.subjects)), rank, world)
    n, h, w = 500, 192, 208
    pred = torch.empty((n, h, w), dtype=torch.int32, device=dev)
    eng.reserve(128, h, w)
    g = torch.Generator(device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for subj in mine:
        g.manual_seed(subj)
        x = torch.rand((n, h, w), device=dev, generator=g)
        for i in range(0, n, 128):
            m = min(128, n - i)
            eng.run_device(x[i].data_ptr(), m, h, w, pred_ptr=pred[i].data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('config 4 rank %d/%d: %d subjects x 500 slices in %.2f s = %.0f slices/s on this GPU (no collective; '
          'aggregate = sum over ranks)' % (rank, world, len(mine), dt, len(mine) * n / dt))
