"""Stability soak on one GPU: many forwards, many engine create / close cycles, many subjects through the subject pipeline;
checks that results stay bit-identical and that device memory returns to where it started.  GPU box only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    import torch
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.subject_pipeline import SubjectPipeline
    from ukbb_cardiac_amd.weights import synthetic_params
    free0 = torch.cuda.mem_get_info()[0]
    arch = MODELS['FCN_sa']
    params = synthetic_params(arch, 1234)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.random((64, 192, 208, 1), dtype=np.float32)).cuda()
    pred = torch.empty((64, 192, 208), dtype=torch.int32, device='cuda')
    eng = Engine(arch, params)
    eng.run_device(x.data_ptr(), 64, 192, 208, pred_ptr=pred.data_ptr())
    torch.cuda.synchronize()
    ref = pred.clone()
    t0 = time.time()
    bad = 0
    for i in range(3000):
        eng.run_device(x.data_ptr(), 64, 192, 208, pred_ptr=pred.data_ptr())
        if i % 500 == 499:
            torch.cuda.synchronize()
            bad += int((pred != ref).sum())
    torch.cuda.synchronize()
    print('3000 forwards of 64x192x208 in %.2f s (%.0f slices/s), label differences vs the first run: %d' %
          (time.time() - t0, 3000 * 64 / (time.time() - t0), bad), flush=True)
    eng.close()
    torch.cuda.synchronize()
    free_mid = torch.cuda.mem_get_info()[0]
    for i in range(40):                                           # create / run / close, different shapes
        e = Engine(arch, params)
        h, w = 16 * (2 + i % 9), 16 * (3 + i % 7)
        xi = x[:8, :h, :w].contiguous()
        pi = torch.empty((8, h, w), dtype=torch.int32, device='cuda')
        e.run_device(xi.data_ptr(), 8, h, w, pred_ptr=pi.data_ptr())
        torch.cuda.synchronize()
        e.close()
    torch.cuda.synchronize()
    print('40 engine create / run / close cycles done; device memory free before / after them: %.1f / %.1f MB' %
          (free_mid / 1e6, torch.cuda.mem_get_info()[0] / 1e6), flush=True)
    eng = Engine(arch, params)
    shape = (192, 208, 10, 50)
    vols = [np.asfortranarray((1000 * rng.gamma(2.0, 1.0, size=shape)).astype(np.float32)) for _ in range(2)]
    pipe = SubjectPipeline(eng, shape, 128)
    first = {}
    n = 0
    t0 = time.time()
    for k, res in enumerate(pipe.run(vols[i % 2] for i in range(120))):
        key = k % 2
        if key not in first:
            first[key] = res.labels.copy()
        elif not np.array_equal(first[key], res.labels):
            bad += 1
        n += 1
    print('%d subjects through the subject pipeline in %.2f s (%.1f ms each), mismatching label volumes: %d' %
          (n, time.time() - t0, (time.time() - t0) / n * 1e3, bad), flush=True)
    # r06: the last Result of the loop keeps the pipeline (and its 3 slots x 260 MB of device buffers) alive through its back reference --
    # r04 / r05 printed "1010.8 MB less free memory" here because `res` outlived `del pipe`, not because anything leaked
    # (bench.py's 1000-subject cohort: HIP-level free memory moves by 2 MB, torch's reserved pool is constant)
    del res, pipe
    import gc
    gc.collect()
    eng.close()
    del x, pred, ref
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    print('device memory free before / after: %.1f / %.1f MB (difference %.1f MB)' % (free0 / 1e6, free1 / 1e6, (free0 - free1) / 1e6), flush=True)
    sys.exit(1 if bad else 0)
