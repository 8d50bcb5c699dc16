"""End-to-end time of ONE subject through the sequence path (BASELINE config 1 volume: 192x208x10x50
float32), host pre/post-processing (numpy mirror of common/deploy_network.py:86-131) vs the device
pipeline (ukbb_cardiac_amd/device_pipeline.py).  File I/O (gzip NIfTI) excluded.  GPU box only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    import torch
    from ukbb_cardiac_amd import device_pipeline as dp
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.image_utils import rescale_intensity
    from ukbb_cardiac_amd.pipeline import pick_ed_es, segment_sequence
    from ukbb_cardiac_amd.weights import synthetic_params

    arch = MODELS['FCN_sa']
    eng = Engine(arch, synthetic_params(arch, 1234))
    rng = np.random.default_rng(0)
    vol = np.asfortranarray((1000 * rng.gamma(2.0, 1.0, size=(192, 208, 10, 50))).astype(np.float32))
    fwd = lambda b: eng.run(b, want_prob=False)

    def timeit(f, reps):
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, r

    th, want = timeit(lambda: segment_sequence(vol.copy(order='F'), fwd, 128), 2)
    td, got = timeit(lambda: dp.segment_sequence_device(vol, eng, 128), 5)
    assert np.array_equal(want, got)
    tp, _ = timeit(lambda: np.percentile(vol, (1, 99)), 2)
    tr, _ = timeit(lambda: rescale_intensity(vol.copy(order='F'), (1, 99)), 2)
    t = torch.from_numpy(vol).cuda()
    ts, _ = timeit(lambda: dp.device_percentiles(t, (1, 99)), 10)
    n = vol.shape[2] * vol.shape[3]
    print('subject 192x208x10x50 (500 slices), labels identical on both paths')
    print('host pre/post-processing + forward_host: %7.1f ms per subject  (%6.0f slices/s)' % (th * 1e3, n / th))
    print('   of which np.percentile(vol, (1,99)):  %7.1f ms;  rescale_intensity total %7.1f ms' % (tp * 1e3, tr * 1e3))
    print('device pipeline (H2D 80 MB, select, pack, forward, unpack, D2H 20 MB + float64 volume): %7.1f ms  (%6.0f slices/s)'
          % (td * 1e3, n / td))
    print('   of which exact percentiles on device: %7.2f ms' % (ts * 1e3))
