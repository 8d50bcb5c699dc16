"""PCIe-inclusive throughput of the host-buffer entry point (ukbb_fcn_forward_host = the
reference's sess.run shape: H2D copy, forward, D2H copy, synchronous).  GPU box only.
    python tools/bench_host_path.py [batch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ukbb_cardiac_amd.arch import MODELS                              # noqa: E402
from ukbb_cardiac_amd.engine import Engine                            # noqa: E402
from ukbb_cardiac_amd.phantom import uniform_slices                   # noqa: E402
from ukbb_cardiac_amd.weights import synthetic_params                 # noqa: E402

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    arch = MODELS['FCN_sa']
    eng = Engine(arch, synthetic_params(arch, 1234))
    x = uniform_slices(n, 192, 208, seed=1)
    for want_prob in (False, True):
        for _ in range(3):
            eng.run(x, want_prob=want_prob)
        t0 = time.perf_counter()
        k = 20
        for _ in range(k):
            eng.run(x, want_prob=want_prob)
        dt = (time.perf_counter() - t0) / k
        print('forward_host N=%d fetch=%s: %.3f ms/call  %.0f slices/s (pageable host memory, includes H2D %d KB + D2H)'
              % (n, 'prob+pred' if want_prob else 'pred', dt * 1e3, n / dt, n * 192 * 208 * 4 // 1024))
    # the reference's own call pattern: 50 frames x batch 10 (deploy_network.py:103-111)
    x10 = x[:10]
    for _ in range(5):
        eng.run(x10)
    t0 = time.perf_counter()
    for _ in range(50):
        eng.run(x10)
    dt = time.perf_counter() - t0
    print('reference call pattern: 50 x sess.run(batch 10, prob+pred): %.1f ms per 500-slice subject  %.0f slices/s'
          % (dt * 1e3, 500 / dt))
