"""Where the host time of the drop-in deploy loop goes: cProfile of `deploy_network.py --io_threads 0` (strictly sequential subjects, so
every stage is attributed to its caller) over a small cohort of full-size phantom subjects with compact label regions (GPU box).
    python tools/profile_host_loop.py [subjects=12] [pipelined]     (pipelined: the default threaded path, stage timers instead of cProfile)"""
import cProfile
import io
import os
import pstats
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == '__main__':
    from ukbb_cardiac_amd import deploy_network, nifti
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import save_blob, threshold_params
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    X, Y, Z, T = 192, 208, 10, 50
    work = tempfile.mkdtemp(prefix='ukbb_hostprof_')
    arch = MODELS['FCN_sa']
    mp = os.path.join(work, 'FCN_sa')
    save_blob(mp + '.ukbbw', arch, threshold_params(arch))
    data = os.path.join(work, 'data')
    aff = np.diag([1.8269, 1.8269, 10.0, 1.0])
    for i in range(n):
        p = cine_phantom(Z * T, X, Y, seed=1000 + i % 4)[..., 0]
        vol = np.round(p.reshape(T, Z, X, Y).transpose(2, 3, 1, 0) * 1000.0).astype(np.float32)
        os.makedirs(os.path.join(data, 'subj%03d' % i))
        nifti.save(vol, os.path.join(data, 'subj%03d' % i, 'sa.nii.gz'), aff)
    import threading
    import time
    if len(sys.argv) > 2 and sys.argv[2] == 'pipelined':
        # the default path (reader / writer threads around the GPU thread): per-call thread CPU time of the stages, by wrapping them
        from ukbb_cardiac_amd import subject_pipeline
        acc, lock = {}, threading.Lock()

        def timed(name, fn):
            def w(*a, **k):
                t0, c0 = time.perf_counter(), time.thread_time()
                try:
                    return fn(*a, **k)
                finally:
                    with lock:
                        e = acc.setdefault(name, [0, 0.0, 0.0])
                        e[0] += 1; e[1] += time.perf_counter() - t0; e[2] += time.thread_time() - c0
            return w
        nifti.load = timed('nifti.load', nifti.load)
        nifti.save = timed('nifti.save', nifti.save)
        nifti._as_label_volume = timed('  nifti._as_label_volume', nifti._as_label_volume)
        nifti._save_labels_gz = timed('  nifti._save_labels_gz', nifti._save_labels_gz)
        subject_pipeline.SubjectPipeline.submit = timed('pipeline.submit', subject_pipeline.SubjectPipeline.submit)
        subject_pipeline.SubjectPipeline.collect = timed('pipeline.collect', subject_pipeline.SubjectPipeline.collect)
        t0 = time.perf_counter()
        deploy_network.main(['--seq_name', 'sa', '--model_path', mp, '--data_dir', data, '--io_threads', '2'])
        wall = time.perf_counter() - t0
        print('pipelined path, --io_threads 2, %d subjects, %.2f s wall; per subject (calls, wall ms, thread CPU ms):' % (n, wall))
        for k, (c, w_, cpu) in acc.items():
            print('  %-28s %5.1f calls %8.1f ms wall %8.1f ms cpu' % (k, c / n, 1e3 * w_ / n, 1e3 * cpu / n))
    else:
        argv = ['--seq_name', 'sa', '--model_path', mp, '--data_dir', data, '--io_threads', '0']
        pr = cProfile.Profile()
        pr.enable()
        deploy_network.main(argv)
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
        print('\n'.join(l[:170] for l in s.getvalue().splitlines()))
    shutil.rmtree(work, ignore_errors=True)
