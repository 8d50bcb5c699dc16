"""End-to-end time per subject through the sequence path (BASELINE config 1 volume: 192x208x10x50 float32 = 500 slices).

  1. host numpy pre/post-processing + forward_host            (the reference's structure, common/deploy_network.py:86-131)
  2. device pipeline, one subject at a time                     (device_pipeline.segment_sequence_device, round 1)
  3. subject pipeline, steady state, files excluded             (subject_pipeline.SubjectPipeline: pinned staging, 3 streams)
  4. the drop-in script over a cohort of gzip NIfTI files       (reader threads -> GPU -> writer threads), files INCLUDED,
     with --io_threads 0 (sequential subjects) and N

GPU box only.   python tools/bench_subject.py [--subjects 8] [--cohort 12] [--io_threads 4,8,16]"""
import argparse
import functools
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

print = functools.partial(print, flush=True)

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--subjects', type=int, default=10)
    ap.add_argument('--cohort', type=int, default=12)
    ap.add_argument('--io_threads', default='4,8,16')
    ap.add_argument('--skip-host', action='store_true')
    ap.add_argument('--data', choices=['noise', 'phantom'], default='noise',
                    help='noise: float32 gamma noise (inflates slowly, gives noisy label maps: the worst case for the file stages); '
                         'phantom: smooth integer-valued cine phantom (closer to an MR magnitude volume)')
    args = ap.parse_args()
    import torch
    from ukbb_cardiac_amd import deploy_network, device_pipeline as dp, nifti
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.pipeline import segment_sequence
    from ukbb_cardiac_amd.subject_pipeline import SubjectPipeline, labels_as_float64
    from ukbb_cardiac_amd.weights import save_blob, synthetic_params

    arch = MODELS['FCN_sa']
    params = synthetic_params(arch, 1234)
    eng = Engine(arch, params)
    rng = np.random.default_rng(0)
    shape = (192, 208, 10, 50)
    if args.data == 'phantom':
        from ukbb_cardiac_amd.phantom import cine_phantom
        X, Y, Z, T = shape
        vols = [np.asfortranarray(np.round(cine_phantom(Z * T, X, Y, seed=40 + i)[..., 0].reshape(T, Z, X, Y).transpose(2, 3, 1, 0) * 1000.0)
                                  .astype(np.float32)) for i in range(3)]
    else:
        vols = [np.asfortranarray((1000 * rng.gamma(2.0, 1.0, size=shape)).astype(np.float32)) for _ in range(3)]
    n = shape[2] * shape[3]
    print('subject %dx%dx%dx%d (%d slices); host cores: %d logical' % (shape + (n, os.cpu_count())))

    def timeit(f, reps):
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, r

    want = None
    if not args.skip_host:
        fwd = lambda b: eng.run(b, want_prob=False)
        th, want = timeit(lambda: segment_sequence(vols[0].copy(order='F'), fwd, 128), 1)
        print('1. host pre/post-processing + forward_host:            %7.1f ms per subject (%6.0f slices/s)' % (th * 1e3, n / th))
    td, got = timeit(lambda: dp.segment_sequence_device(vols[0], eng, 128), 5)
    print('2. device pipeline, one subject at a time (pageable H2D, float64 volume on this thread): %7.1f ms (%6.0f slices/s)' % (td * 1e3, n / td))
    if want is not None:
        assert np.array_equal(want, got)

    pipe = SubjectPipeline(eng, shape, 128, depth=3)
    first = None
    for r in pipe.run([vols[0]]):
        first = r.labels
    assert np.array_equal(labels_as_float64(first), got), 'pipelined labels differ from the sequential device pipeline'
    K = args.subjects

    def staged_source():                     # what a reader thread leaves behind: the volume already in pinned memory
        for i in range(K):
            st = pipe.stage(shape)
            st.array[...] = vols[i % 3]
            yield st
    # (a) volumes already staged in pinned memory by "readers": stage ahead of the GPU thread is not possible from one
    #     thread beyond the pool size, so time the GPU-side work only: pre-stage lazily inside the generator and subtract
    #     the memcpy by measuring it separately
    t0 = time.perf_counter()
    for _ in pipe.run(staged_source()):
        pass
    torch.cuda.synchronize()
    t_staged = (time.perf_counter() - t0) / K
    def one_copy():
        st = pipe.stage(shape)
        np.copyto(st.array, vols[0])
        pipe._staged.pop(id(st.array), None)
        pipe._in_free.put(st.buf)
    tm, _ = timeit(one_copy, 3)
    print('3. subject pipeline, steady state over %d subjects, files excluded: %7.1f ms per subject (%6.0f slices/s) including a '
          'single-thread 80 MB memcpy into pinned memory of %.1f ms that reader threads do off this thread in the script'
          % (K, t_staged * 1e3, n / t_staged, tm * 1e3))

    # (b) GPU-thread time only: inputs pre-staged before the clock starts (as many as the pool holds), depth 3
    pipe2 = SubjectPipeline(eng, shape, 128, depth=3, extra_inputs=K)
    for _ in pipe2.run([vols[0], vols[1]]):                  # warm-up
        pass
    staged = []
    for i in range(K):
        st = pipe2.stage(shape)
        st.array[...] = vols[i % 3]
        staged.append(st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in pipe2.run(staged):
        pass
    torch.cuda.synchronize()
    t_gpu = (time.perf_counter() - t0) / K
    print('   ... inputs already in pinned memory (what reader threads provide): %7.1f ms per subject (%6.0f slices/s); '
          'network alone: %.1f ms' % (t_gpu * 1e3, n / t_gpu, n / 46000.0 * 1e3))
    del pipe, pipe2

    # 4. files included
    root = tempfile.mkdtemp(prefix='ukbb_cohort_')
    try:
        mp = os.path.join(root, 'FCN_sa')
        save_blob(mp + '.ukbbw', arch, params)
        src = os.path.join(root, 'src')
        os.makedirs(src)
        aff = np.diag([1.8, 1.8, 10.0, 1.0])
        pixdim = np.array([1, 1.8, 1.8, 10.0, 0.03, 0, 0, 0], np.float32)
        t0 = time.perf_counter()
        for i in range(args.cohort):
            os.makedirs(os.path.join(src, 's%03d' % i))
            nifti.save(vols[i % 3], os.path.join(src, 's%03d' % i, 'sa.nii.gz'), aff, pixdim)
        tw = (time.perf_counter() - t0) / args.cohort
        size = os.path.getsize(os.path.join(src, 's000', 'sa.nii.gz')) / 1e6
        t0 = time.perf_counter(); nifti.load(os.path.join(src, 's000', 'sa.nii.gz')); tr = time.perf_counter() - t0
        print('4. cohort of %d subjects on disk: sa.nii.gz %.0f MB each (--data %s; reading one: %.0f ms, writing one: %.0f ms, single thread)'
              % (args.cohort, size, args.data, tr * 1e3, tw * 1e3))
        # where one subject's time goes when nothing overlaps (the stages of the sequential loop, deploy_network.py:80-151)
        w1 = os.path.join(root, 'one')
        shutil.copytree(os.path.join(src, 's000'), os.path.join(w1, 's000'))
        t0 = time.perf_counter(); nim = nifti.load(os.path.join(w1, 's000', 'sa.nii.gz')); t1 = time.perf_counter()
        pred, aux = dp.segment_sequence_device(nim.get_data(), eng, return_aux=True); t2 = time.perf_counter()
        lab8 = pred.astype(np.uint8); t3 = time.perf_counter()
        nifti.save(lab8, os.path.join(w1, 's000', 'seg_sa.nii.gz'), nim.affine, nim.header['pixdim'], as_dtype=np.float64); t4 = time.perf_counter()
        nifti.LABEL_FAST_PATH = False
        nifti.save(lab8, os.path.join(w1, 's000', 'seg_zlib.nii.gz'), nim.affine, nim.header['pixdim'], as_dtype=np.float64); t5 = time.perf_counter()
        nifti.LABEL_FAST_PATH = True
        runs = int(np.count_nonzero(np.diff(lab8.reshape(-1, order='F')))) + 1
        print('   one subject, nothing overlapped: read+inflate %.0f ms | segment (device path, pageable) %.0f ms | write seg_sa: run-length gzip %.0f ms '
              '(%.1f MB, %d runs) vs zlib level 1 %.0f ms (%.1f MB)' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t3) * 1e3,
              os.path.getsize(os.path.join(w1, 's000', 'seg_sa.nii.gz')) / 1e6, runs, (t5 - t4) * 1e3,
              os.path.getsize(os.path.join(w1, 's000', 'seg_zlib.nii.gz')) / 1e6))
        shutil.rmtree(w1)
        for thr in [0] + [int(v) for v in args.io_threads.split(',')]:
            work = os.path.join(root, 'run%d' % thr)
            shutil.copytree(src, work)
            t0 = time.perf_counter()
            deploy_network.run(deploy_network.define_flags().parse(['--seq_name', 'sa', '--data_dir', work, '--model_path', mp,
                                                                     '--io_threads', str(thr)])[0],
                               lambda b: {'pred': eng.run(b, want_prob=False)['pred']}, log=lambda *_: None, engine=eng)
            dt = time.perf_counter() - t0
            out = os.path.getsize(os.path.join(work, 's000', 'seg_sa.nii.gz')) / 1e6
            print('   deploy_network.py --io_threads %-2d: %6.2f s for %d subjects = %5.2f subjects/s (%6.0f slices/s), files included '
                  '(seg_sa.nii.gz: 160 MB of float64 -> %.1f MB gzip)' % (thr, dt, args.cohort, args.cohort / dt, args.cohort * n / dt, out))
            shutil.rmtree(work)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    eng.close()
