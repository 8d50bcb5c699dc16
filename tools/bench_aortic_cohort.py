"""Aortic cohort through the drop-in script (deploy_network_ao.py, default UNet-LSTM model), gzip NIfTI files included:
subjects/s for --io_threads 0 (the reference's strictly sequential loop) and with read-ahead / write-behind threads.
GPU box only.   python tools/bench_aortic_cohort.py [--cohort 24] [--io_threads 2,4,8]"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cohort', type=int, default=24)
    ap.add_argument('--io_threads', default='2,4,8')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'])
    args = ap.parse_args()
    from ukbb_cardiac_amd import deploy_network_ao, nifti
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import save_blob, synthetic_params
    arch = MODELS['UNet-LSTM_ao']
    params = synthetic_params(arch, 1234)
    eng = Engine(arch, params)
    if args.precision != 'fp32':
        eng.set_precision(args.precision)
    X, Y, T = 240, 196, 100
    vols = [np.asfortranarray(np.round(cine_phantom(T, X, Y, seed=70 + i)[..., 0].transpose(1, 2, 0)[:, :, None, :] * 1000.0).astype(np.float32))
            for i in range(3)]
    root = tempfile.mkdtemp(prefix='ukbb_ao_')
    try:
        mp = os.path.join(root, 'UNet-LSTM_ao')
        save_blob(mp + '.ukbbw', arch, params)
        src = os.path.join(root, 'src')
        for i in range(args.cohort):
            os.makedirs(os.path.join(src, 's%03d' % i))
            nifti.save(vols[i % 3], os.path.join(src, 's%03d' % i, 'ao.nii.gz'), np.diag([1.6, 1.6, 6.0, 1.0]),
                       pixdim=[1, 1.6, 1.6, 6, 0.01, 0, 0, 0])
        print('aortic cohort (%s): %d subjects of %dx%dx1x%d float32, ao.nii.gz %.1f MB each' %
              (args.precision, args.cohort, X, Y, T, os.path.getsize(os.path.join(src, 's000', 'ao.nii.gz')) / 1e6), flush=True)
        cine = lambda f, R, r, ts=1: eng.run_cine(f, R, r, ts)[0]
        for thr in [0] + [int(v) for v in args.io_threads.split(',')]:
            work = os.path.join(root, 'run%d' % thr)
            shutil.copytree(src, work)
            flags = deploy_network_ao.define_flags().parse(['--data_dir', work, '--model_path', mp, '--io_threads', str(thr)])[0]
            t0 = time.perf_counter()
            deploy_network_ao.run(flags, None, log=lambda *_: None, cine_forward=cine, engine=eng)
            dt = time.perf_counter() - t0
            print('   deploy_network_ao.py --io_threads %-2d: %6.2f s = %5.2f subjects/s (%5.0f frames/s), files included' %
                  (thr, dt, args.cohort / dt, args.cohort * T / dt), flush=True)
            shutil.rmtree(work)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    eng.close()
