"""Host-side scaling rehearsal of BASELINE config 4's code path on ONE GPU: a cohort of full-size
192x208x10x50 subjects through `python -m ukbb_cardiac_amd.shard --gpus 1 --shards_per_gpu S -- deploy_network.py ...` for
S = 1, 2, 4, 8 worker processes sharing the device, gzip NIfTI in and out.  What it shows: the launcher / sharding / file
pipeline scale until the one GPU (or the box's cores) saturate -- NOT the 1 -> 8 GPU curve, which needs an 8-GPU node.
    python tools/shard_rehearsal.py [unique subjects=16] [copies=16] [threshold|random] [io threads=8]   (copies are hard links: 256 subjects = 128 000 slices by default)
Weights: 'threshold' (default, weights.threshold_params: compact label regions, the output statistic of a trained model -- the label
writer then costs what it costs on real segmentations) or 'random' (synthetic_params: noise-like label maps, the writer's worst case and
what r03 / the first r04 run measured)."""
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == '__main__':
    from ukbb_cardiac_amd import nifti
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import save_blob, synthetic_params, threshold_params
    n_uniq = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    copies = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    n_subj = n_uniq * copies
    kind = sys.argv[3] if len(sys.argv) > 3 else 'threshold'
    io_total = int(sys.argv[4]) if len(sys.argv) > 4 else 8      # reader / writer threads over all workers of the GPU
    X, Y, Z, T = 192, 208, 10, 50
    work = tempfile.mkdtemp(prefix='ukbb_rehearsal_')
    arch = MODELS['FCN_sa']
    mp = os.path.join(work, 'FCN_sa')
    save_blob(mp + '.ukbbw', arch, threshold_params(arch) if kind == 'threshold' else synthetic_params(arch, 1234))
    src = os.path.join(work, 'src')
    os.makedirs(src)
    aff = np.diag([1.8269, 1.8269, 10.0, 1.0])
    pixdim = np.array([1, 1.8269, 1.8269, 10.0, 0.0305, 0, 0, 0], np.float32)
    t0 = time.time()
    for i in range(n_uniq):
        p = cine_phantom(Z * T, X, Y, seed=1000 + i)[..., 0]
        vol = np.round(p.reshape(T, Z, X, Y).transpose(2, 3, 1, 0) * 1000.0).astype(np.float32)
        os.makedirs(os.path.join(src, 'subj%03d' % i))
        nifti.save(vol, os.path.join(src, 'subj%03d' % i, 'sa.nii.gz'), aff, pixdim)
    print('weights: %s, %d I/O threads per GPU' % (kind, io_total))
    print('cohort: %d subjects (%d distinct x %d hard links) of %dx%dx%dx%d (%d slices each), %.1f MB of gzip NIfTI per pass over the cohort, '
          'generated in %.0f s; host: %d logical cores' % (
              n_subj, n_uniq, copies, X, Y, Z, T, Z * T, copies * sum(os.path.getsize(os.path.join(src, d, 'sa.nii.gz')) for d in os.listdir(src)) / 1e6,
              time.time() - t0, os.cpu_count()), flush=True)
    script = os.path.join(ROOT, 'ukbb_cardiac_amd', 'deploy_network.py')
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    base = None
    for shards in (1, 2, 4, 8):
        run = os.path.join(work, 'run%d' % shards)
        os.makedirs(run)
        for c in range(copies):                                 # hard links: the inputs are read-only
            for d in sorted(os.listdir(src)):
                os.makedirs(os.path.join(run, '%s_%02d' % (d, c)))
                os.link(os.path.join(src, d, 'sa.nii.gz'), os.path.join(run, '%s_%02d' % (d, c), 'sa.nii.gz'))
        cmd = [sys.executable, '-m', 'ukbb_cardiac_amd.shard', '--gpus', '1', '--shards_per_gpu', str(shards), '--', script,
               '--seq_name', 'sa', '--model_path', mp, '--data_dir', run, '--io_threads', str(max(2, io_total // shards))]
        t0 = time.time()
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        dt = time.time() - t0
        done = r.stdout.count('Segmentation time')
        inner = [float(l.split('it took')[1].split('s for')[0]) for l in r.stdout.splitlines() if 'it took' in l]
        rate = n_subj * Z * T / dt
        base = base or rate
        print('shards %d: rc %d, %d subjects done, wall %.1f s incl. worker start-up = %.0f slices/s (%.2fx of 1 shard); slowest worker loop %.1f s = %.0f slices/s' % (
            shards, r.returncode, done, dt, rate, rate / base, max(inner) if inner else float('nan'),
            n_subj * Z * T / max(inner) if inner else float('nan')), flush=True)
        shutil.rmtree(run)
    shutil.rmtree(work)
