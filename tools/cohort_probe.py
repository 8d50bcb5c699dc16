"""Where does a small cohort's wall time go?  8 full-size short-axis subjects (192x208x10x50) through the drop-in deploy_network.py as a
child process, for several --io_threads settings, with the phases timed from outside (GPU box):
    python tools/cohort_probe.py [n_subjects]
Written for VERDICT r03 'What's weak #8': r03_full_size_report.json showed 52 slices/s for the single-process and 2-shard runs
against 1234 for 8 shards of the same 8 subjects."""
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ukbb_cardiac_amd import nifti                                    # noqa: E402
from ukbb_cardiac_amd.arch import MODELS                              # noqa: E402
from ukbb_cardiac_amd.phantom import cine_phantom                     # noqa: E402
from ukbb_cardiac_amd.weights import save_blob, synthetic_params      # noqa: E402

if __name__ == '__main__':
    n_subj = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    X, Y, Z, T = 192, 208, 10, 50
    tmp = tempfile.mkdtemp(prefix='cohort_probe_')
    arch = MODELS['FCN_sa']
    mp = os.path.join(tmp, 'FCN_sa.ukbbw')
    save_blob(mp, arch, synthetic_params(arch, 1234))
    src = os.path.join(tmp, 'cohort')
    os.mkdir(src)
    aff = np.diag([1.8269, 1.8269, 10.0, 1.0])
    pixdim = np.array([1, 1.8269, 1.8269, 10.0, 0.0305, 0, 0, 0], np.float32)
    t0 = time.time()
    for i in range(n_subj):
        p = cine_phantom(Z * T, X, Y, seed=100 + i)[..., 0]
        vol = np.round(p.reshape(T, Z, X, Y).transpose(2, 3, 1, 0) * 1000.0).astype(np.float32)
        os.mkdir(os.path.join(src, 'subj%02d' % i))
        nifti.save(vol, os.path.join(src, 'subj%02d' % i, 'sa.nii.gz'), aff, pixdim)
    print('cohort of %d subjects written in %.1f s' % (n_subj, time.time() - t0), flush=True)
    script = os.path.join(ROOT, 'ukbb_cardiac_amd', 'deploy_network.py')
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    for tag, extra in (('io8+csv', ['--io_threads', '8', '--output_csv', 'X']), ('io8', ['--io_threads', '8']), ('io2+csv', ['--io_threads', '2', '--output_csv', 'X']),
                       ('io0', ['--io_threads', '0']), ('io8+csv again', ['--io_threads', '8', '--output_csv', 'X'])):
        work = os.path.join(tmp, 'w_' + tag.replace('+', '_').replace(' ', '_'))
        shutil.copytree(src, work)
        extra = [os.path.join(tmp, tag.replace(' ', '_') + '.csv') if e == 'X' else e for e in extra]
        t0 = time.time()
        r = subprocess.run([sys.executable, '-X', 'importtime', script, '--seq_name', 'sa', '--model_path', mp, '--data_dir', work] + extra, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        dt = time.time() - t0
        took = [l for l in r.stdout.splitlines() if 'it took' in l]
        segt = [float(l.split('=')[1].strip().rstrip('s')) for l in r.stdout.splitlines() if 'Segmentation time' in l]
        # the slowest imports of the child (importtime prints cumulative microseconds in the second column)
        imps = []
        for l in r.stderr.splitlines():
            if l.startswith('import time:') and '|' in l:
                parts = l.split('|')
                try:
                    imps.append((int(parts[1]), parts[2].strip()))
                except ValueError:
                    pass
        imps.sort(reverse=True)
        print('%-14s rc %d wall %.1f s -> %.0f slices/s; script: %s; per-subject segmentation times %s; slowest imports %s' % (
            tag, r.returncode, dt, n_subj * Z * T / dt, took[-1].strip() if took else '-', ' '.join('%.2f' % v for v in segt),
            ', '.join('%s %.1f s' % (n, us / 1e6) for us, n in imps[:3])), flush=True)
        shutil.rmtree(work)
    shutil.rmtree(tmp)
