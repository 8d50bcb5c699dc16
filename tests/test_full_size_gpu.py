"""BASELINE configs[0] and configs[3] at their stated size, under ``pytest -m gpu`` (VERDICT r02 "Next round" item 1).

configs[0]: one 192x208x10x50 short-axis volume (SURVEY.md 8(d) config 1 recipe: ``1000 * default_rng(0).gamma(2.0, 1.0)``)
through the drop-in ``deploy_network.py --seq_name sa`` as a reference user would call it, every one of the 500 label slices
against ``oracle/fcn_oracle.c`` driven through the restated loop of common/deploy_network.py:83-131 (``O.deploy_sequence``),
the ES pick, the five files, float64 dtype, affine and pixdim -- the 80 MB-in / 160 MB-float64-out regime of the subject
pipeline, the pinned staging pool and the run-length label writer.

configs[3]: a cohort of full-size subjects through the per-GPU launcher ``python -m ukbb_cardiac_amd.shard`` with 2 and 8
worker processes on the one visible GPU: complete, disjoint, byte-identical to a single process, one merged CSV.
(The 1 -> 8 GPU curve itself needs an 8-GPU node; this is the host-side rehearsal of the same code path.)
"""
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np
import pytest

from oracle import c_oracle, fcn_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = (192, 208, 10, 50)                                  # BASELINE.json configs[0] / configs[3]
NEAR_TIE = 1e-4


def _model(tmp_path, name='FCN_sa'):
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.weights import pack_flat, save_blob, synthetic_params
    arch = MODELS[name]
    params = synthetic_params(arch, 1234)
    mp = str(tmp_path / name)
    save_blob(mp + '.ukbbw', arch, params)
    return arch, params, pack_flat(arch, params), mp


def _record(name, **kw):
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, 'full_size_report.json')
    rec = json.load(open(path)) if os.path.exists(path) else {}
    rec[name] = kw
    with open(path, 'w') as f:
        json.dump(rec, f, indent=1, sort_keys=True)


def test_config0_one_full_size_subject_through_deploy_network(tmp_path):
    from ukbb_cardiac_amd import deploy_network, measures, nifti
    arch, params, flat, mp = _model(tmp_path)
    X, Y, Z, T = SHAPE
    vol = (1000.0 * np.random.default_rng(0).gamma(2.0, 1.0, size=SHAPE)).astype(np.float32)
    d = tmp_path / 'data' / 'subj1'
    d.mkdir(parents=True)
    affine = np.array([[-1.8, 0.2, 0.0, 90.0], [0.2, 1.8, 0.3, -70.0], [0.0, -0.3, 10.0, 15.0], [0, 0, 0, 1]])
    pixdim = np.array([1, 1.8269, 1.8269, 10.0, 0.0305, 0, 0, 0], np.float32)
    nifti.save(vol, str(d / 'sa.nii.gz'), affine, pixdim)
    csv = str(tmp_path / 'sa.csv')
    t0 = time.time()
    deploy_network.main(['--seq_name', 'sa', '--data_dir', str(tmp_path / 'data'), '--model_path', mp, '--output_csv', csv])
    t_deploy = time.time() - t0
    assert sorted(os.listdir(d)) == sorted(['sa.nii.gz', 'seg_sa.nii.gz', 'sa_ED.nii.gz', 'sa_ES.nii.gz', 'seg_sa_ED.nii.gz', 'seg_sa_ES.nii.gz'])
    seg = nifti.load(str(d / 'seg_sa.nii.gz'))
    assert seg.data.dtype == np.float64 and seg.data.shape == SHAPE                                  # deploy_network.py:92,136
    assert np.allclose(seg.affine, affine, atol=1e-5) and np.array_equal(seg.header['pixdim'], pixdim)
    # ---- all 500 slices against the C oracle through the restated loop ----
    t0 = time.time()
    ref_pred, ref_img, ed, es = O.deploy_sequence(vol.copy(), lambda b: c_oracle.forward(arch, flat, b, want_logits=False)[2], 'sa')
    t_oracle = time.time() - t0
    bad = seg.data != ref_pred
    nbad = int(bad.sum())
    assert nbad <= 40 * bad.size // 1000000, '%d label disagreements in %d voxels' % (nbad, bad.size)   # r02: 0-4 per 2.56 M, all ties
    # fp64 arbitration of (up to 12 of) the slices that hold a disagreement: each must sit on a numerical tie
    zt = np.argwhere(bad.any(axis=(0, 1)))
    away = 0
    if len(zt):
        lo, hi = np.percentile(vol, (1, 99))
        for z, t in zt[:12]:
            sl = np.clip(vol[:, :, z, t], lo, hi)
            net_in = ((sl.astype(np.float32).astype(np.float64) - np.float64(lo)) / (np.float64(hi) - np.float64(lo))).astype(np.float32)
            ref64 = O.build_FCN(net_in[None, :, :, None], params, arch.n_class, dtype=np.float64)
            away += int((bad[:, :, z, t] & (O.top2_margin(ref64)[0] > NEAR_TIE)).sum())
    assert away == 0, '%d label disagreements away from a numerical tie' % away
    # ---- ES pick (deploy_network.py:125-131) and the four frame files ----
    got_es_counts = np.sum(seg.data == 1, axis=(0, 1, 2))
    got_es = int(np.argmin(got_es_counts))
    ref_counts = np.sum(ref_pred == 1, axis=(0, 1, 2))
    if np.sort(ref_counts)[1] - np.sort(ref_counts)[0] > nbad:                                       # the pick is not within the tie noise
        assert got_es == es
    for fr, k in (('ED', 0), ('ES', got_es)):
        img_fr = nifti.load(str(d / ('sa_%s.nii.gz' % fr)))
        assert img_fr.data.dtype == np.float32 and np.array_equal(img_fr.data, ref_img[:, :, :, k])  # the CLIPPED frames (App. C.1)
        seg_fr = nifti.load(str(d / ('seg_sa_%s.nii.gz' % fr)))
        assert seg_fr.data.dtype == np.float64 and np.array_equal(seg_fr.data, seg.data[:, :, :, k])
    # ---- --output_csv: eval_ventricular_volume.py's formulas on the written files ----
    from test_host_pipeline import _eval_ventricular_row, _pandas_csv
    want = _eval_ventricular_row(str(d / 'sa.nii.gz'), str(d / 'seg_sa.nii.gz'))
    assert open(csv).read() == _pandas_csv(str(tmp_path / 'pd.csv'), [want], ['subj1'], measures.SA_COLUMNS)
    _record('config0_subject', shape=list(SHAPE), label_disagreements=nbad, voxels=int(bad.size), away_from_tie=away,
            slices_with_disagreement=int(len(zt)), es_frame=got_es, oracle_es_frame=int(es), deploy_seconds_incl_engine_start=round(t_deploy, 2),
            c_oracle_seconds=round(t_oracle, 2), seg_file_bytes=os.path.getsize(str(d / 'seg_sa.nii.gz')))


def _run(cmd, timeout=1500):
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    return subprocess.run([sys.executable] + cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)


def test_config3_full_size_cohort_through_the_shard_launcher(tmp_path):
    from ukbb_cardiac_amd import nifti
    from ukbb_cardiac_amd.phantom import cine_phantom
    arch, params, flat, mp = _model(tmp_path)
    X, Y, Z, T = SHAPE
    n_subj = 8
    src = tmp_path / 'cohort'
    src.mkdir()
    aff = np.diag([1.8269, 1.8269, 10.0, 1.0])
    pixdim = np.array([1, 1.8269, 1.8269, 10.0, 0.0305, 0, 0, 0], np.float32)
    for i in range(n_subj):                                   # structured cines (compress like MR data; noise would cost 4 s each to gzip)
        p = cine_phantom(Z * T, X, Y, seed=100 + i)[..., 0]
        vol = np.round(p.reshape(T, Z, X, Y).transpose(2, 3, 1, 0) * 1000.0).astype(np.float32)
        (src / ('subj%02d' % i)).mkdir()
        nifti.save(vol, str(src / ('subj%02d' % i) / 'sa.nii.gz'), aff, pixdim)
    script = os.path.join(ROOT, 'ukbb_cardiac_amd', 'deploy_network.py')
    flags = ['--seq_name', 'sa', '--model_path', mp]
    names = ['seg_sa.nii.gz', 'sa_ED.nii.gz', 'sa_ES.nii.gz', 'seg_sa_ED.nii.gz', 'seg_sa_ES.nii.gz']
    runs, rates, phases = {}, {}, {}
    for tag, shards in (('single', 0), ('shards2', 2), ('shards8', 8)):
        work = tmp_path / tag
        shutil.copytree(str(src), str(work))
        csv = str(tmp_path / (tag + '.csv'))
        extra = ['--data_dir', str(work), '--output_csv', csv] + (['--io_threads', '2'] if shards == 8 else [])
        t0 = time.time()
        if shards:
            r = _run(['-m', 'ukbb_cardiac_amd.shard', '--gpus', '1', '--shards_per_gpu', str(shards), '--', script] + flags + extra)
        else:
            r = _run([script] + flags + extra)
        dt = time.time() - t0
        assert r.returncode == 0, r.stdout[-3000:]
        assert r.stdout.count('Segmentation time') == n_subj                                          # complete and disjoint
        runs[tag] = {(i, nm): (work / ('subj%02d' % i) / nm).read_bytes() for i in range(n_subj) for nm in names}
        runs[tag]['csv'] = open(csv).read()
        assert not [f for f in os.listdir(tmp_path) if '.shard' in f]                                 # parts merged
        rates[tag] = round(n_subj * Z * T / dt, 1)
        # phases: the launcher's wall clock against what the workers' own loops report
        took = [float(l.split('it took')[1].split('s for')[0]) for l in r.stdout.splitlines() if 'it took' in l]
        segt = [float(l.split('=')[1].strip().rstrip('s')) for l in r.stdout.splitlines() if 'Segmentation time' in l]
        phases[tag] = {'wall_s': round(dt, 2), 'worker_loops_s': [round(v, 3) for v in took], 'slowest_worker_loop_s': round(max(took), 3) if took else None,
                       'outside_the_loops_s': round(dt - (max(took) if took else 0.0), 2),      # interpreter + torch + engine start, CSV merge
                       'sum_of_segmentation_times_s': round(sum(segt), 3)}
    assert runs['single'] == runs['shards2'] == runs['shards8']
    assert len(runs['single']['csv'].splitlines()) == n_subj + 1
    # one subject of the cohort against the C oracle at full size (the rest are byte-compared above)
    seg = nifti.load(str(tmp_path / 'single' / 'subj03' / 'seg_sa.nii.gz')).get_data()
    vol = nifti.load(str(src / 'subj03' / 'sa.nii.gz')).get_data()
    ref_pred, _, _, es = O.deploy_sequence(vol.copy(), lambda b: c_oracle.forward(arch, flat, b, want_logits=False)[2], 'sa')
    nbad = int((seg != ref_pred).sum())
    assert nbad <= 40 * seg.size // 1000000, nbad
    assert len(np.unique(seg)) == 4
    _record('config3_cohort', subjects=n_subj, shape=list(SHAPE), label_disagreements_subj03=nbad,
            slices_per_s_wall_incl_process_start={k: v for k, v in rates.items()}, phases=phases,
            note='one GPU; wall clock of the whole launcher incl. interpreter / torch / engine start of every worker and gzip NIfTI I/O; '
                 'r03 recorded 52 slices/s for the single-process and 2-shard runs of this test on one box (76 s each) against 1234 for 8 shards: '
                 'tools/cohort_probe.py (profiles/r04_cohort_probe.txt) runs the same 8 subjects at 1.9-2.0 k slices/s single-process with --io_threads 8 '
                 'and --output_csv (2.1 s wall, 0.6 s inside the loop); the 76 s did not reproduce on any r04 box and were outside the pipeline')
    # a single process must not be slower than eight workers sharing the GPU by more than their start-up overlap explains
    assert phases['single']['wall_s'] <= 3.0 * phases['shards8']['wall_s'] + 5.0, phases
