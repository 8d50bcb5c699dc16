"""The drop-in scripts on the real engine (no stub forward): ED/ES mode of both deploy scripts against the
oracle-driven restatement of the same loops, the per-GPU launcher with two worker processes on the one visible
GPU (BASELINE config 4's path), and INTEGRATION.md's HipSession stub executed verbatim."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import c_oracle, fcn_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(tmp_path, name):
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.weights import pack_flat, save_blob, synthetic_params
    arch = MODELS[name]
    params = synthetic_params(arch, 1234)
    mp = str(tmp_path / name)
    save_blob(mp + '.ukbbw', arch, params)
    return arch, params, pack_flat(arch, params), mp


def _volume(shape, seed):
    return (1000.0 * np.random.default_rng(seed).gamma(2.0, 1.0, size=shape)).astype(np.float32)


def _labels_match(got, want, max_ties=2):
    """Two fp32 evaluations of the same graph (HIP engine, C oracle) may differ only at numerical ties of the top two
    logits: a handful of isolated pixels at most."""
    assert got.shape == want.shape
    assert int((got != want).sum()) <= max_ties, '%d label mismatches' % int((got != want).sum())


# ---- ED/ES mode, common/deploy_network.py:152-216 -----------------------------------------------------
@pytest.mark.parametrize('seq,model,seg4,shape', [('sa', 'FCN_sa', False, (162, 204, 3)), ('la_4ch', 'FCN_la_4ch_seg4', True, (150, 171)),
                                                 ('la_2ch', 'FCN_la_2ch', False, (160, 208, 1))])
def test_ed_es_mode_on_engine(tmp_path, seq, model, seg4, shape):
    from ukbb_cardiac_amd import deploy_network, nifti
    arch, params, flat, mp = _model(tmp_path, model)
    d = tmp_path / 'data' / 'subj1'
    d.mkdir(parents=True)
    aff = np.diag([1.8, 1.8, 10.0, 1.0])
    pixdim = np.array([1, 1.8, 1.8, 10.0, 1, 0, 0, 0], np.float32)
    vols = {fr: _volume(shape, seed) for fr, seed in (('ED', 11), ('ES', 12))}
    for fr, v in vols.items():
        nifti.save(v, str(d / ('%s_%s.nii.gz' % (seq, fr))), aff, pixdim)
    (tmp_path / 'data' / 'subj0_incomplete').mkdir()
    nifti.save(vols['ED'], str(tmp_path / 'data' / 'subj0_incomplete' / ('%s_ED.nii.gz' % seq)), aff)   # ES missing -> skipped (:156-161)
    argv = ['--seq_name', seq, '--data_dir', str(tmp_path / 'data'), '--model_path', mp, '--noprocess_seq'] + (['--seg4'] if seg4 else [])
    deploy_network.main(argv)
    pre = 'seg4' if seg4 else 'seg'
    assert not os.path.exists(str(tmp_path / 'data' / 'subj0_incomplete' / ('%s_%s_ED.nii.gz' % (pre, seq))))
    assert not os.path.exists(str(d / ('%s_%s.nii.gz' % (pre, seq))))           # no sequence output in this mode
    for fr, v in vols.items():
        seg = nifti.load(str(d / ('%s_%s_%s.nii.gz' % (pre, seq, fr))))
        want = O.deploy_frame(v.copy(), lambda b: c_oracle.forward(arch, flat, b, want_logits=False)[2])
        assert seg.data.dtype == np.int32                                        # TF's int32 straight to disk (:199-216)
        assert seg.data.shape == (shape if len(shape) == 3 else shape + (1,))
        _labels_match(seg.data, want)
        assert np.array_equal(seg.header['pixdim'], pixdim) and np.allclose(seg.affine, aff)
        assert len(np.unique(seg.data)) > 1


# ---- aortic ED/ES mode, common/deploy_network_ao.py:201-269 (frame-wise U-Net) -------------------------
def test_aortic_unet_ed_es_mode_on_engine(tmp_path):
    from ukbb_cardiac_amd import deploy_network_ao, nifti
    arch, params, flat, mp = _model(tmp_path, 'UNet_ao')
    d = tmp_path / 'data' / 'a1'
    d.mkdir(parents=True)
    aff = np.diag([1.6, 1.6, 6.0, 1.0])
    pixdim = np.array([1, 1.6, 1.6, 6.0, 1, 0, 0, 0], np.float32)
    vols = {fr: _volume((170, 150, 1), seed) for fr, seed in (('ED', 21), ('ES', 22))}
    for fr, v in vols.items():
        nifti.save(v, str(d / ('ao_%s.nii.gz' % fr)), aff, pixdim)
    deploy_network_ao.main(['--seq_name', 'ao', '--data_dir', str(tmp_path / 'data'), '--model_path', mp, '--model', 'UNet',
                            '--noprocess_seq'])
    for fr, v in vols.items():
        seg = nifti.load(str(d / ('seg_ao_%s.nii.gz' % fr)))
        want = O.aortic_deploy_frame(v.copy(), lambda b: c_oracle.forward(arch, flat, b, want_logits=False)[2])
        assert seg.data.dtype == np.int32 and seg.data.shape == (170, 150, 1)    # padded to x16 (176x160), not to 256 (:240-243)
        _labels_match(seg.data, want)
        assert np.array_equal(seg.header['pixdim'], pixdim)
    # the rescale (no z-score) branch, :235-236
    for fr in ('ED', 'ES'):
        os.remove(str(d / ('seg_ao_%s.nii.gz' % fr)))
    deploy_network_ao.main(['--seq_name', 'ao', '--data_dir', str(tmp_path / 'data'), '--model_path', mp, '--model', 'UNet',
                            '--noprocess_seq', '--noz_score'])
    seg = nifti.load(str(d / 'seg_ao_ED.nii.gz'))
    want = O.aortic_deploy_frame(vols['ED'].copy(), lambda b: c_oracle.forward(arch, flat, b, want_logits=False)[2], z_score=False)
    _labels_match(seg.data, want)


# ---- BASELINE config 4's path: the per-GPU launcher, two worker processes on the one visible GPU -------
def _run(cmd, **kw):
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    return subprocess.run([sys.executable] + cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          timeout=600, **kw)


def test_two_shards_on_one_gpu_equal_single_process(tmp_path):
    """`python -m ukbb_cardiac_amd.shard --gpus 1 --shards_per_gpu 2 -- deploy_network.py ...` over 8 synthetic
    subjects: every subject segmented exactly once, outputs byte-identical to a single-process run over a copy of
    the cohort, and a rerun finds everything done (deploy_network.py:62-67)."""
    import shutil
    from ukbb_cardiac_amd import nifti
    arch, params, flat, mp = _model(tmp_path, 'FCN_sa')
    a, b = tmp_path / 'cohort_sharded', tmp_path / 'cohort_single'
    a.mkdir()
    aff = np.diag([1.8, 1.8, 10.0, 1.0])
    pixdim = np.array([1, 1.8, 1.8, 10.0, 0.03, 0, 0, 0], np.float32)
    for i in range(8):
        (a / ('subj%02d' % i)).mkdir()
        nifti.save(_volume((100 + 4 * (i % 3), 120, 3, 5), 50 + i), str(a / ('subj%02d' % i) / 'sa.nii.gz'), aff, pixdim)
    shutil.copytree(str(a), str(b))
    script = os.path.join(ROOT, 'ukbb_cardiac_amd', 'deploy_network.py')
    flags = ['--seq_name', 'sa', '--model_path', mp]
    r = _run(['-m', 'ukbb_cardiac_amd.shard', '--gpus', '1', '--shards_per_gpu', '2', '--', script] + flags + ['--data_dir', str(a)])
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count('Segmenting full sequence') == 8                       # each subject once, over both workers
    s = _run([script] + flags + ['--data_dir', str(b)])
    assert s.returncode == 0, s.stdout[-3000:]
    names = ['seg_sa.nii.gz', 'sa_ED.nii.gz', 'sa_ES.nii.gz', 'seg_sa_ED.nii.gz', 'seg_sa_ES.nii.gz']
    for i in range(8):
        for nm in names:
            pa, pb = a / ('subj%02d' % i) / nm, b / ('subj%02d' % i) / nm
            assert pa.exists() and pa.read_bytes() == pb.read_bytes(), (i, nm)
    # the reader -> GPU -> writer pipeline (default --io_threads 4) writes the same bytes as strictly sequential subjects
    c = tmp_path / 'cohort_sequential'
    shutil.copytree(str(tmp_path / 'cohort_single'), str(c), ignore=shutil.ignore_patterns('seg*', 'sa_E*'))
    q = _run([script] + flags + ['--data_dir', str(c), '--io_threads', '0'])
    assert q.returncode == 0, q.stdout[-3000:]
    for i in range(8):
        for nm in names:
            assert (c / ('subj%02d' % i) / nm).read_bytes() == (b / ('subj%02d' % i) / nm).read_bytes(), (i, nm)
    stamp = {str(p): os.stat(str(p)).st_mtime_ns for p in a.rglob('*.nii.gz')}
    r2 = _run(['-m', 'ukbb_cardiac_amd.shard', '--gpus', '1', '--shards_per_gpu', '2', '--', script] + flags + ['--data_dir', str(a)])
    assert r2.returncode == 0 and 'Segmenting' not in r2.stdout                  # rerun is a no-op
    assert stamp == {str(p): os.stat(str(p)).st_mtime_ns for p in a.rglob('*.nii.gz')}
    # a worker that dies must fail the launcher: shard 1 is pointed at a model file that does not exist
    (a / 'subj01' / 'seg_sa.nii.gz').unlink()
    bad = _run(['-m', 'ukbb_cardiac_amd.shard', '--gpus', '1', '--shards_per_gpu', '2', '--', script, '--seq_name', 'sa',
                '--model_path', str(tmp_path / 'missing_model'), '--data_dir', str(a)])
    assert bad.returncode != 0 and 'shard' in bad.stdout


# ---- the stub a reference maintainer would paste, executed as written ---------
def test_integration_md_hipsession_snippet(tmp_path):
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, re.S)
    stub = [b for b in blocks if 'class HipSession' in b]
    assert len(stub) == 1
    ns = {}
    exec(compile(stub[0], 'INTEGRATION.md', 'exec'), ns)                          # verbatim
    arch, params, flat, mp = _model(tmp_path, 'FCN_sa')
    from ukbb_cardiac_amd import _lib
    from ukbb_cardiac_amd.phantom import cine_phantom
    img = cine_phantom(3, 64, 80, seed=5)
    with ns['HipSession'](mp, lib=_lib.LIB_PATH) as sess:                         # the reference's call, deploy_network.py:110-111
        prob, pred = sess.run(['prob:0', 'pred:0'], feed_dict={'image:0': img, 'training:0': False})
        only = sess.run('pred:0', feed_dict={'image:0': img, 'training:0': False})
    lg, pr, pd = c_oracle.forward(arch, flat, img, want_prob=True)
    assert pred.dtype == np.int32 and pred.shape == (3, 64, 80) and prob.shape == (3, 64, 80, 4)
    assert np.array_equal(only, pred)
    _labels_match(pred, pd)
    assert np.abs(prob - pr).max() <= 1e-4
    with pytest.raises(RuntimeError):
        with ns['HipSession'](mp, lib=_lib.LIB_PATH) as sess:
            sess.run('pred:0', feed_dict={'image:0': np.zeros((1, 30, 32, 1), np.float32)})   # not a multiple of 16


# ---- the N > 1 bench launch, rehearsed on the one visible GPU ---------------------------------------------------
def test_bench_two_rank_launch_rehearsal():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` exactly as the driver launches it.  With one
    GPU both ranks share it (gloo barrier instead of RCCL, flagged in the line): what is checked is the launch contract --
    env-driven ranks, one JSON line from rank 0 only, n_gpus = 2, value = slices of BOTH ranks over the max-over-ranks time."""
    import json
    port = 29600 + os.getpid() % 300
    cmd = ['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
              os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2', '--no-cpu-baseline', '--cohort-subjects', '24']
    import torch
    one_gpu = torch.cuda.device_count() < 2
    r = _run(cmd + (['--rehearsal'] if one_gpu else []))
    assert r.returncode == 0, r.stdout[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 4 and j['scaling'] == 'weak' and j['config']['slices_per_gpu_per_step'] == 64
    assert abs(j['value'] - 2 * 64 * 4 / (j['ms_per_step'] * 4e-3)) <= 0.01 * j['value']
    assert 'cpu_baseline' not in j                        # rank 0 at N = 1 only
    # r06: BASELINE configs[3] rides along on every rank -- subject i on rank i mod 2, no collective, the slowest rank's time counts
    c4 = j['other_configs']['config4_cohort']
    assert 'error' not in c4, c4
    assert c4['n_gpus'] == 2 and [q['subjects_this_rank'] for q in c4['per_rank']] == [12, 12]
    assert abs(c4['value'] - 24 * 500 / c4['seconds_max_over_ranks']) <= 0.01 * c4['value']
    assert c4['seconds_max_over_ranks'] == max(q['seconds_this_rank'] for q in c4['per_rank'])
    # r04: per-rank evidence -- every rank's device, bus id and own time; the headline time is the slowest rank's
    assert [q['rank'] for q in j['ranks']] == [0, 1] and len(j['per_rank_ms_per_step']) == 2
    assert all(q['device_name'] and q['visible_devices'] >= 1 and q['ms_per_step'] > 0 for q in j['ranks'])
    assert max(j['per_rank_ms_per_step']) <= j['ms_per_step'] * 1.001
    assert len({q['pid'] for q in j['ranks']}) == 2
    if one_gpu:
        assert 'REHEARSAL' in j['config']['parallelism'] and j['backend'] == 'gloo'
        assert j['ranks'][0]['pci_bus_id'] == j['ranks'][1]['pci_bus_id']
        # without the flag two ranks on one GPU are refused: a scaling line cannot come from shared devices by accident
        refused = _run(cmd)
        assert refused.returncode != 0 and '--rehearsal' in refused.stdout
    else:
        assert j['backend'].startswith('nccl') and len({q['pci_bus_id'] for q in j['ranks']}) == 2
    # a mismatch between --gpus and the launched world size is refused
    bad = _run([os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'])
    assert bad.returncode != 0 and 'torch.distributed.run' in bad.stdout


def test_precision_flag_f32x3_writes_the_fp32_labels(tmp_path):
    """deploy_network.py --precision f32x3 (UKBB_PREC_F32X3): a phantom cine through both arithmetic modes of the drop-in script;
    the label volumes may differ only at numerical ties (here: not at all or in a handful of voxels)."""
    import shutil
    from ukbb_cardiac_amd import deploy_network, nifti
    from ukbb_cardiac_amd.phantom import cine_phantom
    arch, params, flat, mp = _model(tmp_path, 'FCN_sa')
    X, Y, Z, T = 150, 170, 3, 6
    vol = np.round(cine_phantom(Z * T, X, Y, seed=8)[..., 0].reshape(T, Z, X, Y).transpose(2, 3, 1, 0) * 1000.0).astype(np.float32)
    src = tmp_path / 'src' / 'subj1'
    src.mkdir(parents=True)
    nifti.save(vol, str(src / 'sa.nii.gz'), np.diag([1.8, 1.8, 10.0, 1.0]), pixdim=[1, 1.8, 1.8, 10, 0.03, 0, 0, 0])
    seg = {}
    for prec in ('fp32', 'f32x3'):
        work = tmp_path / prec
        shutil.copytree(tmp_path / 'src', work)
        deploy_network.main(['--seq_name', 'sa', '--data_dir', str(work), '--model_path', mp, '--precision', prec])
        seg[prec] = nifti.load(str(work / 'subj1' / 'seg_sa.nii.gz')).get_data()
    assert seg['fp32'].shape == vol.shape and seg['fp32'].dtype == np.float64
    assert int((seg['fp32'] != seg['f32x3']).sum()) <= 3
    assert len(np.unique(seg['f32x3'])) > 1


# ---- r03: aortic script with --output_csv and --precision bf16 on the engine -----------------------------------------------
def test_aortic_unet_sequence_csv_and_bf16_precision(tmp_path):
    """deploy_network_ao.py --model UNet in sequence mode on the device path (z-score, pack, forward, unpack on the GPU):
    --output_csv equals aortic/eval_aortic_area.py:60-95 applied to the written seg_ao.nii.gz (as pandas writes it), and
    --precision bf16 (BASELINE config 5 through the drop-in script) gives label volumes with Dice >= 0.98 against the fp32 run."""
    import shutil
    from ukbb_cardiac_amd import deploy_network_ao, measures, nifti
    from ukbb_cardiac_amd.image_utils import np_categorical_dice
    from ukbb_cardiac_amd.phantom import cine_phantom
    from test_host_pipeline import _pandas_csv
    arch, params, flat, mp = _model(tmp_path, 'UNet_ao')
    src = tmp_path / 'src'
    names = ['2001', '2002']
    aff = np.diag([1.6, 1.6, 6.0, 1.0])
    pixdim = np.array([1, 1.6, 1.6, 6.0, 0.01, 0, 0, 0], np.float32)
    for i, nm in enumerate(names):
        (src / nm).mkdir(parents=True)
        T = 12 + i
        cine = np.round(cine_phantom(T, 200, 180, seed=70 + i)[..., 0].transpose(1, 2, 0)[:, :, None, :] * 1000.0).astype(np.float32)
        nifti.save(cine, str(src / nm / 'ao.nii.gz'), aff, pixdim)
    seg = {}
    for prec in ('fp32', 'bf16'):
        work = tmp_path / prec
        shutil.copytree(str(src), str(work))
        csv = str(tmp_path / (prec + '.csv'))
        deploy_network_ao.main(['--seq_name', 'ao', '--data_dir', str(work), '--model_path', mp, '--model', 'UNet', '--precision', prec,
                                '--output_csv', csv])
        want = []
        for nm in names:
            hdr = nifti.load_header(str(work / nm / 'ao.nii.gz'))
            dx, dy = hdr['pixdim'][1:3]
            s = nifti.load(str(work / nm / 'seg_ao.nii.gz')).get_data()
            assert s.dtype == np.int32 and s.shape[:3] == (200, 180, 1)
            line = []
            for l in (1, 2):
                A = np.sum(s == l, axis=(0, 1, 2)) * (dx * dy)
                line += [A.max(), A.min(), float('nan')]
            want.append(line)
            seg[(prec, nm)] = s
        assert open(csv).read() == _pandas_csv(str(tmp_path / 'pd.csv'), want, names, measures.AO_COLUMNS)
    for nm in names:
        assert len(np.unique(seg[('fp32', nm)])) == 3
        for k in (1, 2):
            assert np_categorical_dice(seg[('bf16', nm)], seg[('fp32', nm)], k) >= 0.98
