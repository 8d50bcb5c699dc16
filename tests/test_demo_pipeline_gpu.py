"""The outer boundary as the reference drives it: demo_pipeline.py:50-54,63-64,89-96,116-117 rehearsed on the GPU with this
repository's drop-in scripts in place of common/deploy_network.py / deploy_network_ao.py -- two subjects under ``demo_image/``,
the five trained models as TF checkpoint-V2 triples under ``trained_model/`` (written by tests/tf_bundle_writer.py: no
TensorFlow here), the command lines verbatim (relative paths, ``CUDA_VISIBLE_DEVICES=0 python3 ...``), then every file the
evaluation scripts open checked for existence, shape, dtype and header, and one of them per model against
the C / numpy oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import c_oracle, fcn_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_demo_pipeline_command_lines(tmp_path):
    from tests.tf_bundle_writer import write_checkpoint
    from test_tf_checkpoint import _tf_tensors
    from ukbb_cardiac_amd import nifti
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import pack_flat, synthetic_params
    os.makedirs(str(tmp_path / 'trained_model'))
    models = {}
    for name in ('FCN_sa', 'FCN_la_2ch', 'FCN_la_4ch', 'FCN_la_4ch_seg4', 'UNet-LSTM_ao'):       # demo_pipeline.py:50-54
        arch = MODELS[name]
        params = synthetic_params(arch, 1234)
        write_checkpoint(str(tmp_path / 'trained_model' / name), _tf_tensors(arch, params))
        open(str(tmp_path / 'trained_model' / (name + '.meta')), 'wb').write(b'')                 # downloaded, never read by the engine
        models[name] = (arch, params)
    shapes = {'sa': (120, 132, 4, 6), 'la_2ch': (130, 150, 1, 6), 'la_4ch': (130, 150, 1, 6), 'ao': (140, 120, 1, 12)}
    aff = {'sa': np.diag([1.8, 1.8, 10.0, 1.0]), 'la_2ch': np.diag([1.8, 1.8, 6.0, 1.0]), 'la_4ch': np.diag([1.8, 1.8, 6.0, 1.0]),
           'ao': np.diag([1.6, 1.6, 6.0, 1.0])}
    vols = {}
    for subj in ('1', '2'):                                                                       # demo_pipeline.py:31-37
        os.makedirs(str(tmp_path / 'demo_image' / subj))
        for seq, (X, Y, Z, T) in shapes.items():
            p = cine_phantom(Z * T, X, Y, seed=int(subj) * 10 + len(seq))[..., 0]
            vol = np.round(p.reshape(T, Z, X, Y).transpose(2, 3, 1, 0) * 1000.0).astype(np.float32)
            pixdim = np.array([1, aff[seq][0, 0], aff[seq][1, 1], aff[seq][2, 2], 0.03, 0, 0, 0], np.float32)
            nifti.save(vol, str(tmp_path / 'demo_image' / subj / (seq + '.nii.gz')), aff[seq], pixdim)
            vols[(subj, seq)] = vol
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    env.pop('HIP_VISIBLE_DEVICES', None)
    dn, dao = os.path.join(ROOT, 'ukbb_cardiac_amd', 'deploy_network.py'), os.path.join(ROOT, 'ukbb_cardiac_amd', 'deploy_network_ao.py')
    lines = [                                                                                     # verbatim but for the script path
        'CUDA_VISIBLE_DEVICES=0 python3 {0} --seq_name sa --data_dir demo_image --model_path trained_model/FCN_sa'.format(dn),
        'CUDA_VISIBLE_DEVICES=0 python3 {0} --seq_name la_2ch --data_dir demo_image --model_path trained_model/FCN_la_2ch'.format(dn),
        'CUDA_VISIBLE_DEVICES=0 python3 {0} --seq_name la_4ch --data_dir demo_image --model_path trained_model/FCN_la_4ch'.format(dn),
        'CUDA_VISIBLE_DEVICES=0 python3 {0} --seq_name la_4ch --data_dir demo_image --seg4 --model_path trained_model/FCN_la_4ch_seg4'.format(dn),
        'CUDA_VISIBLE_DEVICES=0 python3 {0} --seq_name ao --data_dir demo_image --model_path trained_model/UNet-LSTM_ao'.format(dao),
    ]
    for cmd in lines:
        r = subprocess.run(cmd.replace('python3', sys.executable, 1), shell=True, cwd=str(tmp_path), env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, cmd + '\n' + r.stdout[-2000:]
        assert 'Start' in r.stdout and r.stdout.count('Saving segmentation') == 2, r.stdout[-1500:]
    # ---- the files the evaluation scripts read ----
    for subj in ('1', '2'):
        d = tmp_path / 'demo_image' / subj
        for seq, pre in (('sa', 'seg'), ('la_2ch', 'seg'), ('la_4ch', 'seg'), ('la_4ch', 'seg4')):
            seg = nifti.load(str(d / ('%s_%s.nii.gz' % (pre, seq))))
            assert seg.data.dtype == np.float64 and seg.data.shape == shapes[seq]                  # deploy_network.py:92,136
            assert np.allclose(seg.affine, aff[seq]) and seg.header['pixdim'][4] == np.float32(0.03)
            for fr in ('ED', 'ES'):
                assert nifti.load(str(d / ('%s_%s_%s.nii.gz' % (pre, seq, fr)))).data.shape == shapes[seq][:3]
                assert nifti.load(str(d / ('%s_%s.nii.gz' % (seq, fr)))).data.dtype == np.float32
        ao = nifti.load(str(d / 'seg_ao.nii.gz'))
        assert ao.data.dtype == np.int32 and ao.data.shape == shapes['ao']                         # deploy_network_ao.py:189-196
    # ---- content: one subject per FCN model against the C oracle through the restated loop ----
    for seq, pre, name, seg4 in (('sa', 'seg', 'FCN_sa', False), ('la_2ch', 'seg', 'FCN_la_2ch', False),
                                 ('la_4ch', 'seg', 'FCN_la_4ch', False), ('la_4ch', 'seg4', 'FCN_la_4ch_seg4', True)):
        arch, params = models[name]
        flat = pack_flat(arch, params)
        want, _, _, _ = O.deploy_sequence(vols[('1', seq)].copy(), lambda b: c_oracle.forward(arch, flat, b, want_logits=False)[2], seq, seg4)
        got = nifti.load(str(tmp_path / 'demo_image' / '1' / ('%s_%s.nii.gz' % (pre, seq)))).data
        assert int((got != want).sum()) <= 3, (name, int((got != want).sum()))
        assert len(np.unique(got)) == arch.n_class
