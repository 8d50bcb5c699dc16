"""UNet-LSTM (BiConvLSTM head) on the GPU against the numpy restatement of
common/network_ao.py:255-399 and the tiling loop of common/deploy_network_ao.py:129-183."""
import numpy as np
import pytest

from oracle import fcn_oracle as O

pytestmark = pytest.mark.gpu
LOGIT_RTOL = 1e-3           # north_star: logits within 1e-3 relative (fp32)


@pytest.fixture(scope='module')
def model():
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['UNet-LSTM_ao']
    params = synthetic_params(arch, 1234)
    eng = Engine(arch, params)
    yield arch, params, eng
    eng.close()


@pytest.mark.parametrize('shape', [(1, 9, 32, 48), (2, 9, 64, 32)])
def test_forward_seq_vs_oracle(model, shape):
    arch, params, eng = model
    x = np.random.default_rng(shape[2]).standard_normal(shape + (1,)).astype(np.float32)
    out = eng.run_seq(x, want_logits=True)
    ref = O.unet_lstm(x, params, arch.n_hidden, n_block=arch.n_block, dtype=np.float64)
    scale = np.abs(ref).max()
    err = np.abs(out['logits'] - ref).max()
    assert err <= LOGIT_RTOL * scale, 'logits err %.3e vs scale %.3e' % (err, scale)
    prob_ref = O.softmax(ref)
    assert np.abs(out['prob'] - prob_ref).max() < 1e-4
    bad = out['pred'] != O.argmax_pred(ref)
    assert np.all(O.top2_margin(ref)[bad] < 1e-4), 'label disagreement away from a tie'
    assert out['pred'].dtype == np.int32 and out['pred'].shape == shape
    assert np.array_equal(out['pred'], np.argmax(out['prob'], -1))           # pred is the argmax of prob
    assert len(np.unique(out['pred'])) > 1


def test_sequences_are_independent_and_order_sensitive(model):
    arch, params, eng = model
    x = np.random.default_rng(2).standard_normal((3, 9, 32, 32, 1)).astype(np.float32)
    all_ = eng.run_seq(x, want_logits=True)['logits']
    one = eng.run_seq(x[1:2], want_logits=True)['logits']
    assert np.array_equal(all_[1:2], one)                                    # batch composition does not matter
    rev = eng.run_seq(x[1:2, ::-1].copy(), want_logits=True)['logits'][:, ::-1]
    assert np.abs(rev - one).max() > 1e-3                                    # forward/backward cells differ


def test_forward_cine_equals_reference_tiling_of_window_calls(model):
    """ukbb_fcn_forward_cine (features once per frame, all windows batched, tiling on device) against the
    reference's loop -- one forward_seq call per window centre, numpy accumulation (deploy_network_ao.py:147-183)."""
    arch, params, eng = model
    F, H, W = 13, 32, 48
    frames = np.random.default_rng(7).standard_normal((F, H, W)).astype(np.float32)
    prob, pred = eng.run_cine(frames)
    image = np.transpose(frames, (1, 2, 0))[:, :, None, :]                    # (X, Y, Z=1, T)
    prob_ref = np.zeros((H, W, 1, F, 3), np.float32)
    weight = np.zeros((1, 1, 1, F, 1))
    w = np.reshape(O.aortic_window_weights(5, 0.1), (1, 1, 1, 9, 1))
    for t in range(F):
        idx = O.aortic_window_indices(t, F, 5)
        image_idx = np.transpose(image[:, :, :, idx], axes=(2, 3, 0, 1)).astype(np.float32)[..., None]
        prob_idx = np.transpose(eng.run_seq(image_idx)['prob'], axes=(2, 3, 0, 1, 4))
        prob_ref[:, :, :, idx] += prob_idx * w
        weight[:, :, :, idx] += w
    prob_ref /= weight
    want = np.transpose(prob_ref[:, :, 0], (2, 0, 1, 3))                      # [F,H,W,C]
    np.testing.assert_array_equal(prob, want)                                 # same arithmetic, same order
    np.testing.assert_array_equal(pred, np.argmax(want, -1).astype(np.int32))


def _reference_tiling(eng, frames, time_step, weight_R=5, weight_r=0.1):
    """deploy_network_ao.py:129-183 verbatim in numpy (fancy-indexed `+=`, float64 weights, `prob /= weight`) on an
    un-padded (X,Y,1,T) volume, with this engine's forward_seq standing for sess.run."""
    F, H, W = frames.shape
    image = np.transpose(frames, (1, 2, 0))[:, :, None, :]                    # (X, Y, Z=1, T)
    K = 2 * weight_R - 1
    prob = np.zeros((H, W, 1, F, 3), np.float32)
    weight = np.zeros((1, 1, 1, F, 1))
    w = np.reshape(O.aortic_window_weights(weight_R, weight_r), (1, 1, 1, K, 1))
    for t in range(0, F, time_step):
        idx = O.aortic_window_indices(t, F, weight_R)
        image_idx = np.transpose(image[:, :, :, idx], axes=(2, 3, 0, 1)).astype(np.float32)[..., None]
        prob_idx = np.transpose(eng.run_seq(image_idx)['prob'], axes=(2, 3, 0, 1, 4))
        prob[:, :, :, idx] += prob_idx * w
        weight[:, :, :, idx] += w
    with np.errstate(invalid='ignore', divide='ignore'):
        prob /= weight
    return np.transpose(prob[:, :, 0], (2, 0, 1, 3))                          # [F,H,W,C]


@pytest.mark.parametrize('F,time_step', [(13, 2), (13, 3), (20, 7), (12, 10), (25, 12), (8, 1), (6, 1), (5, 2), (4, 1)])
def test_forward_cine_time_step_and_short_cines(model, F, time_step):
    """--time_step (deploy_network_ao.py:26,147): window centres range(0, T, time_step).  Frames no window reaches
    divide 0/0 -> NaN probabilities and label 0, exactly as numpy gives the reference; cines shorter than the window
    (duplicate indices inside one window) follow numpy's last-write-wins `a[idx] += b`."""
    arch, params, eng = model
    frames = np.random.default_rng(100 * F + time_step).standard_normal((F, 32, 32)).astype(np.float32)
    prob, pred = eng.run_cine(frames, time_step=time_step)
    want = _reference_tiling(eng, frames, time_step)
    np.testing.assert_array_equal(prob, want)                                 # NaNs compare equal here
    np.testing.assert_array_equal(pred, np.argmax(want, -1).astype(np.int32))
    uncovered = np.isnan(want).any(axis=(1, 2, 3))
    if (F, time_step) in ((12, 10), (25, 12)):
        assert uncovered.any() and (pred[uncovered] == 0).all()
    if time_step <= 9 and F >= 9:
        assert not uncovered.any()


def test_forward_seq_256_vs_oracle(model):
    """The fixed aortic network input size (256 x 256, deploy_network_ao.py:105): one 9-frame window through the U-Net
    + bidirectional ConvLSTM against the fp64 numpy restatement (network_ao.py:255-399)."""
    arch, params, eng = model
    from ukbb_cardiac_amd.phantom import cine_phantom
    x = ((cine_phantom(9, 256, 256, seed=71) - 0.3) / 0.25).astype(np.float32)[None]       # [1,9,256,256,1], z-score-like range
    out = eng.run_seq(x, want_logits=True)
    ref = O.unet_lstm(x, params, arch.n_hidden, n_block=arch.n_block, dtype=np.float64)
    scale = np.abs(ref).max()
    err = np.abs(out['logits'] - ref).max()
    assert err <= LOGIT_RTOL * scale, 'logits err %.3e vs scale %.3e' % (err, scale)
    bad = out['pred'] != O.argmax_pred(ref)
    assert np.all(O.top2_margin(ref)[bad] < 1e-4) and bad.sum() <= 8
    assert np.abs(out['prob'] - O.softmax(ref)).max() < 1e-4
    assert len(np.unique(out['pred'])) > 1


def test_errors(model):
    from ukbb_cardiac_amd import _lib
    arch, params, eng = model
    with pytest.raises(_lib.UkbbFcnError, match='sequences'):
        eng.run(np.zeros((1, 32, 32, 1), np.float32))
    with pytest.raises(_lib.UkbbFcnError, match='at least'):                # the reference raises IndexError here
        eng.run_cine(np.zeros((3, 32, 32), np.float32))
    with pytest.raises(_lib.UkbbFcnError, match='time_step'):
        eng.run_cine(np.zeros((12, 32, 32), np.float32), time_step=0)
    with pytest.raises(_lib.UkbbFcnError, match='window'):
        eng.run_cine(np.zeros((12, 32, 32), np.float32), weight_R=4)


def test_drop_in_aortic_script_with_the_default_model(tmp_path, model):
    """demo_pipeline.py:116-117 (`--model UNet-LSTM` is the default): seg_ao.nii.gz == argmax of the tiling loop
    restated in the oracle, driven by this engine's per-window forward_seq."""
    import gzip
    from ukbb_cardiac_amd import deploy_network_ao, nifti
    from ukbb_cardiac_amd.image_utils import normalise_intensity
    from ukbb_cardiac_amd.weights import save_blob
    arch, params, eng = model
    mp = str(tmp_path / 'UNet-LSTM_ao')
    save_blob(mp + '.ukbbw', arch, params)
    rng = np.random.default_rng(31)
    vol = (100 * rng.gamma(2.0, 1.0, size=(70, 90, 1, 11))).astype(np.float32)
    d = tmp_path / 'data' / 'subj1'
    d.mkdir(parents=True)
    nifti.save(vol, str(d / 'ao.nii.gz'), np.diag([1.6, 1.6, 6.0, 1.0]), pixdim=[1, 1.6, 1.6, 6, 0.01, 0, 0, 0])
    deploy_network_ao.main(['--seq_name', 'ao', '--data_dir', str(tmp_path / 'data'), '--model_path', mp])
    seg = nifti.load(str(d / 'seg_ao.nii.gz')).get_data()
    want = O.aortic_lstm_prob_sequence(normalise_intensity(vol.copy(), 10.0), lambda x: eng.run_seq(x)['prob'])
    assert seg.dtype == np.int32 and seg.shape == vol.shape
    np.testing.assert_array_equal(seg, np.argmax(want, -1).astype(np.int32))
    with pytest.raises(SystemExit):                                        # a plain UNet flag on an LSTM model
        deploy_network_ao.main(['--seq_name', 'ao', '--data_dir', str(tmp_path / 'data'), '--model_path', mp, '--model', 'UNet'])


@pytest.mark.parametrize('shape,time_step', [((70, 90, 1, 11), 1), ((240, 196, 1, 25), 1), ((64, 48, 2, 13), 2)])
def test_aortic_device_pipeline_equals_host_pipeline(model, shape, time_step):
    """z-score, pad, transposes, windows, argmax on the GPU (device_pipeline.aortic_lstm_sequence_device) against the numpy
    mirror of deploy_network_ao.py:92-108,129-189 driven by the same engine: same probabilities bit for bit, same labels."""
    from ukbb_cardiac_amd import pipeline
    from ukbb_cardiac_amd.device_pipeline import aortic_lstm_sequence_device
    arch, params, eng = model
    rng = np.random.default_rng(shape[0] + time_step)
    vol = np.asfortranarray(np.round(100 * rng.gamma(2.0, 1.0, size=shape)).astype(np.float32))
    keep = vol.copy()
    pred, aux = aortic_lstm_sequence_device(vol, eng, time_step=time_step, return_aux=True)
    assert np.array_equal(vol, keep)                                       # input untouched
    prob = pipeline.aortic_lstm_prob_sequence(vol, lambda f, R, r, ts=1: eng.run_cine(f, R, r, ts)[0], time_step=time_step)
    np.testing.assert_array_equal(aux['prob'], prob)
    want = np.argmax(prob, axis=-1).astype(np.int32)
    assert pred.dtype == np.int32 and pred.shape == shape
    np.testing.assert_array_equal(pred, want)
    for c in range(arch.n_class):                                          # per-frame class areas (eval_aortic_area.py:60-78)
        np.testing.assert_array_equal(aux['counts'][:, c], (want == c).sum(axis=(0, 1, 2)))
    assert len(np.unique(pred)) > 1


def test_aortic_script_device_and_host_preprocessing_write_the_same_file(tmp_path, model):
    from ukbb_cardiac_amd import deploy_network_ao, nifti
    from ukbb_cardiac_amd.weights import save_blob
    arch, params, eng = model
    mp = str(tmp_path / 'UNet-LSTM_ao')
    save_blob(mp + '.ukbbw', arch, params)
    rng = np.random.default_rng(77)
    vol = np.round(100 * rng.gamma(2.0, 1.0, size=(96, 80, 1, 14))).astype(np.float32)
    d = tmp_path / 'data' / 'subj1'
    d.mkdir(parents=True)
    nifti.save(vol, str(d / 'ao.nii.gz'), np.diag([1.6, 1.6, 6.0, 1.0]), pixdim=[1, 1.6, 1.6, 6, 0.01, 0, 0, 0])
    out = {}
    for flag in ('--device_preproc', '--nodevice_preproc'):
        deploy_network_ao.main(['--data_dir', str(tmp_path / 'data'), '--model_path', mp, '--time_step', '2', flag])
        out[flag] = (d / 'seg_ao.nii.gz').read_bytes()
    assert out['--device_preproc'] == out['--nodevice_preproc']


def test_aortic_script_read_ahead_write_behind_gives_the_sequential_files(tmp_path, model):
    """--io_threads only moves file work off the GPU thread: same subjects, same files as the strictly sequential loop."""
    import shutil
    from ukbb_cardiac_amd import deploy_network_ao, nifti
    from ukbb_cardiac_amd.weights import save_blob
    arch, params, eng = model
    mp = str(tmp_path / 'UNet-LSTM_ao')
    save_blob(mp + '.ukbbw', arch, params)
    rng = np.random.default_rng(5)
    src = tmp_path / 'src'
    for i in range(5):
        d = src / ('s%02d' % i)
        d.mkdir(parents=True)
        if i == 3:
            continue                                                        # a subject directory without ao.nii.gz: skipped, as in the reference
        vol = np.round(100 * rng.gamma(2.0, 1.0, size=(80 + 8 * i, 72, 1, 10 + i))).astype(np.float32)
        nifti.save(vol, str(d / 'ao.nii.gz'), np.diag([1.6, 1.6, 6.0, 1.0]), pixdim=[1, 1.6, 1.6, 6, 0.01, 0, 0, 0])
    out = {}
    for thr in (0, 3):
        work = tmp_path / ('run%d' % thr)
        shutil.copytree(src, work)
        lines = []
        deploy_network_ao.run(deploy_network_ao.define_flags().parse(['--data_dir', str(work), '--model_path', mp, '--io_threads', str(thr)])[0],
                              None, log=lines.append, cine_forward=lambda f, R, r, ts=1: eng.run_cine(f, R, r, ts)[0], engine=eng)
        out[thr] = ({p.relative_to(work).as_posix(): p.read_bytes() for p in sorted(work.rglob('seg_ao.nii.gz'))},
                    [l.replace(str(work), 'DIR') for l in lines if 'time' not in l and 'took' not in l])
    assert sorted(out[0][0]) == ['s00/seg_ao.nii.gz', 's01/seg_ao.nii.gz', 's02/seg_ao.nii.gz', 's04/seg_ao.nii.gz']
    assert out[0][0] == out[3][0]
    assert out[0][1] == out[3][1]                                           # same log lines in the same order


def test_bf16_cine_dice_vs_fp32(model):
    """ukbb_fcn_set_precision(UKBB_PREC_BF16) on a UNet-LSTM handle (the script accepts --precision bf16 with the default
    model): the U-Net runs its bf16-storage plan, the ConvLSTM keeps gx and the hidden maps as bf16 in HBM (cell state and arithmetic
    fp32).  One 100-frame 256 x 256 cine against the fp32 path of the same handle: per-class Dice (common/image_utils.py:171-175) >= 0.98
    over the cine and >= 0.97 on every frame where the class is populated; pred = argmax(prob); switching back is exact."""
    from ukbb_cardiac_amd.image_utils import np_categorical_dice
    from ukbb_cardiac_amd.phantom import cine_phantom
    arch, params, eng = model
    frames = ((cine_phantom(100, 256, 256, seed=17)[..., 0] - 0.3) / 0.25).astype(np.float32)
    prob32, pred32 = eng.run_cine(frames)
    eng.set_precision('bf16')
    try:
        prob16, pred16 = eng.run_cine(frames)
        names = eng.kernel_names()
        seq16 = eng.run_seq(frames[None, :9], want_logits=True)
    finally:
        eng.set_precision('fp32')
    again = eng.run_cine(frames)
    assert np.array_equal(again[0], prob32) and np.array_equal(again[1], pred32)
    assert names[0] == 'conv0_0+conv0_1', names                             # the bf16-storage plan (fused stem), not the r01 operand-only mode
    assert np.array_equal(pred16, np.argmax(prob16, -1)) and np.isfinite(prob16).all()
    assert np.array_equal(seq16['pred'], np.argmax(seq16['prob'], -1))
    rel = np.abs(prob16 - prob32).max()
    assert 1e-6 < rel < 0.5, rel                                             # really another precision, and sane
    assert len(np.unique(pred32)) == 3
    for k in (1, 2):
        d = float(np_categorical_dice(pred16, pred32, k))
        worst = min(float(np_categorical_dice(pred16[i], pred32[i], k)) for i in range(100) if (pred32[i] == k).sum() > 500)
        assert d >= 0.98 and worst >= 0.97, (k, d, worst)
    assert (pred16 != pred32).mean() < 0.02


def test_every_step_of_the_fused_convlstm_matches_numpy():
    """r05 white box: the x pass's per-frame first step (h1, both directions) and every later step's hidden map of the fused gate-conv / cell
    kernel (csrc/kernels_wino24.hip, ConvArgs::ls_mode) against oracle conv_lstm_cell on the engine's own feature maps, step by step
    (tools/debug_lstm.py reads the buffers through ukbb_fcn_get_activation('lstm:h1' / 'lstm:hall')); two map sizes, the second with ragged regions."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for hw in (('32', '48'), ('48', '80')):
        r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'debug_lstm.py')] + list(hw), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, timeout=600, env=dict(os.environ, PYTHONPATH=root))
        assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == 'OK', r.stdout[-2000:]


def test_bf16_direct_convlstm_is_deterministic_and_agrees_with_the_winograd_form():
    """r05: in UKBB_PREC_BF16 the ConvLSTM runs as direct 3x3 convs on v_mfma_f32_32x32x16_bf16 with the cell in the epilogue (csrc/kernels_ws.hip, ws_main LS).
    tools/check_lstm_bf16.py: seven cine shapes (ragged tiles, time_step 1-3, 9-100 frames): two runs bit-identical, probabilities within bf16-weight precision of
    the fp32-Winograd-on-bf16-storage form (UKBB_LSTM_BF16_WINOGRAD=1), labels >= 99 % equal."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_lstm_bf16.py')], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == 'OK', r.stdout[-2000:]


def test_other_window_lengths_and_class_counts():
    """The fused ConvLSTM for unrolled lengths T in {1, 3, 5, 13} and 2 / 3 / 4 classes (tools/check_lstm_variants.py): fp32 logits within 1e-3 of the fp64
    restatement, the cine path finite with pred = argmax(prob), the bf16 form >= 97 % label agreement."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_lstm_variants.py')], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == 'OK', r.stdout[-2000:]


def test_aortic_script_with_precision_bf16_on_the_default_model(tmp_path, model):
    """`deploy_network_ao.py --precision bf16` with the default model (UNet-LSTM) ran an untested mode.  Now: the script's seg_ao.nii.gz under
    --precision bf16 against its own fp32 output on the same subject: int32, same shape, >= 98 % of the voxels equal."""
    from ukbb_cardiac_amd import deploy_network_ao, nifti
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import save_blob
    arch, params, eng = model
    mp = str(tmp_path / 'UNet-LSTM_ao')
    save_blob(mp + '.ukbbw', arch, params)
    vol = np.asfortranarray(np.round(cine_phantom(20, 200, 180, seed=9)[..., 0].transpose(1, 2, 0)[:, :, None, :] * 1000.0).astype(np.float32))
    segs = {}
    for prec in ('fp32', 'bf16'):
        d = tmp_path / prec / 'subj1'
        d.mkdir(parents=True)
        nifti.save(vol, str(d / 'ao.nii.gz'), np.diag([1.6, 1.6, 6.0, 1.0]), pixdim=[1, 1.6, 1.6, 6, 0.01, 0, 0, 0])
        deploy_network_ao.main(['--seq_name', 'ao', '--data_dir', str(tmp_path / prec), '--model_path', mp, '--precision', prec])
        segs[prec] = nifti.load(str(d / 'seg_ao.nii.gz')).get_data()
    assert segs['bf16'].dtype == np.int32 and segs['bf16'].shape == vol.shape
    assert len(np.unique(segs['fp32'])) > 1
    assert (segs['bf16'] == segs['fp32']).mean() >= 0.98
