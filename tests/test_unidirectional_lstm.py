"""The single-direction ConvLSTM head (reference common/network_ao.py:214-252 Conv_LSTM, UNet_LSTM_Model with bidirectional=False
:349-352, train_network_ao.py:67) served by the bidirectional engine through weights.embed_unidirectional_lstm (zero backward cell).

CPU: the oracle's own restatement of Conv_LSTM against its BiConv_LSTM on the embedded parameter set -- the backward hidden maps are exactly
zero and the logits agree to the rounding of a 16- against a 32-term sum (the extra terms are 0 * 0), so the embedding is exact at the level of
the graph.  GPU: the engine on the embedded set against the oracle's Conv_LSTM, fp32 and bf16, sequence and cine."""
import numpy as np
import pytest

from oracle import fcn_oracle as O
from ukbb_cardiac_amd.arch import MODELS
from ukbb_cardiac_amd.weights import embed_unidirectional_lstm, synthetic_params


def _uni_params(seed):
    arch = MODELS['UNet-LSTM_ao']
    rng = np.random.default_rng(seed)
    p = synthetic_params(arch, seed)
    nh = arch.same_dim
    uni = {k: v for k, v in p.items() if not k.startswith('lstm')}
    uni['lstm'] = p['lstm_fw']
    uni['lstm_conv'] = {'kernel': rng.normal(0, 0.4, size=(1, 1, nh, arch.n_class)).astype(np.float32),
                        'bias': rng.normal(0, 0.1, size=arch.n_class).astype(np.float32)}
    return arch, uni


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_zero_backward_cell_is_the_single_direction_graph_exactly(dtype):
    arch, uni = _uni_params(5)
    emb = embed_unidirectional_lstm(uni, arch.same_dim)
    assert set(emb) == {s.name for s in arch.layer_specs()}
    rng = np.random.default_rng(1)
    feats = rng.normal(0, 1, size=(2, 5, 12, 16, arch.n_filter[0])).astype(dtype)
    a = O.conv_lstm(feats, uni, arch.same_dim)
    b = O.biconv_lstm(feats, emb, arch.same_dim)
    assert a.dtype == b.dtype == dtype and a.shape == (2, 5, 12, 16, arch.n_class)
    tol = 1e-13 if dtype is np.float64 else 2e-6                       # summation order of the 1x1 conv only (numpy's matmul blocks K = 16 and K = 32 differently)
    assert np.abs(a - b).max() <= tol * np.abs(a).max()
    # the backward direction of the embedded set is exactly zero at every step
    h = c = np.zeros((2, 12, 16, arch.same_dim), dtype)
    for t in range(4, -1, -1):
        h, c = O.conv_lstm_cell(feats[:, t], h, c, emb['lstm_bw'])
        assert not h.any() and not c.any()
    # and it is NOT the bidirectional model with a trained backward cell feeding the output conv
    emb2 = dict(emb); emb2['lstm_bw'] = synthetic_params(arch, 6)['lstm_bw']
    emb2['lstm_out'] = {'kernel': np.concatenate([emb['lstm_out']['kernel'][:, :, :16]] * 2, axis=2), 'bias': emb['lstm_out']['bias']}
    assert np.abs(a - O.biconv_lstm(feats, emb2, arch.same_dim)).max() > 1e-3
    # whole graph, images in
    img = rng.normal(0, 1, size=(1, 3, 32, 32, 1)).astype(np.float32)
    u, e = O.unet_lstm(img, uni, arch.same_dim, dtype=dtype, bidirectional=False), O.unet_lstm(img, emb, arch.same_dim, dtype=dtype)
    assert np.abs(u - e).max() <= tol * np.abs(u).max()


def test_embedding_checks_the_logits_kernel_shape():
    arch, uni = _uni_params(3)
    uni['lstm_conv']['kernel'] = uni['lstm_conv']['kernel'][:, :, :8]
    with pytest.raises(ValueError):
        embed_unidirectional_lstm(uni, arch.same_dim)


@pytest.mark.gpu
@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_engine_runs_the_single_direction_head(prec):
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.phantom import cine_phantom
    arch, uni = _uni_params(21)
    emb = embed_unidirectional_lstm(uni, arch.same_dim)
    T = arch.fc
    img = ((cine_phantom(2 * T, 48, 64, seed=4) - 0.3) / 0.25).astype(np.float32).reshape(2, T, 48, 64, 1)
    ref = O.unet_lstm(img, uni, arch.same_dim, dtype=np.float64, bidirectional=False)
    with Engine(arch, emb) as eng:
        eng.set_precision(prec)
        out = eng.run_seq(img, want_logits=True)
        frames = img[0, :, :, :, 0]
        prob_c, pred_c = eng.run_cine(frames)
    scale = float(np.abs(ref).max())
    err = float(np.abs(out['logits'] - ref).max()) / scale
    if prec == 'fp32':
        assert err < 1e-3, err                                           # measured ~3e-6
        flips = out['pred'] != ref.argmax(-1)
        assert np.all(O.top2_margin(ref)[flips] < 1e-4)
    else:
        assert err < 0.08, err                                           # bf16 storage: judged like test_bf16_cine_dice_vs_fp32, by agreement of the labels
        assert (out['pred'] == ref.argmax(-1)).mean() > 0.97
    assert np.isfinite(prob_c).all() and pred_c.shape == frames.shape


@pytest.mark.gpu
@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_clearing_the_zero_cells_maps_equals_running_it(prec, monkeypatch):
    """engine.cpp run_bilstm clears the hidden maps of an all-zero backward cell instead of running its T - 1 time steps; UKBB_LSTM_RUN_ZERO_CELL=1
    (read at plan build) runs them anyway: every output bit must agree, sequence and cine."""
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.phantom import cine_phantom
    arch, uni = _uni_params(8)
    emb = embed_unidirectional_lstm(uni, arch.same_dim)
    T = arch.fc
    frames = ((cine_phantom(14, 64, 48, seed=9)[..., 0] - 0.3) / 0.25).astype(np.float32)
    seq = frames[:T].reshape(1, T, 64, 48, 1)
    got = {}
    for tag in ('skip', 'run'):
        if tag == 'run':
            monkeypatch.setenv('UKBB_LSTM_RUN_ZERO_CELL', '1')
        else:
            monkeypatch.delenv('UKBB_LSTM_RUN_ZERO_CELL', raising=False)
        with Engine(arch, emb) as eng:
            eng.set_precision(prec)
            got[tag] = (eng.run_seq(seq, want_logits=True), eng.run_cine(frames))
            got[tag + '2'] = eng.run_cine(frames)                        # a second cine on the same handle (the cleared maps are cleared again)
    for k in ('logits', 'prob', 'pred'):
        assert np.array_equal(got['skip'][0][k], got['run'][0][k]), k
    for i in (0, 1):
        assert np.array_equal(got['skip'][1][i], got['run'][1][i]) and np.array_equal(got['skip'][1][i], got['skip2'][i])


@pytest.mark.gpu
def test_drop_in_script_on_a_single_direction_checkpoint(tmp_path):
    """deploy_network_ao.py on a checkpoint-V2 triple of the Conv_LSTM model (variables LSTM/<cell>/{kernel,biases}, LSTM/conv2d/{kernel,bias}, written by
    tests/tf_bundle_writer.py): seg_ao.nii.gz = argmax of the reference's window tiling (deploy_network_ao.py:129-183) of this model's sequence outputs."""
    from tests.tf_bundle_writer import write_checkpoint
    from test_tf_checkpoint import _tf_tensors
    from ukbb_cardiac_amd import deploy_network_ao, nifti
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.image_utils import normalise_intensity
    arch, uni = _uni_params(33)
    emb = embed_unidirectional_lstm(uni, arch.same_dim)
    t = _tf_tensors(arch, emb, with_slots=False)
    ck = {n: v for n, v in t.items() if not n.startswith('LSTM/')}
    ck['LSTM/conv_lstm_cell/kernel'] = t['LSTM/forward/conv_lstm_cell/kernel']
    ck['LSTM/conv_lstm_cell/biases'] = t['LSTM/forward/conv_lstm_cell/biases']
    ck['LSTM/conv2d/kernel'] = uni['lstm_conv']['kernel']
    ck['LSTM/conv2d/bias'] = uni['lstm_conv']['bias']
    mp = str(tmp_path / 'UNet-LSTM_uni')
    write_checkpoint(mp, ck, tensor_crc=False)
    rng = np.random.default_rng(5)
    vol = np.round(100 * rng.gamma(2.0, 1.0, size=(80, 96, 1, 12))).astype(np.float32)
    d = tmp_path / 'data' / 'subj1'
    d.mkdir(parents=True)
    nifti.save(vol, str(d / 'ao.nii.gz'), np.diag([1.6, 1.6, 6.0, 1.0]), pixdim=[1, 1.6, 1.6, 6, 0.01, 0, 0, 0])
    deploy_network_ao.main(['--seq_name', 'ao', '--data_dir', str(tmp_path / 'data'), '--model_path', mp])
    seg = nifti.load(str(d / 'seg_ao.nii.gz')).get_data()
    with Engine(arch, emb) as eng:
        want = O.aortic_lstm_prob_sequence(normalise_intensity(vol.copy(), 10.0), lambda x: eng.run_seq(x)['prob'])
    assert seg.dtype == np.int32 and seg.shape == vol.shape
    np.testing.assert_array_equal(seg, np.argmax(want, -1).astype(np.int32))
    assert len(np.unique(seg)) > 1
