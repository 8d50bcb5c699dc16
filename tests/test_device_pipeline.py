"""Device-side pre/post-processing.  CPU: the host half of the percentile
(ranks + numpy's interpolation) against np.percentile.  GPU: radix select vs np.partition, the
packed network input and the unpacked labels bit for bit against the numpy mirror of
common/deploy_network.py:86-131."""
import numpy as np
import pytest

from ukbb_cardiac_amd import device_pipeline as dp
from ukbb_cardiac_amd.pipeline import pad_amounts, pick_ed_es, segment_sequence


@pytest.mark.parametrize('n', [2, 3, 101, 1000, 99991])
@pytest.mark.parametrize('q', [1, 10, 50, 99, 0, 100, 37.5])
def test_host_half_of_percentile_matches_numpy(n, q):
    rng = np.random.default_rng(n * 1000 + int(q * 10))
    a = (1000 * rng.gamma(2.0, 1.0, size=n)).astype(np.float32)
    a[rng.integers(0, n, size=max(1, n // 10))] = a[0]              # ties
    s = np.sort(a)
    k, k1, g = dp.percentile_ranks(n, q)
    got = dp.lerp_like_numpy(s[k], s[k1], g)
    want = np.percentile(a, (q, 50.0))[0]        # tuple q, as rescale_intensity passes it (float64 quantiles)
    assert got == want and got.dtype == want.dtype


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1, 5, 4096, 1000003])
def test_select_kth_exact(n):
    import ctypes as C
    import torch
    from ukbb_cardiac_amd import _lib
    rng = np.random.default_rng(n)
    a = (rng.standard_normal(n) * 1000).astype(np.float32)
    if n > 10:
        a[::7] = a[3]                                              # heavy ties
        a[1] = 0.0; a[2] = -0.0; a[4] = np.float32(3.0e38); a[5] = np.float32(-3.0e38); a[6] = np.float32(1e-42)   # denormal
    t = torch.from_numpy(a).cuda()
    ranks = sorted({0, n - 1, n // 2, n // 100, (99 * n) // 100, min(n - 1, n // 100 + 1)})
    r = (C.c_uint64 * len(ranks))(*ranks)
    out = np.empty(len(ranks), np.float32)
    _lib.check(_lib.lib.ukbb_fcn_select_kth(t.data_ptr(), n, r, len(ranks), _lib.f32ptr(out), 0), 'select')
    s = np.sort(a)
    np.testing.assert_array_equal(out, s[ranks])                    # -0.0 == 0.0 compares equal, as in a sort
    with pytest.raises(_lib.UkbbFcnError):
        bad = (C.c_uint64 * 1)(n)
        _lib.check(_lib.lib.ukbb_fcn_select_kth(t.data_ptr(), n, bad, 1, _lib.f32ptr(out), 0), 'select')


@pytest.mark.gpu
@pytest.mark.parametrize('order', ['F', 'C'])
def test_device_percentiles_equal_numpy(order):
    import torch
    rng = np.random.default_rng(5)
    vol = np.asarray((1000 * rng.gamma(2.0, 1.0, size=(37, 41, 3, 5))).astype(np.float32), order=order)
    lo, hi = dp.device_percentiles(torch.from_numpy(vol).cuda(), (1, 99))
    want = np.percentile(vol, (1, 99))
    assert lo == want[0] and hi == want[1]


def _engine(model='FCN_sa'):
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS[model]
    return Engine(arch, synthetic_params(arch, 1234))


@pytest.mark.gpu
@pytest.mark.parametrize('shape,order', [((162, 204, 2, 3), 'F'), ((192, 208, 3, 4), 'F'), ((50, 33, 1, 2), 'C')])
def test_rescale_pack_and_unpack_bit_exact(shape, order):
    """Network input and label volume equal the numpy mirror of the reference loop exactly."""
    import torch
    from ukbb_cardiac_amd import _lib
    from ukbb_cardiac_amd.image_utils import rescale_intensity
    X, Y, Z, T = shape
    rng = np.random.default_rng(X)
    vol = np.asarray((1000 * rng.gamma(2.0, 1.0, size=shape)).astype(np.float32), order=order)
    ref = vol.copy(order='K')
    scaled = rescale_intensity(ref, (1, 99))                        # clips ref in place
    X2, Y2, x_pre, x_post, y_pre, y_post = pad_amounts(X, Y)
    padded = np.pad(scaled, ((x_pre, x_post), (y_pre, y_post), (0, 0), (0, 0)), 'constant')
    want = np.transpose(padded, (3, 2, 0, 1)).reshape(T * Z, X2, Y2).astype(np.float32)
    t = torch.from_numpy(vol).cuda()
    lo, hi = dp.device_percentiles(t, (1, 99))
    batch = torch.empty((T * Z, X2, Y2), dtype=torch.float32, device='cuda')
    sx, sy, sz, st = t.stride()
    _lib.check(_lib.lib.ukbb_fcn_rescale_pack(t.data_ptr(), X, Y, Z, T, sx, sy, sz, st, float(lo), float(hi), X2, Y2, x_pre, y_pre,
                                              batch.data_ptr(), 0), 'pack')
    np.testing.assert_array_equal(batch.cpu().numpy(), want)
    np.testing.assert_array_equal(dp.clip_like_reference(vol[..., 0], (lo, hi)), ref[..., 0])
    # labels: random label batch -> volume + counts
    n_class = 4
    lab = rng.integers(0, n_class, size=(T * Z, X2, Y2)).astype(np.int32)
    pred = torch.from_numpy(lab).cuda()
    out = torch.empty(X * Y * Z * T, dtype=torch.uint8, device='cuda')
    counts = torch.empty((T, n_class), dtype=torch.int64, device='cuda')
    _lib.check(_lib.lib.ukbb_fcn_unpack_labels(pred.data_ptr(), X, Y, Z, T, X2, Y2, x_pre, y_pre, n_class, out.data_ptr(),
                                               counts.data_ptr(), 0), 'unpack')
    want_vol = lab.reshape(T, Z, X2, Y2).transpose(2, 3, 1, 0)[x_pre:x_pre + X, y_pre:y_pre + Y]
    got_vol = out.cpu().numpy().reshape((X, Y, Z, T), order='F')
    np.testing.assert_array_equal(got_vol, want_vol)
    want_counts = np.stack([[np.sum(want_vol[..., t_] == c) for c in range(n_class)] for t_ in range(T)])
    np.testing.assert_array_equal(counts.cpu().numpy(), want_counts)


@pytest.mark.gpu
def test_segment_sequence_device_equals_host_path():
    from ukbb_cardiac_amd.phantom import cine_phantom
    eng = _engine()
    Z, T = 4, 6
    vol = cine_phantom(Z * T, 162, 204, seed=3)[..., 0].reshape(T, Z, 162, 204).transpose(2, 3, 1, 0) * 1000.0
    vol = np.asfortranarray(vol.astype(np.float32))
    host_in = vol.copy(order='F')
    want = segment_sequence(host_in, lambda b: eng.run(b, want_prob=False), batch_slices=7)
    got, aux = dp.segment_sequence_device(vol, eng, batch_slices=7, return_aux=True)
    assert got.dtype == np.float64 and got.shape == vol.shape
    np.testing.assert_array_equal(got, want)
    assert pick_ed_es(want, 'sa') == dp.pick_ed_es_from_counts(aux['counts'], 'sa')
    assert pick_ed_es(want, 'la_2ch') == dp.pick_ed_es_from_counts(aux['counts'], 'la_2ch')
    np.testing.assert_array_equal(dp.clip_like_reference(vol[..., 2], aux['clip']), host_in[..., 2])   # what ED/ES frames save
    with pytest.raises(TypeError):
        dp.segment_sequence_device(vol.astype(np.float64), eng)
    eng.close()


@pytest.mark.gpu
def test_drop_in_script_writes_identical_files_with_and_without_device_preproc(tmp_path):
    """The reference command line (demo_pipeline.py:63-64) through ukbb_cardiac_amd.deploy_network: the five
    output files (deploy_network.py:136-151) are byte-identical between the host and the device pre-processing."""
    import gzip
    from ukbb_cardiac_amd import deploy_network, nifti
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.weights import save_blob, synthetic_params
    arch = MODELS['FCN_sa']
    model = str(tmp_path / 'FCN_sa')
    save_blob(model + '.ukbbw', arch, synthetic_params(arch, 1234))
    rng = np.random.default_rng(11)
    vol = (1000 * rng.gamma(2.0, 1.0, size=(100, 90, 3, 4))).astype(np.float32)
    outs = {}
    for mode in ('device', 'host'):
        d = tmp_path / mode / 'subj1'
        d.mkdir(parents=True)
        nifti.save(vol, str(d / 'sa.nii.gz'), np.diag([1.8, 1.8, 10.0, 1.0]), pixdim=[1, 1.8, 1.8, 10, 0.03, 0, 0, 0])
        deploy_network.main(['--seq_name', 'sa', '--data_dir', str(tmp_path / mode), '--model_path', model,
                             '--device_preproc' if mode == 'device' else '--nodevice_preproc'])
        outs[mode] = {f: gzip.open(str(d / f)).read() for f in
                      ('seg_sa.nii.gz', 'sa_ED.nii.gz', 'sa_ES.nii.gz', 'seg_sa_ED.nii.gz', 'seg_sa_ES.nii.gz')}
    for f in outs['host']:
        assert outs['device'][f] == outs['host'][f], f


@pytest.mark.gpu
def test_subject_pipeline_equals_sequential_device_path():
    """subject_pipeline.SubjectPipeline (pinned staging pool, copy-in / compute / copy-out streams, several subjects in
    flight) returns, subject by subject and in order, exactly what the one-at-a-time device path returns: labels, per-frame
    class counts, clip bounds; staged inputs (what nifti.load(alloc=...) fills) and plain arrays are both accepted."""
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.subject_pipeline import SubjectPipeline, labels_as_float64
    eng = _engine()
    shapes = [(162, 204, 2, 3), (160, 200, 2, 3), (162, 204, 2, 3), (130, 204, 1, 5), (162, 204, 2, 3)]
    vols = []
    for i, (X, Y, Z, T) in enumerate(shapes):
        v = cine_phantom(Z * T, X, Y, seed=40 + i)[..., 0].reshape(T, Z, X, Y).transpose(2, 3, 1, 0) * (900.0 + 50 * i)
        vols.append(np.asfortranarray(v.astype(np.float32)))
    want = [dp.segment_sequence_device(v, eng, batch_slices=5, return_aux=True) for v in vols]
    pipe = SubjectPipeline(eng, (162, 204, 2, 3), batch_slices=5, depth=3, extra_inputs=2)

    def source():
        for i, v in enumerate(vols):
            if i % 2 == 0:
                st = pipe.stage(v.shape)                   # a reader thread's view: fill the pinned buffer in place
                st.array[...] = v
                yield st.array
            else:
                yield v                                    # plain array: copied into a pinned buffer by submit()
    n = 0
    for res, (w_pred, w_aux) in zip(pipe.run(source()), want):
        assert res.labels.dtype == np.uint8 and res.labels.shape == w_pred.shape
        np.testing.assert_array_equal(labels_as_float64(res.labels), w_pred)
        np.testing.assert_array_equal(res.counts, w_aux['counts'])
        assert res.clip == w_aux['clip']
        np.testing.assert_array_equal(res.image, vols[n])  # the staged volume is still there for the ED / ES frames
        n += 1
    assert n == len(vols) and pipe.pending() == 0
    assert pipe._in_free.qsize() == 5                      # every pinned input buffer came back
    with pytest.raises(ValueError):
        pipe.stage((400, 400, 10, 50))                     # larger than the staging buffers
    with pytest.raises(TypeError):
        pipe.stage((10, 10, 1, 1), np.float64)
    eng.close()


# ---- aortic z-score on the device (common/image_utils.py:60-67) -------------------------------------------------------

@pytest.mark.parametrize('n', [1, 2, 3, 7, 100, 1001, 240 * 196 * 50])
@pytest.mark.parametrize('q', [10.0, 1.0, 99.0, 50.0, 0.0, 100.0, 33.3])
def test_host_half_of_scalar_percentile_matches_numpy(n, q):
    """np.percentile(float32 array, Python scalar) stays in float32 (normalise_intensity calls it that way)."""
    rng = np.random.default_rng(n + int(q * 10))
    a = (rng.random(n) * rng.integers(1, 5000)).astype(np.float32)
    s = np.sort(a)
    k, k1, g = dp.scalar_percentile_ranks(n, q)
    got = np.quantile(np.array([s[k], s[k1]], np.float32), g)
    want = np.percentile(a, q)
    assert got == want and type(got) is type(want)


def _aortic_like(shape, seed, order='F'):
    rng = np.random.default_rng(seed)
    v = (rng.gamma(1.5, 120.0, size=shape)).astype(np.float32)
    v[rng.random(shape) < 0.08] = 0.0                               # background ties at the bottom of the histogram
    return np.asarray(np.round(v), dtype=np.float32, order=order)  # MR magnitudes are integers: many ties at the threshold


@pytest.mark.gpu
@pytest.mark.parametrize('shape,order', [((37, 41, 3, 5), 'F'), ((37, 41, 3, 5), 'C'), ((240, 196, 1, 50), 'F'), ((5, 3, 1, 1), 'F')])
def test_roi_compact_is_numpy_boolean_indexing(shape, order):
    import ctypes as C
    import torch
    from ukbb_cardiac_amd import _lib
    img = _aortic_like(shape, 3, order)
    t = torch.from_numpy(img).cuda()
    out = torch.full((img.size,), -1.0, dtype=torch.float32, device='cuda')
    for thr in (np.percentile(img, 10.0), np.float32(-1.0), np.float32(1e9), np.float32(0.0)):
        n = C.c_uint64(0)
        sx, sy, sz, st = t.stride()
        _lib.check(_lib.lib.ukbb_fcn_roi_compact(t.data_ptr(), *shape, sx, sy, sz, st, float(thr), out.data_ptr(), C.byref(n), 0), 'compact')
        want = img[img >= thr]
        assert n.value == want.size
        np.testing.assert_array_equal(out.cpu().numpy()[:n.value], want)


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1, 7, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 16385, 100003, 4233599])
def test_pairwise_sum_is_numpy_add_reduce(n):
    import ctypes as C
    import torch
    from ukbb_cardiac_amd import _lib
    rng = np.random.default_rng(n)
    a = (rng.random(n) * 1000).astype(np.float32)
    t = torch.from_numpy(a).cuda()
    s = C.c_float(0)
    _lib.check(_lib.lib.ukbb_fcn_pairwise_sum(t.data_ptr(), n, 0, 0.0, C.byref(s), 0), 'sum')
    assert np.float32(s.value) == np.add.reduce(a)
    m = np.mean(a)
    _lib.check(_lib.lib.ukbb_fcn_pairwise_sum(t.data_ptr(), n, 1, float(m), C.byref(s), 0), 'sumsq')
    x = a - m
    assert np.float32(s.value) == np.add.reduce(x * x)
    _lib.check(_lib.lib.ukbb_fcn_pairwise_sum(t.data_ptr(), 0, 0, 0.0, C.byref(s), 0), 'empty')
    assert s.value == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('shape,order,seed', [((240, 196, 1, 50), 'F', 1), ((208, 256, 1, 30), 'F', 2), ((61, 47, 2, 11), 'C', 3),
                                              ((256, 256, 1, 20), 'F', 4), ((33, 250, 1, 9), 'F', 5)])
def test_zscore_on_device_is_normalise_intensity(shape, order, seed):
    """mu, sigma + eps and the packed network input bit for bit against image_utils.normalise_intensity + the pad /
    transpose of deploy_network_ao.py:105-108,147-150."""
    import torch
    from ukbb_cardiac_amd import _lib
    from ukbb_cardiac_amd.image_utils import normalise_intensity
    from ukbb_cardiac_amd.pipeline import pad_amounts_fixed
    img = _aortic_like(shape, seed, order)
    X, Y, Z, T = shape
    t = torch.from_numpy(img).cuda()
    mu, den, n_roi, val_l = dp.device_zscore_stats(t, 10.0)
    roi = img >= np.percentile(img, 10.0)
    assert val_l == np.percentile(img, 10.0) and n_roi == int(roi.sum())
    assert mu == np.mean(img[roi]) and type(mu) is np.float32
    assert den == np.std(img[roi]) + 1e-6 and type(den) is np.float32
    X2, Y2, x_pre, x_post, y_pre, y_post = pad_amounts_fixed(X, Y)
    batch = torch.empty((T * Z, X2, Y2), dtype=torch.float32, device='cuda')
    sx, sy, sz, st = t.stride()
    _lib.check(_lib.lib.ukbb_fcn_zscore_pack(t.data_ptr(), X, Y, Z, T, sx, sy, sz, st, float(mu), float(den), X2, Y2, x_pre, y_pre,
                                             batch.data_ptr(), 0), 'zscore_pack')
    norm = normalise_intensity(img, 10.0)
    assert norm.dtype == np.float32
    padded = np.pad(norm, ((x_pre, x_post), (y_pre, y_post), (0, 0), (0, 0)), 'constant')
    want = np.transpose(padded, (3, 2, 0, 1)).reshape(T * Z, X2, Y2)
    np.testing.assert_array_equal(batch.cpu().numpy(), want)
