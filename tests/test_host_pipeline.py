"""CPU tests of the host side: NIfTI I/O, absl-style flags, the deployment loop
and its file contract, with a stub in place of the device forward."""
import gzip
import os
import struct

import numpy as np
import pytest

from oracle import fcn_oracle as O
from ukbb_cardiac_amd import deploy_network as DN, deploy_network_ao as DA, nifti, pipeline
from ukbb_cardiac_amd.flags import FlagError


# ---- NIfTI ------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', [np.float32, np.float64, np.int32, np.int16, np.uint8])
def test_nifti_roundtrip(tmp_path, dtype):
    rng = np.random.default_rng(0)
    data = (rng.random((7, 5, 3, 4)) * 100).astype(dtype)
    affine = np.array([[-1.8, 0.1, 0, 90.5], [0.05, 1.8, -0.2, -80], [0, 0.3, 10.0, 12], [0, 0, 0, 1]])
    pixdim = np.array([1, 1.8, 1.8, 10.0, 0.03125, 0, 0, 0], np.float32)
    p = str(tmp_path / 'x.nii.gz')
    nifti.save(data, p, affine, pixdim)
    im = nifti.load(p)
    assert im.data.dtype == np.dtype(dtype) and im.data.shape == data.shape
    assert np.array_equal(im.data, data)
    assert np.allclose(im.affine, affine, atol=1e-5)
    assert np.array_equal(im.header['pixdim'], pixdim)
    raw = gzip.open(p, 'rb').read()
    assert struct.unpack('<i', raw[:4])[0] == 348 and raw[344:348] == b'n+1\x00'
    assert struct.unpack('<8h', raw[40:56])[:5] == (4, 7, 5, 3, 4)
    # Fortran order on disk: x fastest
    first = np.frombuffer(raw, np.dtype(dtype).newbyteorder('<'), count=7, offset=352)
    assert np.array_equal(first, data[:, 0, 0, 0])


def test_nifti_qform_and_scaling(tmp_path):
    data = np.arange(24, dtype=np.int16).reshape(2, 3, 4)
    hdr = bytearray(348)
    struct.pack_into('>i', hdr, 0, 348)                       # big-endian file
    struct.pack_into('>8h', hdr, 40, 3, 2, 3, 4, 1, 1, 1, 1)
    struct.pack_into('>2h', hdr, 70, 4, 16)
    struct.pack_into('>8f', hdr, 76, 1, 2, 3, 4, 1, 0, 0, 0)
    struct.pack_into('>3f', hdr, 108, 352, 2.0, 1.0)          # value = raw*2 + 1
    struct.pack_into('>2h', hdr, 252, 1, 0)                   # qform only
    struct.pack_into('>6f', hdr, 256, 0, 0, 0, 5, 6, 7)       # identity rotation
    hdr[344:348] = b'n+1\x00'
    p = str(tmp_path / 'q.nii')
    with open(p, 'wb') as f:
        f.write(bytes(hdr) + b'\0' * 4 + data.astype('>i2').tobytes(order='F'))
    im = nifti.load(p)
    assert np.array_equal(im.data, data * 2.0 + 1.0)
    assert np.allclose(im.affine, [[2, 0, 0, 5], [0, 3, 0, 6], [0, 0, 4, 7], [0, 0, 0, 1]])


def test_nifti_rejects_garbage(tmp_path):
    p = str(tmp_path / 'bad.nii.gz')
    with gzip.open(p, 'wb') as f:
        f.write(b'\0' * 400)
    with pytest.raises(ValueError):
        nifti.load(p)


# ---- flags ------------------------------------------------------------------------
def test_flags_reference_command_lines():
    fs = DN.define_flags()
    F, _ = fs.parse('--seq_name sa --data_dir demo_image --model_path trained_model/FCN_sa'.split())
    assert (F.seq_name, F.data_dir, F.model_path, F.process_seq, F.save_seg, F.seg4) == \
        ('sa', 'demo_image', 'trained_model/FCN_sa', True, True, False)
    F, _ = fs.parse('--seq_name la_4ch --data_dir d --seg4 --model_path m'.split())   # demo_pipeline.py:95-96
    assert F.seg4 is True and F.seq_name == 'la_4ch'
    F, _ = fs.parse(['--seq_name=la_2ch', '--noprocess_seq', '--save_seg=false', '-data_dir', 'x'])
    assert F.seq_name == 'la_2ch' and F.process_seq is False and F.save_seg is False and F.data_dir == 'x'
    with pytest.raises(FlagError):
        fs.parse(['--seq_name', 'ao'])                        # not in the enum
    with pytest.raises(FlagError):
        fs.parse(['--bogus', '1'])
    fa = DA.define_flags()
    F, _ = fa.parse('--seq_name ao --data_dir demo_image --model_path trained_model/UNet-LSTM_ao'.split())
    assert F.model == 'UNet-LSTM' and F.z_score is True and F.weight_R == 5 and F.weight_r == 0.1


# ---- deployment loop ----------------------------------------------------------------
def stub_forward(batch):
    """Deterministic stand-in for the network: label = intensity band (0..3);
    'prob' is a one-hot-ish map over 3 classes."""
    x = batch[..., 0]
    pred = np.clip((x * 4).astype(np.int32), 0, 3)
    prob = np.stack([(x < -0.2), (np.abs(x) <= 0.2), (x > 0.2)], axis=-1).astype(np.float32)
    return {'pred': pred, 'prob': prob}


def make_volume(shape, seed):
    rng = np.random.default_rng(seed)
    return (1000.0 * rng.gamma(2.0, 1.0, size=shape)).astype(np.float32)


@pytest.mark.parametrize('shape', [(30, 44, 3, 5), (32, 48, 2, 4), (35, 21, 1, 6)])
def test_segment_sequence_matches_reference_loop(shape):
    vol = make_volume(shape, 1)
    a_in, b_in = vol.copy(), vol.copy()
    pred = pipeline.segment_sequence(a_in, stub_forward, batch_slices=7)
    ref_pred, ref_img, ed, es = O.deploy_sequence(b_in, lambda x: stub_forward(x)['pred'], 'sa')
    assert pred.dtype == np.float64 and np.array_equal(pred, ref_pred)
    assert np.array_equal(a_in, b_in)                         # both clipped the caller's array in place
    assert pipeline.pick_ed_es(pred, 'sa') == (ed, es)


def test_pad_amounts_known_answers():
    assert pipeline.pad_amounts(162, 204) == (176, 208, 7, 7, 2, 2)
    assert pipeline.pad_amounts(163, 205) == (176, 208, 6, 7, 1, 2)
    assert pipeline.pad_amounts(192, 208) == (192, 208, 0, 0, 0, 0)
    assert pipeline.pad_amounts_fixed(240, 196) == (256, 256, 8, 8, 30, 30)
    with pytest.raises(ValueError):
        pipeline.pad_amounts_fixed(260, 100)


def _write_subject(root, name, seq, shape, seed):
    d = root / name
    d.mkdir()
    affine = np.diag([1.8, 1.8, 10.0, 1.0]); affine[:3, 3] = [-10, 5, 3]
    pixdim = np.array([1, 1.8, 1.8, 10.0, 0.03, 0, 0, 0], np.float32)
    nifti.save(make_volume(shape, seed), str(d / (seq + '.nii.gz')), affine, pixdim)
    return d, affine, pixdim


@pytest.mark.parametrize('seq,seg4,prefix', [('sa', False, 'seg'), ('la_4ch', True, 'seg4'), ('la_2ch', False, 'seg')])
def test_deploy_file_contract(tmp_path, seq, seg4, prefix):
    d, affine, pixdim = _write_subject(tmp_path, 'subj1', seq, (30, 44, 2, 5), 3)
    (tmp_path / 'subj0_no_image').mkdir()
    argv = ['--seq_name', seq, '--data_dir', str(tmp_path), '--model_path', 'unused'] + (['--seg4'] if seg4 else [])
    F, _ = DN.define_flags().parse(argv)
    logs = []
    done = DN.run(F, stub_forward, log=logs.append)
    assert done == ['subj1']
    assert any('does not contain an image' in l for l in logs)          # print-and-continue (:73-76)
    names = sorted(os.listdir(d))
    assert names == sorted([seq + '.nii.gz', '%s_%s.nii.gz' % (prefix, seq), seq + '_ED.nii.gz', seq + '_ES.nii.gz',
                            '%s_%s_ED.nii.gz' % (prefix, seq), '%s_%s_ES.nii.gz' % (prefix, seq)])
    seg = nifti.load(str(d / ('%s_%s.nii.gz' % (prefix, seq))))
    assert seg.data.dtype == np.float64 and seg.data.shape == (30, 44, 2, 5)
    assert np.allclose(seg.affine, affine) and np.array_equal(seg.header['pixdim'], pixdim)
    vol = make_volume((30, 44, 2, 5), 3)
    ref_pred, ref_img, ed, es = O.deploy_sequence(vol, lambda x: stub_forward(x)['pred'], seq, seg4)
    assert np.array_equal(seg.data, ref_pred)
    es_img = nifti.load(str(d / (seq + '_ES.nii.gz')))
    assert es_img.data.dtype == np.float32 and np.array_equal(es_img.data, ref_img[:, :, :, es])   # clipped frames
    es_seg = nifti.load(str(d / ('%s_%s_ES.nii.gz' % (prefix, seq))))
    assert np.array_equal(es_seg.data, ref_pred[:, :, :, es])
    # second run: everything already there -> skipped (idempotent, :62-67)
    assert DN.run(F, stub_forward, log=lambda *_: None) == []


def test_deploy_ed_es_mode(tmp_path):
    d = tmp_path / 's'
    d.mkdir()
    aff = np.eye(4)
    for fr, seed in (('ED', 1), ('ES', 2)):
        nifti.save(make_volume((20, 24, 3), seed), str(d / ('sa_%s.nii.gz' % fr)), aff)
    F, _ = DN.define_flags().parse(['--data_dir', str(tmp_path), '--noprocess_seq'])
    DN.run(F, stub_forward, log=lambda *_: None)
    seg = nifti.load(str(d / 'seg_sa_ED.nii.gz'))
    assert seg.data.dtype == np.int32 and seg.data.shape == (20, 24, 3)     # int32 in this mode (App. C.3)
    v = make_volume((20, 24, 3), 1)
    assert np.array_equal(seg.data, pipeline.segment_frame(v, stub_forward))


def test_aortic_deploy(tmp_path):
    d, affine, pixdim = _write_subject(tmp_path, 'a1', 'ao', (40, 36, 1, 6), 5)
    F, _ = DA.define_flags().parse(['--data_dir', str(tmp_path), '--model', 'UNet', '--model_path', 'x'])
    DA.run(F, stub_forward, log=lambda *_: None)
    seg = nifti.load(str(d / 'seg_ao.nii.gz'))
    assert seg.data.dtype == np.int32 and seg.data.shape == (40, 36, 1, 6)
    assert np.array_equal(seg.header['pixdim'], pixdim)
    vol = make_volume((40, 36, 1, 6), 5)
    norm = O.normalise_intensity(vol, 10.0)
    want = np.argmax(np.stack([(norm < -0.2), (np.abs(norm) <= 0.2), (norm > 0.2)], -1), -1).astype(np.int32)
    assert np.array_equal(seg.data, want)
    F2, _ = DA.define_flags().parse(['--data_dir', str(tmp_path)])          # default model UNet-LSTM
    with pytest.raises(ValueError):                                          # ... needs the windowed forward
        DA.run(F2, stub_forward, log=lambda *_: None)
    F3, _ = DA.define_flags().parse(['--data_dir', str(tmp_path), '--model', 'Temporal-UNet'])
    with pytest.raises(NotImplementedError):
        DA.run(F3, stub_forward, log=lambda *_: None)
    F4, _ = DA.define_flags().parse(['--data_dir', str(tmp_path), '--time_step', '0'])
    with pytest.raises(ValueError):
        DA.run(F4, stub_forward, log=lambda *_: None, cine_forward=lambda *a: None)
    # --time_step reaches the windowed forward (deploy_network_ao.py:26,147)
    seen = []

    def cine(frames, weight_R, weight_r, time_step=1):
        seen.append((frames.shape, weight_R, weight_r, time_step))
        return np.zeros(frames.shape + (3,), np.float32)
    os.remove(str(d / 'seg_ao.nii.gz'))
    F5, _ = DA.define_flags().parse(['--data_dir', str(tmp_path), '--time_step', '3', '--weight_R', '5'])
    DA.run(F5, stub_forward, log=lambda *_: None, cine_forward=cine)
    assert seen == [((6, 256, 256), 5, 0.1, 3)]


def test_product_path_does_not_import_oracle():
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ukbb_cardiac_amd')
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(('.py', '.cpp', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f
                assert 'fcn_oracle' not in src.replace('oracle/fcn_oracle.py', ''), f


def test_aortic_lstm_sequence_vs_oracle_loop():
    """'UNet-LSTM' branch: pad to 256, per-slice circular windows, weighted tiling, crop -- the host mirror
    (pipeline.aortic_lstm_prob_sequence + a per-slice cine function) against the restatement of
    deploy_network_ao.py:99-107,129-183 (oracle) with the same stand-in network."""
    from oracle import fcn_oracle as O
    from ukbb_cardiac_amd import pipeline
    rng = np.random.default_rng(21)
    X, Y, Z, T = 20, 31, 2, 11
    image = (100 * rng.gamma(2.0, 1.0, size=(X, Y, Z, T))).astype(np.float32)

    def forward_seq(image_idx):                      # [N,9,256,256,1] -> prob [N,9,256,256,3], depends on the whole window
        x = image_idx[..., 0].astype(np.float64)
        ctx = x.mean(axis=1, keepdims=True) + 0.1 * np.arange(x.shape[1])[None, :, None, None]
        lg = np.stack([x, ctx - x, 0.5 * ctx], axis=-1)
        return O.softmax(lg).astype(np.float32)

    def cine_forward(frames, weight_R, weight_r):    # independent per-slice tiling in plain numpy
        F = frames.shape[0]
        w = O.aortic_window_weights(weight_R, weight_r)
        acc = np.zeros((F,) + frames.shape[1:] + (3,), np.float32)
        ws = np.zeros(F)
        for t in range(F):
            idx = O.aortic_window_indices(t, F, weight_R)
            p = forward_seq(frames[idx][None, ..., None])[0]
            for k, f in enumerate(idx):
                acc[f] = (acc[f].astype(np.float64) + p[k].astype(np.float64) * w[k]).astype(np.float32)
                ws[f] += w[k]
        return (acc.astype(np.float64) / ws[:, None, None, None]).astype(np.float32)

    got = pipeline.aortic_lstm_prob_sequence(image.copy(), cine_forward, z_score=True)
    from ukbb_cardiac_amd.image_utils import normalise_intensity
    want = O.aortic_lstm_prob_sequence(normalise_intensity(image.copy(), 10.0), forward_seq)
    assert got.shape == (X, Y, Z, T, 3) and got.dtype == np.float32
    np.testing.assert_array_equal(got, want)


def test_rescale_numpy1_casting_is_float32_arithmetic_within_one_ulp():
    """numpy 1.x (the reference's era) keeps float32_array - float64_scalar in float32; numpy 2 (this image,
    the goldens) computes in float64.  The opt-in variant reproduces the former; both clip identically."""
    from ukbb_cardiac_amd.image_utils import rescale_intensity
    v = make_volume((40, 36, 3, 4), 9)
    a, b = v.copy(), v.copy()
    r2 = rescale_intensity(a, (1, 99))
    r1 = rescale_intensity(b, (1, 99), numpy1_casting=True)
    assert r1.dtype == np.float32 and r2.dtype == np.float64 and np.array_equal(a, b)
    lo, hi = np.percentile(v, (1, 99))
    want = (b.astype(np.float32) - np.float32(lo)) / np.float32(hi - lo)      # all-float32 evaluation
    assert np.array_equal(r1, want)
    d = np.abs(r1.astype(np.float64) - r2.astype(np.float32))
    assert d.max() <= 2 * np.spacing(np.float32(1.0)) and (r1 != r2.astype(np.float32)).any()
    # the deploy loop accepts the switch and then stays on the host path
    p2 = pipeline.segment_sequence(v.copy(), stub_forward, numpy1_casting=True)
    assert p2.shape == v.shape


# NIfTI-1 header layout (nifti1.h of the NIfTI-1.1 standard): name -> (byte offset, struct format).  nibabel, which every
# evaluation script of the reference uses to read our outputs (short_axis/eval_ventricular_volume.py:40-52,
# aortic/eval_aortic_area.py:52-66), parses exactly this table; it is absent from this image, so the files are pinned
# against the standard itself.
NIFTI1_FIELDS = {
    'sizeof_hdr': (0, '<i'), 'dim_info': (39, '<b'), 'dim': (40, '<8h'), 'intent_code': (68, '<h'), 'datatype': (70, '<h'),
    'bitpix': (72, '<h'), 'slice_start': (74, '<h'), 'pixdim': (76, '<8f'), 'vox_offset': (108, '<f'), 'scl_slope': (112, '<f'),
    'scl_inter': (116, '<f'), 'slice_end': (120, '<h'), 'slice_code': (122, '<b'), 'xyzt_units': (123, '<b'),
    'cal_max': (124, '<f'), 'cal_min': (128, '<f'), 'descrip': (148, '<80s'), 'aux_file': (228, '<24s'),
    'qform_code': (252, '<h'), 'sform_code': (254, '<h'), 'quatern_b': (256, '<f'), 'qoffset_x': (268, '<f'),
    'srow_x': (280, '<4f'), 'srow_y': (296, '<4f'), 'srow_z': (312, '<4f'), 'intent_name': (328, '<16s'), 'magic': (344, '<4s'),
}
NIFTI1_DATATYPES = {np.uint8: (2, 8), np.int16: (4, 16), np.int32: (8, 32), np.float32: (16, 32), np.float64: (64, 64)}


@pytest.mark.parametrize('dtype', [np.float64, np.int32, np.float32, np.uint8, np.int16])
def test_nifti_header_matches_the_nifti1_field_table(tmp_path, dtype):
    """The three label-map flavours the deploy scripts write (float64 sequence volumes, int32 ED/ES + aortic,
    float32 image frames) field by field against the standard's table."""
    data = (np.arange(5 * 4 * 3 * 2).reshape(5, 4, 3, 2) % 4).astype(dtype)
    affine = np.array([[-1.8, 0.0, 0.1, 90.0], [0.0, 1.8, 0.2, -80.0], [0.05, 0.0, 10.0, 12.0], [0, 0, 0, 1]])
    pixdim = np.array([-1, 1.8, 1.8, 10.0, 0.03, 0, 0, 0], np.float32)
    p = str(tmp_path / 'f.nii.gz')
    nifti.save(data, p, affine, pixdim)
    raw = gzip.open(p, 'rb').read()
    f = {k: struct.unpack_from(fmt, raw, off) for k, (off, fmt) in NIFTI1_FIELDS.items()}
    assert f['sizeof_hdr'] == (348,) and f['magic'] == (b'n+1\x00',)            # single-file NIfTI-1
    assert f['dim'] == (4, 5, 4, 3, 2, 1, 1, 1)                                   # unused dims are 1, as the standard asks
    assert (f['datatype'][0], f['bitpix'][0]) == NIFTI1_DATATYPES[dtype]
    assert f['vox_offset'] == (352.0,) and len(raw) == 352 + data.nbytes          # 348 + 4 extension-flag bytes, all zero
    assert raw[348:352] == b'\x00\x00\x00\x00'
    assert np.allclose(f['pixdim'], pixdim)                                       # the reference copies the input's pixdim (:142)
    assert f['scl_slope'] == (1.0,) and f['scl_inter'] == (0.0,)                  # stored values are the values
    assert f['sform_code'] == (2,) and f['qform_code'] == (0,)                    # nibabel's Nifti1Image(data, affine): sform 'aligned', qform 'unknown'
    assert np.allclose(f['srow_x'] + f['srow_y'] + f['srow_z'], affine[:3].ravel(), atol=1e-6)
    assert f['intent_code'] == (0,) and f['slice_code'] == (0,) and f['dim_info'] == (0,)
    vox = np.frombuffer(raw, np.dtype(dtype).newbyteorder('<'), offset=352).reshape(data.shape, order='F')   # x fastest
    assert np.array_equal(vox, data)
    # and the reader takes the same table: a header assembled from the table alone round-trips
    hdr = bytearray(348)
    for k, v in (('sizeof_hdr', (348,)), ('dim', (3, 2, 3, 4, 1, 1, 1, 1)), ('datatype', (16,)), ('bitpix', (32,)),
                 ('pixdim', (1, 2, 3, 4, 0, 0, 0, 0)), ('vox_offset', (352,)), ('scl_slope', (0.0,)), ('sform_code', (1,)),
                 ('srow_x', (2, 0, 0, -5)), ('srow_y', (0, 3, 0, -6)), ('srow_z', (0, 0, 4, -7)), ('magic', (b'n+1\x00',))):
        struct.pack_into(NIFTI1_FIELDS[k][1], hdr, NIFTI1_FIELDS[k][0], *v)
    vol = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    q = str(tmp_path / 'g.nii')
    open(q, 'wb').write(bytes(hdr) + b'\0' * 4 + vol.tobytes(order='F'))
    im = nifti.load(q)
    assert np.array_equal(im.data, vol) and np.allclose(im.affine, [[2, 0, 0, -5], [0, 3, 0, -6], [0, 0, 4, -7], [0, 0, 0, 1]])


def test_nifti_save_as_dtype_streams_the_same_bytes(tmp_path):
    """The writer threads of the subject pipeline hand over uint8 labels; the file must be the float64 volume the
    reference writes (deploy_network.py:92,136), byte for byte."""
    lab = (np.random.default_rng(3).integers(0, 4, size=(13, 11, 3, 5))).astype(np.uint8)
    aff = np.diag([1.8, 1.8, 10.0, 1.0])
    pd = np.array([1, 1.8, 1.8, 10.0, 0.03, 0, 0, 0], np.float32)
    a, b = str(tmp_path / 'a.nii.gz'), str(tmp_path / 'b.nii.gz')
    f64 = np.zeros(lab.shape); f64[...] = lab
    nifti.save(f64, a, aff, pd)
    nifti.save(np.asfortranarray(lab), b, aff, pd, as_dtype=np.float64)
    assert open(a, 'rb').read() == open(b, 'rb').read()
    assert nifti.load(b).data.dtype == np.float64
    nifti.save(lab[:, :, 0, 0], b, aff, as_dtype=np.int32)
    nifti.save(lab[:, :, 0, 0].astype(np.int32), a, aff)
    assert open(a, 'rb').read() == open(b, 'rb').read()


# ---- label volumes through the run-length gzip writer of the C library (include/ukbb_fcn.h: ukbb_fcn_gzip_labels) ----------

def _blobs(shape, seed):
    rng = np.random.default_rng(seed)
    lab = np.zeros(shape, np.uint8)
    idx = np.indices(shape[:2])
    for k in range(1, 4):
        cy, cx, r = rng.integers(10, shape[0] - 10), rng.integers(10, shape[1] - 10), rng.integers(3, 12)
        m = (idx[0] - cy) ** 2 + (idx[1] - cx) ** 2 < r * r
        lab[m] = k
    return lab


@pytest.mark.parametrize('dtype', [np.uint8, np.int16, np.int32, np.float32, np.float64])
def test_label_gzip_inflates_to_what_zlib_path_writes(tmp_path, dtype, monkeypatch):
    """deploy_network.py:136-138 / deploy_network_ao.py:189-196: the segmentation file.  The run-length writer and the zlib
    path must give the same bytes after inflation (gzip.open also checks the member's CRC-32 and length)."""
    import gzip
    from ukbb_cardiac_amd import nifti
    lab = _blobs((48, 40, 3, 4), 3)
    aff = np.diag([1.8, 1.8, 10.0, 1.0])
    fast, slow = str(tmp_path / 'fast.nii.gz'), str(tmp_path / 'slow.nii.gz')
    nifti.save(lab.astype(dtype), fast, aff)
    monkeypatch.setattr(nifti, 'LABEL_FAST_PATH', False)
    nifti.save(lab.astype(dtype), slow, aff)
    a, b = gzip.open(fast, 'rb').read(), gzip.open(slow, 'rb').read()
    assert a == b and len(a) == 352 + lab.size * np.dtype(dtype).itemsize
    assert open(fast, 'rb').read() != open(slow, 'rb').read()              # it really was the other encoder
    back = nifti.load(fast)
    assert back.get_data().dtype == dtype and np.array_equal(back.get_data(), lab)


@pytest.mark.parametrize('n', [1, 2, 3, 33, 34, 259, 260, 261, 262, 516, 517, 30000])
@pytest.mark.parametrize('fill', ['zero', 'one', 'random', 'blocks'])
def test_label_gzip_run_lengths_and_crc(tmp_path, n, fill, monkeypatch):
    """Run lengths around the 258-byte match limit, all-zero volumes (CRC of zero runs by polynomial shift), noise."""
    import gzip
    from ukbb_cardiac_amd import nifti
    rng = np.random.default_rng(n)
    lab = {'zero': np.zeros(n, np.uint8), 'one': np.ones(n, np.uint8), 'random': rng.integers(0, 4, n).astype(np.uint8),
           'blocks': np.repeat(rng.integers(0, 4, n // 7 + 1), 7)[:n].astype(np.uint8)}[fill].reshape(n, 1, 1)
    p = str(tmp_path / 'l.nii.gz')
    nifti.save(lab, p, np.eye(4), as_dtype=np.float64)                     # the sequence-mode form: uint8 labels -> float64 file
    got = gzip.open(p, 'rb').read()
    assert got[352:] == lab.astype('<f8').tobytes(order='F')


def test_label_gzip_same_bytes_from_uint8_and_float64_volumes(tmp_path):
    """Device path (uint8 labels, as_dtype=float64) and host path (float64 volume) write the same FILE."""
    from ukbb_cardiac_amd import nifti
    lab = np.asfortranarray(_blobs((64, 48, 2, 5), 9))
    aff = np.diag([1.8, 1.8, 10.0, 1.0])
    nifti.save(lab, str(tmp_path / 'a.nii.gz'), aff, as_dtype=np.float64)
    vol = np.zeros(lab.shape)
    vol[...] = lab
    nifti.save(vol, str(tmp_path / 'b.nii.gz'), aff)
    assert (tmp_path / 'a.nii.gz').read_bytes() == (tmp_path / 'b.nii.gz').read_bytes()


def test_label_gzip_leaves_images_and_odd_values_to_zlib(tmp_path):
    from ukbb_cardiac_amd import nifti
    for data in (np.random.default_rng(0).random((16, 16, 2)).astype(np.float32) * 900,     # MR intensities
                 np.array([[[0.0, 1.0, 2.5]]]), np.array([[[0, -1, 3]]], np.int32), np.array([[[0.0, np.nan]]])):
        assert nifti._as_label_volume(data) is None
        p = str(tmp_path / 'x.nii.gz')
        nifti.save(data, p, np.eye(4))
        assert np.array_equal(nifti.load(p).get_data(), data, equal_nan=True)


def test_label_gzip_reports_a_short_buffer():
    import ctypes as C
    from ukbb_cardiac_amd import _lib
    lab = np.random.default_rng(1).integers(0, 4, 5000).astype(np.uint8)
    out = np.empty(64, np.uint8)
    assert _lib.lib.ukbb_fcn_gzip_labels(lab.ctypes.data, lab.size, 64, b'', 0, out.ctypes.data, out.size) == -4      # UKBB_ENOMEM
    cap = _lib.lib.ukbb_fcn_gzip_labels_bound(lab.size, 64, 0)
    out = np.empty(cap, np.uint8)
    got = _lib.lib.ukbb_fcn_gzip_labels(lab.ctypes.data, lab.size, 64, b'', 0, out.ctypes.data, cap)
    assert 0 < got <= cap
    import gzip
    assert gzip.decompress(out[:got].tobytes()) == lab.astype('<f8').tobytes()
    assert _lib.lib.ukbb_fcn_gzip_labels(lab.ctypes.data, lab.size, 1024, b'', 0, out.ctypes.data, cap) == -1         # int64: not offered


# ---- whole-file gzip decoder (csrc/gz_inflate.cpp, ukbb_fcn_gunzip) ------------------------------------------------------------

def _gunzip(blob, cap, verify=1):
    from ukbb_cardiac_amd import _labelgz
    src = np.frombuffer(blob, np.uint8) if len(blob) else np.zeros(1, np.uint8)
    dst = np.empty(cap + 1, np.uint8)
    dst[cap] = 0xA5                                                           # canary behind the capacity
    r = int(_labelgz.lib.ukbb_fcn_gunzip(src.ctypes.data, len(blob), dst.ctypes.data, cap, verify))
    assert dst[cap] == 0xA5
    return r, dst[:max(r, 0)].tobytes()


def _deflate(raw, level=6, strategy=0):
    import zlib
    co = zlib.compressobj(level, zlib.DEFLATED, 31, 9, strategy)
    return co.compress(raw) + co.flush()


def test_gunzip_inflates_what_zlib_writes_at_every_level_and_strategy():
    """Stored, fixed-Huffman (Z_FIXED), dynamic blocks, Huffman-only, RLE; match distances 1..32768 (every copy width of the
    decoder); output buffers of exactly the content's size (the last 320 bytes go through the careful loop) and larger."""
    import zlib
    rng = np.random.default_rng(3)
    cases = [b'', b'a', b'hello hello hello hello', bytes(200000), rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
             rng.integers(0, 4, 120000, dtype=np.uint8).tobytes(), np.cumsum(rng.normal(0, 3, 150000)).astype(np.int16).tobytes(),
             np.round(rng.gamma(2.0, 300.0, 60000)).astype(np.float32).tobytes(),
             b''.join(b'the quick brown fox %d jumps over the lazy dog\n' % i for i in range(6000))]
    for d in (1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 31, 33, 258, 259, 4097, 32768):
        cases.append((rng.integers(0, 256, d, dtype=np.uint8).tobytes() * (90000 // d + 2))[:90000])
    for raw in cases:
        for level, strategy in ((1, 0), (6, 0), (9, 0), (0, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            g = _deflate(raw, level, strategy)
            for cap in (len(raw), len(raw) + 777):
                r, out = _gunzip(g, cap)
                assert r == len(raw) and out == raw, (len(raw), level, strategy, cap, r)
            if raw:
                assert _gunzip(g, len(raw) - 1)[0] == -4                      # UKBB_ENOMEM: does not fit


def test_gunzip_members_padding_names_and_what_it_refuses():
    import gzip
    import io
    rng = np.random.default_rng(4)
    a, b = rng.integers(0, 7, 50000, dtype=np.uint8).tobytes(), np.cumsum(rng.normal(0, 2, 40000)).astype(np.int16).tobytes()
    g = _deflate(a) + b'\x00' * 5 + _deflate(b, 1) + b'\x00' * 100
    assert _gunzip(g, len(a) + len(b)) == (len(a) + len(b), a + b)
    bio = io.BytesIO()
    with gzip.GzipFile(filename='cine.nii', mode='wb', fileobj=bio, mtime=1234) as f:     # FNAME + MTIME
        f.write(b)
    assert _gunzip(bio.getvalue(), len(b)) == (len(b), b)
    g = _deflate(a)
    assert _gunzip(g + b'junk', len(a))[0] == -1                              # trailing bytes that are no member
    assert _gunzip(g[:3] + bytes([g[3] | 2]) + g[4:], len(a))[0] == -1       # FHCRC: left to zlib
    assert _gunzip(b'\x1f\x8b\x07' + g[3:], len(a))[0] == -1                  # not deflate
    for cut in list(range(0, 30)) + list(range(len(g) - 30, len(g))):
        assert _gunzip(g[:cut], len(a))[0] < 0, cut
    bad = bytearray(g); bad[-6] ^= 0x10                                       # CRC-32
    assert _gunzip(bytes(bad), len(a))[0] == -1 and _gunzip(bytes(bad), len(a), verify=0) == (len(a), a)
    bad = bytearray(g); bad[-2] ^= 0x10                                       # ISIZE
    assert _gunzip(bytes(bad), len(a))[0] == -1


def test_gunzip_agrees_with_zlib_on_corrupted_and_random_streams():
    """Whatever it accepts, zlib accepts with the same bytes (a flipped distance bit that lands on an identical run, padding bits);
    it never writes behind the capacity, never crashes (run under the address sanitiser by the test below)."""
    import zlib
    from ukbb_cardiac_amd import _labelgz
    rng = np.random.default_rng(5)
    raw = np.cumsum(rng.normal(0, 3, 60000)).astype(np.int16).tobytes()
    g = _deflate(raw)
    accepted = 0
    for _ in range(4000):
        bad = bytearray(g)
        for _ in range(int(rng.integers(1, 3))):
            bad[int(rng.integers(10, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        r, out = _gunzip(bytes(bad), len(raw))
        if r >= 0:
            accepted += 1
            z = zlib.decompressobj(31)
            assert z.decompress(bytes(bad)) == out and z.eof
    hdr = b'\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03'
    for i in range(20000):
        body = rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8).tobytes()
        if i % 3 == 0:
            body = bytes([int(rng.integers(0, 8)) | 4]) + body                 # more dynamic-block headers
        r, out = _gunzip(hdr + body, int(rng.integers(0, 3000)))
        if r >= 0:
            z = zlib.decompressobj(31)
            assert z.decompress(hdr + body) == out and z.eof
    for n in list(range(0, 200)) + [1000, 4096, 65537, 1 << 20]:              # CRC-32: table path, carry-less path, every tail length
        d = rng.integers(0, 256, n, dtype=np.uint8)
        for init in (0, 0x12345678):
            assert _labelgz.lib.ukbb_fcn_gzip_crc(init, d.ctypes.data, n) == zlib.crc32(d.tobytes(), init)


def test_gunzip_never_writes_behind_the_capacity_on_long_matches():
    """the fast loop's slack check has to hold for the position AFTER a match.  Streams of back-to-back 258-byte
    matches whose content is longer than the capacity (-> UKBB_ENOMEM, nothing stored behind dst + cap), exact-size buffers followed
    by zero padding or a second member, and every capacity around the end of a run.  The buffers here are exactly cap (+ 1 canary)
    bytes, so under the address sanitiser (test_label_gzip_clean_under_address_and_ub_sanitizers) any overshoot aborts."""
    rng = np.random.default_rng(11)
    head = rng.integers(0, 256, 2000, dtype=np.uint8).tobytes()
    for run in (bytes(200000), b'\x01\x02' * 100000, b'abcd' * 50000, rng.integers(0, 256, 9, dtype=np.uint8).tobytes() * 22000,
                rng.integers(0, 256, 300, dtype=np.uint8).tobytes() * 700):
        raw = head + run
        for level in (1, 6, 9):
            g = _deflate(raw, level)
            for cap in (0, 1, 319, 320, 321, 600, 2000, 2322, 2322 + 258, 5000, len(raw) - 259, len(raw) - 17, len(raw) - 1):
                assert _gunzip(g, cap)[0] == -4, (len(run), level, cap)
            for extra in (b'', bytes(64), bytes(1000)):
                assert _gunzip(g + extra, len(raw)) == (len(raw), raw)
            # second member: the first one ends exactly at its share of the buffer, the second does not fit / fits exactly
            g2 = _deflate(run[:70000], level)
            assert _gunzip(g + g2, len(raw) + 70000) == (len(raw) + 70000, raw + run[:70000])
            for short in (1, 16, 200, 320, 69999):
                assert _gunzip(g + g2, len(raw) + 70000 - short)[0] == -4
    # flipped header bits: whenever the strict decoder accepts a damaged stream, zlib accepts it too and inflates the same bytes
    import zlib
    g = bytearray(_deflate(b'abcabcabcabc' * 50 + bytes(range(256)), 9))
    for pos in range(10, min(len(g), 120)):
        for bit in range(8):
            bad = bytes(g[:pos]) + bytes([g[pos] ^ (1 << bit)]) + bytes(g[pos + 1:])
            r, out = _gunzip(bad, 4096, verify=0)
            if r >= 0:
                z = zlib.decompressobj(-15)
                try:
                    ref = z.decompress(bad[10:])
                except zlib.error:
                    ref = None
                assert ref is not None and ref[:r] == out, (pos, bit)


def _dynamic_block_with_one_distance_code(dist_len):
    """A hand-assembled gzip member: one dynamic-Huffman block (RFC 1951 3.2.7) with HDIST = 1 distance code of length `dist_len`,
    literal/length codes {257 (length 3): 1 bit, 'a': 2 bits, end-of-block: 2 bits}; tokens: 'a', match(length 3, distance 1), EOB
    -> b'aaaa'."""
    import struct
    import zlib
    bits = []

    def put(v, n):                                            # plain fields: least-significant bit first
        bits.extend((v >> i) & 1 for i in range(n))

    def huff(code, n):                                        # Huffman codes: most-significant bit first
        bits.extend((code >> (n - 1 - i)) & 1 for i in range(n))
    put(1, 1); put(2, 2)                                      # BFINAL, BTYPE = dynamic
    put(258 - 257, 5); put(1 - 1, 5); put(18 - 4, 4)          # HLIT: 258 lit/len codes, HDIST: 1 distance code, HCLEN: 18 lengths
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    cl_len = {18: 1, 1: 2, 2: 2}                              # code-length code: 18 -> 0, 1 -> 10, 2 -> 11
    for sym in order[:18]:
        put(cl_len.get(sym, 0), 3)
    cl_code = {18: (0, 1), 1: (2, 2), 2: (3, 2)}

    def zeros(n):                                             # symbol 18: 11..138 zeros, 7 extra bits
        huff(*cl_code[18]); put(n - 11, 7)
    zeros(97); huff(*cl_code[2])                              # lengths of symbols 0..96 = 0, 'a' (97) = 2
    zeros(138); zeros(20); huff(*cl_code[2]); huff(*cl_code[1])   # 98..255 = 0, 256 = 2, 257 = 1
    huff(*cl_code[dist_len])                                  # the single distance code's length
    huff(2, 2)                                                # 'a'
    huff(0, 1); huff(0, dist_len)                             # length 3 (symbol 257, no extra bits), distance symbol 0 = distance 1
    huff(3, 2)                                                # end of block
    bits.extend([0] * (-len(bits) % 8))
    body = bytes(sum(bits[i + j] << j for j in range(8)) for i in range(0, len(bits), 8))
    raw = b'aaaa'
    return b'\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03' + body + struct.pack('<II', zlib.crc32(raw), len(raw)), body, raw


def test_gunzip_single_distance_code_rule_is_zlibs():
    """RFC 1951 allows a block with ONE distance code; zlib's inflate_table accepts it only with length 1 (an
    incomplete code of any other shape is an error), and so does csrc/gz_inflate.cpp build_table -- a longer single code would leave
    second-level table slots unwritten.  Hand-assembled streams, zlib as the referee."""
    import zlib
    g1, body1, raw = _dynamic_block_with_one_distance_code(1)
    assert zlib.decompressobj(-15).decompress(body1) == raw                   # the assembler writes what zlib reads
    assert _gunzip(g1, 16) == (4, raw) and _gunzip(g1, 4) == (4, raw)
    g2, body2, _ = _dynamic_block_with_one_distance_code(2)
    with pytest.raises(zlib.error):
        zlib.decompressobj(-15).decompress(body2)                             # "invalid distances set"
    assert _gunzip(g2, 16)[0] == -1 and _gunzip(g2, 16, verify=0)[0] == -1    # UKBB_EINVAL, nothing inflated


def test_load_takes_the_whole_file_decoder_and_falls_back_to_zlib(tmp_path):
    """nifti.load of a .nii.gz: same arrays from the whole-file decoder and from the zlib reader (plain, alloc'd, alloc'd with headroom
    = decoded in place, big-endian, scaled); streams the decoder refuses still load -- or raise -- through zlib as before."""
    import gzip
    import zlib
    from ukbb_cardiac_amd import nifti, _labelgz
    rng = np.random.default_rng(6)
    vol = np.round(rng.gamma(2.0, 300.0, (40, 36, 3, 7))).astype(np.float32)
    p = str(tmp_path / 'v.nii.gz')
    nifti.save(vol, p, np.eye(4))
    calls = []
    real = _labelgz.lib.ukbb_fcn_gunzip

    class Spy:                                                                # counts calls and results of the native decoder
        def __call__(self, *a):
            r = real(*a)
            calls.append(int(r))
            return r
    _labelgz.lib.ukbb_fcn_gunzip = Spy()
    try:
        assert np.array_equal(nifti.load(p).get_data(), vol) and calls == [352 + vol.nbytes]
        buf = np.empty(vol.shape, np.float32, order='F')
        assert nifti.load(p, alloc=lambda sh, dt: buf).get_data() is buf and np.array_equal(buf, vol)
        # headroom: the header lands in front of the array, the voxels in it, nothing behind it
        flat = np.full(1024 + vol.size + 16, np.float32(-7.0))
        view = flat[1024:1024 + vol.size].reshape(vol.shape, order='F')
        out = nifti.load(p, alloc=lambda sh, dt: (view, 4096)).get_data()
        assert out is view and np.array_equal(view, vol) and np.all(flat[1024 + vol.size:] == -7.0) and np.all(flat[:1024 - 88] == -7.0)
        assert bytes(flat[1024 - 88:1024].view(np.uint8)[:4]) == (348).to_bytes(4, 'little')       # the header's sizeof_hdr
        n_native = len(calls)
        nifti.NATIVE_GUNZIP = False
        ref = nifti.load(p)
        nifti.NATIVE_GUNZIP = True
        assert len(calls) == n_native and np.array_equal(ref.get_data(), vol) and np.array_equal(ref.affine, nifti.load(p).affine)
        # int16 with scl_slope / big-endian: the conversions behind the decoder are the old ones
        raw = gzip.open(p, 'rb').read()
        hdr = bytearray(raw[:352])
        i16 = rng.integers(-300, 3000, vol.size).astype('>i2')
        import struct
        be = bytearray(hdr)
        be[0:4] = struct.pack('>i', 348)
        be[40:56] = struct.pack('>8h', 4, *vol.shape, 1, 1, 1)
        be[70:74] = struct.pack('>2h', 4, 16)
        be[76:108] = struct.pack('>8f', 1, 1, 1, 1, 1, 1, 1, 1)
        be[108:120] = struct.pack('>3f', 352, 2.0, 5.0)
        be[252:256] = struct.pack('>2h', 0, 0)
        q = str(tmp_path / 'be.nii.gz')
        open(q, 'wb').write(gzip.compress(bytes(be) + i16.tobytes()))
        got = nifti.load(q).get_data()
        nifti.NATIVE_GUNZIP = False
        want = nifti.load(q).get_data()
        nifti.NATIVE_GUNZIP = True
        assert got.dtype == want.dtype == np.float64 and np.array_equal(got, want) and got.flags.writeable
        assert np.array_equal(got, i16.astype(np.float64).reshape(vol.shape, order='F') * 2.0 + 5.0)
        # refused by the decoder, accepted by zlib: members with a tiny first one, junk behind the data
        whole = open(p, 'rb').read()
        multi = str(tmp_path / 'm.nii.gz')
        open(multi, 'wb').write(gzip.compress(raw[:100]) + gzip.compress(raw[100:]))
        assert np.array_equal(nifti.load(multi).get_data(), vol)
        junk = str(tmp_path / 'j.nii.gz')
        open(junk, 'wb').write(whole + b'not a gzip member')
        calls.clear()
        staged = []

        def alloc_once(sh, dt):                                               # the fall-back must reuse the array the first reader asked for
            staged.append(np.empty(sh, dt, order='F'))
            return staged[-1]
        assert np.array_equal(nifti.load(junk, alloc=alloc_once).get_data(), vol) and calls == [-1] and len(staged) == 1
        # refused by both
        cut = str(tmp_path / 't.nii.gz')
        open(cut, 'wb').write(whole[:len(whole) // 2])
        with pytest.raises((EOFError, ValueError)):
            nifti.load(cut)
        bad = bytearray(whole); bad[-6] ^= 0xff
        open(cut, 'wb').write(bytes(bad))
        with pytest.raises(zlib.error):
            nifti.load(cut)
    finally:
        _labelgz.lib.ukbb_fcn_gunzip = real
        nifti.NATIVE_GUNZIP = True


# ---- .gz reader: zlib fed in large pieces (nifti._GzReader) ---------------------------------------------------------------

def test_gz_reader_members_padding_truncation_and_crc(tmp_path):
    """What gzip.open accepted must still load: several members, zero padding between / after them; what it refused must still fail."""
    import gzip
    import zlib
    from ukbb_cardiac_amd import nifti
    rng = np.random.default_rng(0)
    vol = np.round(rng.gamma(2.0, 300.0, (40, 36, 3, 7))).astype(np.float32)
    p = str(tmp_path / 'v.nii.gz')
    nifti.save(vol, p, np.eye(4))
    assert np.array_equal(nifti.load(p).get_data(), vol)
    raw = gzip.open(p, 'rb').read()
    multi = str(tmp_path / 'm.nii.gz')
    with open(multi, 'wb') as f:
        f.write(gzip.compress(raw[:100]) + b'\x00' * 5 + gzip.compress(raw[100:5000]) + gzip.compress(raw[5000:]) + b'\x00\x00')
    assert np.array_equal(nifti.load(multi).get_data(), vol)
    buf = np.empty(vol.shape, np.float32, order='F')
    assert nifti.load(multi, alloc=lambda sh, dt: buf).get_data() is buf and np.array_equal(buf, vol)
    whole = open(p, 'rb').read()
    cut = str(tmp_path / 't.nii.gz')
    open(cut, 'wb').write(whole[:len(whole) // 2])
    with pytest.raises((EOFError, ValueError)):
        nifti.load(cut)
    bad = bytearray(whole)
    bad[-6] ^= 0xff                                                          # CRC-32 of the member
    open(cut, 'wb').write(bytes(bad))
    with pytest.raises(zlib.error):
        nifti.load(cut)
    junk = str(tmp_path / 'j.nii.gz')                                        # bytes behind the voxel data are never reached, as with nibabel
    open(junk, 'wb').write(whole + b'not a gzip member')
    assert np.array_equal(nifti.load(junk).get_data(), vol)


def test_gz_reader_small_pieces(tmp_path, monkeypatch):
    """Piece boundaries anywhere in the stream (header, trailer, between members)."""
    import gzip
    from ukbb_cardiac_amd import nifti
    vol = np.arange(30 * 20 * 4, dtype=np.int16).reshape(30, 20, 4)
    p = str(tmp_path / 'v.nii.gz')
    nifti.save(vol, p, np.eye(4))
    raw = gzip.open(p, 'rb').read()
    open(p, 'wb').write(gzip.compress(raw[:353]) + gzip.compress(raw[353:]))
    for piece in (1, 2, 3, 7, 64, 1000):
        monkeypatch.setattr(nifti._GzReader, 'PIECE', piece)
        assert np.array_equal(nifti.load(p).get_data(), vol)


def test_label_gzip_property_random_volumes(tmp_path):
    """Property test: for random label volumes (any run structure, any supported voxel type, 1-4 dimensions) the run-length writer's
    file inflates to header + the voxel bytes, and loads back to the array."""
    import gzip
    from hypothesis import given, settings, strategies as st
    from ukbb_cardiac_amd import nifti
    p = str(tmp_path / 'h.nii.gz')

    @settings(max_examples=120, deadline=None)
    @given(st.integers(0, 2 ** 32 - 1), st.sampled_from(['u1', 'i2', 'i4', 'f4', 'f8']), st.integers(1, 4), st.integers(1, 6),
           st.sampled_from([2, 4, 256]), st.floats(0.0, 1.0))
    def check(seed, dt, ndim, run_scale, n_labels, zero_frac):
        rng = np.random.default_rng(seed)
        shape = tuple(int(v) for v in rng.integers(1, 14, ndim))
        n = int(np.prod(shape))
        runs = rng.integers(1, 1 + 40 * run_scale, n)                       # run-structured labels
        vals = rng.integers(0, n_labels, n)
        vals[rng.random(n) < zero_frac] = 0
        lab = np.repeat(vals, runs)[:n].astype(np.uint8).reshape(shape, order='F')
        data = lab.astype(dt)
        nifti.save(data, p, np.eye(4))
        raw = gzip.open(p, 'rb').read()
        assert raw[352:] == data.astype('<' + dt).tobytes(order='F')
        assert np.array_equal(nifti.load(p).get_data(), data)

    check()


def test_aortic_deploy_read_ahead_threads_change_nothing(tmp_path):
    """deploy_network_ao --io_threads: several subjects (one directory without a cine, one stray file) with the stand-in network;
    files and log lines equal those of the strictly sequential loop."""
    import shutil
    src = tmp_path / 'src'
    src.mkdir()
    for i in range(6):
        if i == 2:
            (src / 'b2').mkdir()                                             # no ao.nii.gz inside
            continue
        _write_subject(src, 'a%d' % i, 'ao', (24 + 2 * i, 20, 1, 4 + i), 10 + i)
    (src / 'notes.txt').write_text('not a subject directory')
    out = {}
    for thr in (0, 1, 4):
        work = tmp_path / ('run%d' % thr)
        shutil.copytree(src, work)
        lines = []
        F, _ = DA.define_flags().parse(['--data_dir', str(work), '--model', 'UNet', '--model_path', 'x', '--io_threads', str(thr)])
        done = DA.run(F, stub_forward, log=lines.append)
        out[thr] = (done, {p.relative_to(work).as_posix(): p.read_bytes() for p in sorted(work.rglob('seg_ao.nii.gz'))},
                    [l.replace(str(work), 'DIR') for l in lines if 'time' not in l and 'took' not in l])
    assert out[0][0] == ['a0', 'a1', 'a3', 'a4', 'a5'] and len(out[0][1]) == 5
    assert out[0] == out[1] == out[4]


# ---- r03: label gzip modes, atomic output files, --output_csv ---------------------------------------------------------------

@pytest.mark.parametrize('dtype', [np.uint8, np.int16, np.int32, np.float32, np.float64])
def test_label_gzip_modes_inflate_identically_and_small_is_not_larger_than_zlib(tmp_path, dtype):
    """--label_gzip small / fast / zlib: the same inflated bytes; 'small' (dynamic Huffman codes from the exact token
    histogram) stays within 1.2x of zlib level 1 on a blob segmentation and well below 'fast'."""
    import gzip
    from ukbb_cardiac_amd import nifti
    lab = _blobs((96, 80, 4, 6), 5)
    sizes, raws = {}, {}
    try:
        for mode in nifti.LABEL_GZIP_MODES:
            nifti.set_label_gzip(mode)
            p = str(tmp_path / ('%s.nii.gz' % mode))
            nifti.save(lab.astype(dtype), p, np.eye(4))
            sizes[mode], raws[mode] = os.path.getsize(p), gzip.open(p, 'rb').read()
    finally:
        nifti.set_label_gzip('small')
    assert raws['small'] == raws['fast'] == raws['zlib']
    assert sizes['small'] <= 1.2 * sizes['zlib'], sizes
    assert sizes['small'] < sizes['fast'], sizes
    with pytest.raises(ValueError):
        nifti.set_label_gzip('tiny')


def test_label_gzip_dynamic_codes_on_degenerate_histograms(tmp_path):
    """Length-limited code construction at its corners: one label only (two-symbol alphabets), all 256 labels once (flat
    histogram), a geometric histogram deep enough to need the 15-bit limit, every voxel type."""
    import ctypes as C
    import gzip
    from ukbb_cardiac_amd import _labelgz
    rng = np.random.default_rng(0)
    geo = np.concatenate([np.full(2 ** min(k, 18), k % 256, np.uint8) for k in range(24)])
    rng.shuffle(geo)
    cases = [np.zeros(1, np.uint8), np.zeros(100000, np.uint8), np.full(777, 3, np.uint8), np.arange(256, dtype=np.uint8),
             np.repeat(np.arange(256, dtype=np.uint8), 3), geo, rng.integers(0, 256, 50000).astype(np.uint8)]
    for lab in cases:
        for code, dt in ((2, 'u1'), (4, '<i2'), (8, '<i4'), (16, '<f4'), (64, '<f8')):
            prefix = bytes(rng.integers(0, 256, 352, dtype=np.uint8))
            cap = int(_labelgz.lib.ukbb_fcn_gzip_labels_bound(lab.size, code, len(prefix)))
            for mode in (_labelgz.FIXED, _labelgz.DYNAMIC):
                out = np.empty(cap, np.uint8)
                got = _labelgz.lib.ukbb_fcn_gzip_labels_mode(lab.ctypes.data, lab.size, code, prefix, len(prefix), out.ctypes.data, cap, mode)
                assert got > 0
                assert gzip.decompress(out[:got].tobytes()) == prefix + lab.astype(dt).tobytes()
    out = np.empty(1024, np.uint8)
    assert _labelgz.lib.ukbb_fcn_gzip_labels_mode(cases[0].ctypes.data, 1, 64, b'', 0, out.ctypes.data, 1024, 7) == -1   # unknown mode


def test_output_files_appear_only_when_complete(tmp_path, monkeypatch):
    """seg_{seq}.nii.gz is the 'already segmented' marker (deploy_network.py:66-67).  Every file is
    written under a temporary name and renamed; the marker is written last; a writer that dies leaves neither a truncated
    file nor a marker, so the rerun segments the subject again."""
    d, affine, pixdim = _write_subject(tmp_path, 's1', 'sa', (30, 44, 2, 5), 3)
    F, _ = DN.define_flags().parse(['--seq_name', 'sa', '--data_dir', str(tmp_path), '--model_path', 'x'])
    order = []
    real_save = nifti.save

    def spying_save(data, path, *a, **k):
        order.append(os.path.basename(path))
        return real_save(data, path, *a, **k)
    monkeypatch.setattr(nifti, 'save', spying_save)
    DN.run(F, stub_forward, log=lambda *_: None)
    assert order[-1] == 'seg_sa.nii.gz' and sorted(order[:-1]) == ['sa_ED.nii.gz', 'sa_ES.nii.gz', 'seg_sa_ED.nii.gz', 'seg_sa_ES.nii.gz']
    assert not [n for n in os.listdir(d) if '.tmp.' in n]
    # a crash inside the writer of the marker: no marker, no stray temporary file, and the rerun redoes the subject
    for n in os.listdir(d):
        if n != 'sa.nii.gz':
            os.remove(str(d / n))
    monkeypatch.setattr(nifti, 'save', real_save)
    real_write = nifti._AtomicFile.write

    def dying_write(self, b):
        if self.path.endswith('seg_sa.nii.gz'):
            real_write(self, bytes(b)[:10])
            raise OSError('disk full')
        return real_write(self, b)
    monkeypatch.setattr(nifti._AtomicFile, 'write', dying_write)
    with pytest.raises(OSError):
        DN.run(F, stub_forward, log=lambda *_: None)
    assert 'seg_sa.nii.gz' not in os.listdir(d) and not [n for n in os.listdir(d) if '.tmp.' in n]
    monkeypatch.setattr(nifti._AtomicFile, 'write', real_write)
    assert DN.run(F, stub_forward, log=lambda *_: None) == ['s1']
    assert 'seg_sa.nii.gz' in os.listdir(d)


def _eval_ventricular_row(image_name, seg_name):
    """short_axis/eval_ventricular_volume.py:40-73 restated on this repo's NIfTI reader (test-side checker)."""
    hdr = nifti.load_header(image_name)
    pixdim = hdr['pixdim'][1:4]
    volume_per_pix = pixdim[0] * pixdim[1] * pixdim[2] * 1e-3
    density = 1.05
    duration_per_cycle = hdr['dim'][4] * hdr['pixdim'][4]
    heart_rate = 60.0 / duration_per_cycle
    seg = nifti.load(seg_name).get_data()
    frame = {'ED': 0}
    vol_t = np.sum(seg == 1, axis=(0, 1, 2)) * volume_per_pix
    frame['ES'] = np.argmin(vol_t)
    val = {}
    for fr_name, fr in frame.items():
        val['LV{0}V'.format(fr_name)] = np.sum(seg[:, :, :, fr] == 1) * volume_per_pix
        val['LV{0}M'.format(fr_name)] = np.sum(seg[:, :, :, fr] == 2) * volume_per_pix * density
        val['RV{0}V'.format(fr_name)] = np.sum(seg[:, :, :, fr] == 3) * volume_per_pix
    val['LVSV'] = val['LVEDV'] - val['LVESV']
    val['LVCO'] = val['LVSV'] * heart_rate * 1e-3
    val['LVEF'] = val['LVSV'] / val['LVEDV'] * 100
    val['RVSV'] = val['RVEDV'] - val['RVESV']
    val['RVCO'] = val['RVSV'] * heart_rate * 1e-3
    val['RVEF'] = val['RVSV'] / val['RVEDV'] * 100
    return [val['LVEDV'], val['LVESV'], val['LVSV'], val['LVEF'], val['LVCO'], val['LVEDM'],
            val['RVEDV'], val['RVESV'], val['RVSV'], val['RVEF']]


def _pandas_csv(path, rows, index, columns):
    import pandas as pd
    pd.DataFrame(rows, index=index, columns=columns).to_csv(path)
    return open(path).read()


def test_output_csv_equals_the_evaluation_script_on_the_written_files(tmp_path):
    """VERDICT r02 item 5 (f4), host loop: deploy_network.py --output_csv writes what eval_ventricular_volume.py computes
    from sa.nii.gz + seg_sa.nii.gz, as pandas would format it; subjects segmented by an earlier run are measured from
    their files; two workers + merge give the same table."""
    from ukbb_cardiac_amd import measures
    data = tmp_path / 'data'
    data.mkdir()
    names = ['s%02d' % i for i in range(5)]
    for i, n in enumerate(names):
        _write_subject(data, n, 'sa', (30 + 2 * i, 44, 3, 6), 20 + i)
    (data / 'zz_empty').mkdir()

    def four_class(batch):                                        # stand-in network with all four short-axis labels
        return {'pred': np.digitize(batch, [0.25, 0.5, 0.75]).astype(np.int32)}
    csv1 = str(tmp_path / 'one.csv')
    F, _ = DN.define_flags().parse(['--seq_name', 'sa', '--data_dir', str(data), '--model_path', 'x', '--output_csv', csv1])
    assert DN.run(F, four_class, log=lambda *_: None) == names
    want = [_eval_ventricular_row(str(data / n / 'sa.nii.gz'), str(data / n / 'seg_sa.nii.gz')) for n in names]
    assert open(csv1).read() == _pandas_csv(str(tmp_path / 'pd.csv'), want, names, measures.SA_COLUMNS)
    # rerun: everything is skipped, the table is rebuilt from the files -- identical
    os.remove(csv1)
    assert DN.run(F, four_class, log=lambda *_: None) == []
    assert open(csv1).read() == open(str(tmp_path / 'pd.csv')).read()
    # two shards, then the launcher's merge
    csv2 = str(tmp_path / 'two.csv')
    for n in names:
        for f in os.listdir(data / n):
            if f != 'sa.nii.gz':
                os.remove(str(data / n / f))
    for idx in range(2):
        Fs, _ = DN.define_flags().parse(['--seq_name', 'sa', '--data_dir', str(data), '--model_path', 'x', '--output_csv', csv2,
                                        '--num_shards', '2', '--shard_index', str(idx)])
        DN.run(Fs, four_class, log=lambda *_: None)
    assert not os.path.exists(csv2)
    assert measures.merge_shard_csv(csv2, 2)
    assert open(csv2).read() == open(str(tmp_path / 'pd.csv')).read()
    assert not [f for f in os.listdir(tmp_path) if 'shard' in f]
    with pytest.raises(ValueError):
        Fb, _ = DN.define_flags().parse(['--seq_name', 'la_2ch', '--data_dir', str(data), '--output_csv', csv2])
        DN.run(Fb, four_class, log=lambda *_: None)


def test_aortic_output_csv_equals_the_evaluation_script_formulas(tmp_path):
    """deploy_network_ao.py --output_csv [--pressure_csv]: aortic/eval_aortic_area.py:60-95 on the written files (the script's
    quality control switched off here: test_aortic_output_csv_applies_the_count_only_quality_control covers it)."""
    from ukbb_cardiac_amd import measures
    names = ['1001', '1002', '1003']
    (tmp_path / 'd').mkdir()
    for i, n in enumerate(names):
        _write_subject(tmp_path / 'd', n, 'ao', (40, 36, 1, 6 + i), 5 + i)
    pcsv = tmp_path / 'p.csv'
    pcsv.write_text('eid,Central pulse pressure during PWA,Central pulse pressure during PWA,Other\n,12678-2.0,12678-2.1,x\n'
                    '1001,40,44,1\n1002,,38,2\n1003,5,7,3\n')
    out = str(tmp_path / 'ao.csv')
    F, _ = DA.define_flags().parse(['--data_dir', str(tmp_path / 'd'), '--model', 'UNet', '--model_path', 'x', '--io_threads', '0',
                                    '--output_csv', out, '--pressure_csv', str(pcsv), '--noaortic_qc'])
    DA.run(F, stub_forward, log=lambda *_: None)
    pp = {'1001': 42.0, '1002': 38.0, '1003': float('nan')}
    want = []
    for n in names:
        hdr = nifti.load_header(str(tmp_path / 'd' / n / 'ao.nii.gz'))
        dx, dy = hdr['pixdim'][1:3]
        area_per_pixel = dx * dy
        seg = nifti.load(str(tmp_path / 'd' / n / 'seg_ao.nii.gz')).get_data()
        line = []
        for l in (1, 2):
            A = np.sum(seg == l, axis=(0, 1, 2)) * area_per_pixel
            line += [A.max(), A.min(), (A.max() - A.min()) / (A.min() * pp[n]) * 1e3]
        want.append(line)
    assert open(out).read() == _pandas_csv(str(tmp_path / 'pd.csv'), want, names, measures.AO_COLUMNS)


def test_aorta_qc_from_counts_follows_the_script_criteria():
    """cardiac_utils.aorta_pass_quality_control criteria 1, 4, 5 (reference common/cardiac_utils.py:1741-1749,1782-1795) on per-frame
    areas: order per label (AAo first), the wrap-around of criterion 4 (frame 0 against the last frame), the script's messages."""
    from ukbb_cardiac_amd.measures import aorta_qc_from_counts
    T = 6
    good = np.stack([np.full(T, 500), np.array([100, 110, 120, 130, 120, 105]), np.array([80, 82, 85, 90, 86, 81])], axis=1)
    assert aorta_qc_from_counts(good) == (True, '')
    c = good.copy(); c[3, 2] = 0                                   # DAo vanishes in frame 3
    assert aorta_qc_from_counts(c) == (False, 'The area of DAo is 0 at time frame 3.')
    c = good.copy(); c[2, 1] = 0; c[3, 2] = 0                      # AAo is checked first
    assert aorta_qc_from_counts(c) == (False, 'The area of AAo is 0 at time frame 2.')
    c = good.copy(); c[4, 1] = 260                                 # 260 / 130 = 2: abrupt change at frame 4 (>= 2)
    assert aorta_qc_from_counts(c) == (False, 'There is abrupt change of area at time frame 4.')
    c = good.copy(); c[:, 1] = [100, 120, 144, 172, 190, 201]      # smooth steps, but frame 0 / LAST frame = 100 / 201 <= 0.5:
    assert aorta_qc_from_counts(c) == (False, 'There is abrupt change of area at time frame 0.')      # the script's A[t-1] wraps
    c = good.copy(); c[:, 2] = [100, 120, 144, 172, 190, 150]      # every step < 2x, wrap 100/150 fine, max / min = 1.9 -> passes
    assert aorta_qc_from_counts(c) == (True, '')
    c[:, 2] = [100, 130, 165, 200, 160, 125]                       # max / min = 2.0 with no abrupt step: criterion 5
    assert aorta_qc_from_counts(c) == (False, 'There is large change of area between maximum and minimum areas.')


def test_aortic_output_csv_applies_the_count_only_quality_control(tmp_path):
    """deploy_network_ao.py --output_csv drops the subjects eval_aortic_area.py:68-69 would drop on criteria 1 / 4 / 5, prints the
    script's message, keeps the others; a subject missing from the pressure spreadsheet is reported, not silently left empty."""
    from ukbb_cardiac_amd import measures
    names = ['2001', '2002', '2003']
    (tmp_path / 'd').mkdir()
    for i, n in enumerate(names):
        _write_subject(tmp_path / 'd', n, 'ao', (40, 36, 1, 6), 11 + i)
    # a forward whose labels depend on the subject: 2002 loses its DAo in frame 2, the others get steady discs
    def forward(batch):
        n = batch.shape[0]
        pred = np.zeros(batch.shape[:3], np.int32)
        cy, cx = batch.shape[1] // 2, batch.shape[2] // 2          # the 40 x 36 cine sits centred in the 256 x 256 pad (deploy_network_ao.py:105-107)
        pred[:, cy - 16:cy - 6, cx - 14:cx - 4] = 1
        pred[:, cy + 2:cy + 12, cx + 2:cx + 10] = 2
        prob = np.zeros(batch.shape[:3] + (3,), np.float32)
        np.put_along_axis(prob, pred[..., None], 1.0, axis=-1)
        return {'prob': prob, 'pred': pred}
    state = {'subject': None, 'frame': 0}
    lines = []
    def log(*a):                                               # the script prints the subject's name before it works on it (:137)
        line = ' '.join(str(x) for x in a)
        lines.append(line)
        if line in names:
            state['subject'], state['frame'] = line, 0
    def forward_2002_breaks(batch):
        out = forward(batch)
        for k in range(batch.shape[0]):
            if state['subject'] == '2002' and state['frame'] + k == 2:
                out['pred'][k][out['pred'][k] == 2] = 0
                out['prob'][k] = 0.0
                np.put_along_axis(out['prob'][k], out['pred'][k][..., None], 1.0, axis=-1)
        state['frame'] += batch.shape[0]
        return out
    pcsv = tmp_path / 'p.csv'
    pcsv.write_text('eid,Central pulse pressure during PWA,Central pulse pressure during PWA\n,12678-2.0,12678-2.1\n2001,40,44\n2002,30,30\n')
    out = str(tmp_path / 'ao.csv')
    F, _ = DA.define_flags().parse(['--data_dir', str(tmp_path / 'd'), '--model', 'UNet', '--model_path', 'x', '--io_threads', '0',
                                    '--output_csv', out, '--pressure_csv', str(pcsv)])
    DA.run(F, forward_2002_breaks, log=log)
    rows = open(out).read().splitlines()
    assert [r.split(',')[0] for r in rows[1:]] == ['2001', '2003']
    assert any('The area of DAo is 0 at time frame 2.' in l for l in lines)
    assert any('2003' in l and 'pressure spreadsheet' in l for l in lines)
    assert rows[1].split(',')[3] != '' and rows[2].split(',')[3] == ''       # distensibility: 2001 has a pulse pressure, 2003 has none
    assert measures.AO_COLUMNS[0] in rows[0]
    # --noaortic_qc: every segmented subject gets its row
    F2, _ = DA.define_flags().parse(['--data_dir', str(tmp_path / 'd'), '--model', 'UNet', '--model_path', 'x', '--io_threads', '0',
                                     '--output_csv', out, '--noaortic_qc'])
    DA.run(F2, forward_2002_breaks, log=log)
    assert [r.split(',')[0] for r in open(out).read().splitlines()[1:]] == names


def test_label_gzip_small_never_loses_to_zlib_on_noise(tmp_path):
    """Noise-like label volumes (runs of 1-2 voxels, not a segmentation) are where zlib's cross-row matches beat run-length
    tokens: 'small' then keeps the zlib level-1 stream, so its files are never larger than nibabel's, and inflate identically."""
    import gzip
    from ukbb_cardiac_amd import nifti
    lab = np.random.default_rng(5).integers(0, 4, size=(96, 104, 5, 8)).astype(np.uint8)
    sizes, raws = {}, {}
    try:
        for mode in ('small', 'zlib'):
            nifti.set_label_gzip(mode)
            p = str(tmp_path / (mode + '.nii.gz'))
            nifti.save(lab, p, np.eye(4), as_dtype=np.float64)
            sizes[mode], raws[mode] = os.path.getsize(p), gzip.open(p, 'rb').read()
    finally:
        nifti.set_label_gzip('small')
    assert raws['small'] == raws['zlib'] and sizes['small'] <= sizes['zlib'] + 16, sizes


def test_label_gzip_clean_under_address_and_ub_sanitizers(tmp_path):
    """csrc/label_gzip.cpp (bit writer, length-limited Huffman construction, run splitting, CRC shortcuts) built with
    -fsanitize=address,undefined and driven through its corner-case tests in a child interpreter (GPU sanitizers are not
    available on the pool; this is host code)."""
    import shutil
    import subprocess
    import sys
    if not shutil.which('g++'):
        pytest.skip('no g++')
    asan = subprocess.run(['gcc', '-print-file-name=libasan.so'], stdout=subprocess.PIPE, text=True).stdout.strip()
    ubsan = subprocess.run(['gcc', '-print-file-name=libubsan.so'], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not (os.path.isabs(asan) and os.path.exists(asan) and os.path.isabs(ubsan) and os.path.exists(ubsan)):
        pytest.skip('sanitizer runtimes not installed')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = str(tmp_path / 'libukbb_labelgz_asan.so')
    subprocess.check_call(['g++', '-O1', '-g', '-std=c++17', '-fPIC', '-shared', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined',
                           '-fno-omit-frame-pointer', '-o', lib, os.path.join(root, 'ukbb_cardiac_amd', 'csrc', 'label_gzip.cpp'),
                           os.path.join(root, 'ukbb_cardiac_amd', 'csrc', 'gz_inflate.cpp')])
    env = dict(os.environ, LD_PRELOAD=asan + ' ' + ubsan, ASAN_OPTIONS='detect_leaks=0', UKBB_LABELGZ_LIB=lib)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-p', 'no:cacheprovider', '-k',
                        'degenerate_histograms or run_lengths_and_crc or modes_inflate_identically or never_loses_to_zlib or test_gunzip_ or whole_file_decoder'],
                       env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and ' passed' in r.stdout and 'ERROR: AddressSanitizer' not in r.stdout and 'runtime error' not in r.stdout, r.stdout[-3000:]


def test_stale_tmp_files_of_dead_writers_are_swept_and_noise_fallback_keeps_the_header(tmp_path):
    """(a) a worker killed mid-write leaves ``<name>.tmp.<pid>.<tid>`` behind; the next writer of that target removes it
    (files of live processes stay); (b) when 'small' keeps the zlib stream on a noise-like volume the file still starts with the
    10-byte header of the other writers (no name, no time, OS = 255), so it is cmp-identical to a --label_gzip zlib run."""
    import gzip
    import subprocess
    import sys
    from ukbb_cardiac_amd import nifti
    target = tmp_path / 'seg_sa.nii.gz'
    p = subprocess.Popen([sys.executable, '-c', 'pass']); p.wait()        # a pid that no longer exists
    dead = tmp_path / ('seg_sa.nii.gz.tmp.%d.12345' % p.pid)
    dead.write_bytes(b'x' * 100)
    alive = tmp_path / ('seg_sa.nii.gz.tmp.%d.777' % os.getppid())       # the parent is alive: not ours to touch
    alive.write_bytes(b'y')
    # names carry a host tag; the pid probe only applies to this host's files, another host's file goes by age alone
    dead_here = tmp_path / ('seg_sa.nii.gz.tmp.%d.5.%s' % (p.pid, nifti._host_tag()))
    dead_here.write_bytes(b'z')
    foreign_young = tmp_path / ('seg_sa.nii.gz.tmp.%d.5.h0123456789' % p.pid)          # "dead" pid here, but a live writer elsewhere
    foreign_young.write_bytes(b'w')
    foreign_old = tmp_path / ('seg_sa.nii.gz.tmp.%d.6.h0123456789' % os.getppid())     # "live" pid here, abandoned a day ago elsewhere
    foreign_old.write_bytes(b'v')
    os.utime(str(foreign_old), (1.0e9, 1.0e9))
    assert nifti._tmp_name(str(target)).endswith('.' + nifti._host_tag())
    lab = np.random.default_rng(7).integers(0, 4, size=(48, 40, 3, 4)).astype(np.uint8)
    raws = {}
    try:
        for mode in ('small', 'zlib'):
            nifti.set_label_gzip(mode)
            nifti.save(lab, str(target), np.eye(4), as_dtype=np.float64)
            raws[mode] = target.read_bytes()
            assert not dead.exists() and alive.exists()
            assert not dead_here.exists() and foreign_young.exists() and not foreign_old.exists()
    finally:
        nifti.set_label_gzip('small')
    assert raws['small'][:10] == raws['zlib'][:10] == bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4, 0xff])
    assert gzip.decompress(raws['small']) == gzip.decompress(raws['zlib'])
    assert raws['small'] == raws['zlib']                                  # noise: 'small' kept zlib's stream, byte for byte the zlib-mode file


def test_pmc_summary_keeps_kernels_of_anonymous_namespaces():
    """tools/pmc_summary.py shortens rocprofv3 kernel names; the '(anonymous namespace)' of kernels_ws / tail / stem has parentheses of
    its own and the r04 mid-round summaries lost those kernels to a greedy regular expression."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('pmc_summary', os.path.join(root, 'tools', 'pmc_summary.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.short('void ukbb::(anonymous namespace)::unet_tail_kernel<8, 8, 1, 3, false>(ukbb::TailArgs)') == 'unet_tail_kernel<8, 8, 1, 3, false>'
    assert m.short('void ukbb::wino_pc_kernel<4, 4>(ukbb::ConvArgs)') == 'wino_pc_kernel<4, 4>'
    assert m.short('clock_probe_kernel(unsigned long long*, unsigned long long)') == 'clock_probe_kernel'


def test_threshold_params_give_the_label_map_they_promise():
    """weights.threshold_params (the 'realistic label statistics' model of tools/shard_rehearsal.py): through the fp64 oracle the label is
    the number of thresholds below the twice 3x3-averaged image, for every FCN model."""
    from oracle import fcn_oracle as O
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import pack_flat, threshold_params
    img = cine_phantom(2, 48, 64, seed=5).astype(np.float64)

    def mean3(a):
        pad = np.pad(a, ((0, 0), (1, 1), (1, 1)))
        return sum(pad[:, i:i + a.shape[1], j:j + a.shape[2]] for i in range(3) for j in range(3)) / 9.0
    m = mean3(mean3(img[..., 0]))
    for name in ('FCN_sa', 'FCN_la_2ch', 'FCN_la_4ch_seg4'):
        arch = MODELS[name]
        P = threshold_params(arch)
        assert pack_flat(arch, P).size == arch.n_weight_floats()
        lg = O.build_FCN(img, P, arch.n_class, arch.n_level, arch.n_filter, arch.n_block, arch.same_dim, arch.fc)
        th = [0.3 + 0.5 * c / max(1, arch.n_class - 2) for c in range(arch.n_class - 1)]
        want = sum((m > t).astype(int) for t in th)
        lab = np.argmax(lg, -1)
        clear = np.min([np.abs(m - t) for t in th], axis=0) > 1e-9            # not exactly on a threshold
        assert np.array_equal(lab[clear], want[clear]) and len(np.unique(lab)) == arch.n_class
    with pytest.raises(ValueError):
        threshold_params(MODELS['UNet_ao'])
