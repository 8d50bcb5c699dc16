"""Parity of the HIP path (through the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): argmax label maps bit-exact, pre-softmax logits
within 1e-3 relative fp32.  "Relative" is taken against max|logits| of the
batch (the scale the argmax decision lives on).  Label maps must be identical
to the fp64 oracle's at every pixel whose decision is numerically determined:
a disagreement is tolerated only where the oracle's own top-2 logit margin is
below NEAR_TIE = 1e-4 (fp32 evaluation noise of the logits is ~2e-5, so there
any fp32 implementation -- including TensorFlow's -- is a coin flip), and the
number of such pixels is bounded (observed: 0-1 per 40 k pixels).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
LOGIT_RTOL = 1e-3          # north-star tolerance
NEAR_TIE = 1e-4            # margin (in logit units, |logits| = O(5)) below which a flip is a tie


@pytest.fixture(scope='module')
def engines():
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.weights import synthetic_params
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Engine(MODELS[name], synthetic_params(MODELS[name], 1234))
        return cache[name]
    yield get
    for e in cache.values():
        e.close()


FCN_GOLDENS = ['fcn_sa_2x32x48', 'fcn_sa_1x192x208', 'fcn_sa_1x192x208_uniform', 'fcn_la2ch_1x176x208',
               'fcn_seg4_1x80x112', 'fcn_la4ch_2x48x16']


@pytest.mark.parametrize('tag', FCN_GOLDENS)
def test_fcn_golden(engines, tag, parity_log):
    g = np.load(os.path.join(GOLD, tag + '.npz'))
    eng = engines(str(g['model']))
    out = eng.run(g['image'], want_logits=True, want_prob=True, want_pred=True)
    ref = g['logits64']
    scale = np.abs(ref).max()
    err = np.abs(out['logits'] - ref).max()
    assert err <= LOGIT_RTOL * scale, 'logits err %.3e vs scale %.3e' % (err, scale)
    assert out['pred'].dtype == np.int32
    bad = out['pred'] != g['pred64']
    parity_log(model=str(g['model']), shape=list(g['image'].shape[:3]), oracle='numpy fp64 (committed golden)',
               max_logits_err=float(err), logits_scale=float(scale), rel_err=float(err / scale),
               pixels=int(bad.size), label_flips=int(bad.sum()),
               flips_away_from_tie=int((bad & (g['margin64'] > NEAR_TIE)).sum()),
               near_tie_pixels=int((g['margin64'] <= NEAR_TIE).sum()))
    assert not np.any(bad & (g['margin64'] > NEAR_TIE)), 'label flip away from a numerical tie'
    assert bad.sum() <= max(1, int((g['margin64'] <= NEAR_TIE).sum())), '%d label mismatches' % int(bad.sum())
    # prob = softmax(logits); pred = argmax(prob)
    e = np.exp(ref - ref.max(-1, keepdims=True)); p = e / e.sum(-1, keepdims=True)
    assert np.abs(out['prob'] - p).max() <= 1e-4
    assert np.array_equal(np.argmax(out['prob'], -1).astype(np.int32), out['pred'])


def test_intermediate_activations_match_oracle(engines):
    from oracle import fcn_oracle as O
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['FCN_sa']
    img = cine_phantom(2, 48, 64, seed=21)
    eng = engines('FCN_sa')
    eng.run(img)
    _, net = O.build_FCN(img, synthetic_params(arch, 1234), 4, dtype=np.float64, return_net=True)
    params = synthetic_params(arch, 1234)
    for l in range(5):
        a = eng.activation('conv%d' % l).reshape(net['conv%d' % l].shape)
        assert np.abs(a - net['conv%d' % l]).max() <= 1e-4 * max(1.0, np.abs(net['conv%d' % l]).max()), l
    # g_l = (out0 weights of level l, BN scale folded) applied to the squeezed map at LOW resolution
    p0 = params['out0']
    scale = p0['gamma'].astype(np.float64) / np.sqrt(p0['var'].astype(np.float64) + 1e-3)
    for l in range(1, 5):
        ref = net['conv%d_same_dim' % l] @ (p0['kernel'][0, 0, 32 * l:32 * (l + 1), :].astype(np.float64) * scale)
        a = eng.activation('g%d' % l).reshape(ref.shape)
        assert np.abs(a - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), l


@pytest.mark.parametrize('n', [1, 3, 10])
def test_batch_independence_and_determinism(engines, n):
    """Each slice's result must not depend on its batch mates (the reference
    batches the Z slices of a frame, deploy_network.py:105-111) and repeated
    runs are bit-identical."""
    from ukbb_cardiac_amd.phantom import cine_phantom
    eng = engines('FCN_sa')
    img = cine_phantom(n, 64, 80, seed=31)
    a = eng.run(img, want_logits=True)
    b = eng.run(img, want_logits=True)
    assert np.array_equal(a['logits'], b['logits']) and np.array_equal(a['pred'], b['pred'])
    for i in range(n):
        s = eng.run(img[i:i + 1], want_logits=True)
        assert np.array_equal(s['logits'][0], a['logits'][i])


def _grade_vs_c_oracle(eng, arch, params, img, parity_log, chunk=8, max_flips_per_million=40):
    """Engine vs oracle/fcn_oracle.c (fp32, unfused op-by-op restatement) on the same batch: logits within
    LOGIT_RTOL * max|logits|; label disagreements are then classified with the numpy fp64 oracle run on just the
    slices that contain one -- each must sit where the fp64 top-2 margin is below NEAR_TIE (two correct fp32
    evaluations may differ there), and their rate is bounded."""
    from oracle import c_oracle, fcn_oracle as O
    from ukbb_cardiac_amd.weights import pack_flat
    flat = pack_flat(arch, params)
    out = eng.run(img, want_logits=True, want_prob=False)
    n = img.shape[0]
    ref_l = np.empty_like(out['logits'])
    ref_p = np.empty_like(out['pred'])
    for i in range(0, n, chunk):                              # the C oracle materialises the 160-channel concat
        lg, _, pd = c_oracle.forward(arch, flat, img[i:i + chunk])
        ref_l[i:i + chunk], ref_p[i:i + chunk] = lg, pd
    scale = float(np.abs(ref_l).max())
    err = float(np.abs(out['logits'] - ref_l).max())
    bad = out['pred'] != ref_p
    slices = np.nonzero(bad.reshape(n, -1).any(axis=1))[0]
    away, hip_wrong, c_wrong = 0, 0, 0
    for i in slices:                                          # fp64 arbitration, only where the two fp32 results differ
        ref64 = O.build_FCN(img[i:i + 1], params, arch.n_class, dtype=np.float64) if arch.kind == 0 else \
            O.UNet(img[i:i + 1], params, arch.n_class, n_block=arch.n_block, dtype=np.float64)
        margin = O.top2_margin(ref64)[0]
        p64 = O.argmax_pred(ref64)[0]
        away += int((bad[i] & (margin > NEAR_TIE)).sum())
        hip_wrong += int((bad[i] & (out['pred'][i] != p64)).sum())
        c_wrong += int((bad[i] & (ref_p[i] != p64)).sum())
    parity_log(model=arch.name, shape=list(img.shape[:3]), oracle='oracle/fcn_oracle.c fp32 (+ numpy fp64 on disagreeing slices)',
               max_logits_err=err, logits_scale=scale, rel_err=err / scale, pixels=int(bad.size),
               label_flips=int(bad.sum()), flips_away_from_tie=away,
               flips_where_hip_differs_from_fp64=hip_wrong, flips_where_c_oracle_differs_from_fp64=c_wrong)
    assert err <= LOGIT_RTOL * scale, 'logits err %.3e vs scale %.3e' % (err, scale)
    assert away == 0, '%d label disagreements away from a numerical tie' % away
    assert bad.sum() <= max(2, max_flips_per_million * bad.size // 1000000), '%d near-tie flips in %d pixels' % (bad.sum(), bad.size)
    return out


def test_full_batch64_vs_c_oracle(engines, parity_log):
    """BASELINE config 2, the exact bench workload: N = 64 x 192 x 208 uniform-random slices,
    every logit and every label against the C oracle."""
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import uniform_slices
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['FCN_sa']
    _grade_vs_c_oracle(engines('FCN_sa'), arch, synthetic_params(arch, 1234), uniform_slices(64, 192, 208, seed=1), parity_log)


def test_full_batch64_phantom_vs_c_oracle(engines, parity_log):
    """Same size on structured images (blood-pool phantom): all four classes populated, large flat regions."""
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['FCN_sa']
    out = _grade_vs_c_oracle(engines('FCN_sa'), arch, synthetic_params(arch, 1234), cine_phantom(64, 192, 208, seed=41), parity_log)
    assert len(np.unique(out['pred'])) == 4


@pytest.mark.parametrize('model,shape', [('FCN_sa', (10, 192, 208)), ('FCN_la_2ch', (50, 176, 208)), ('FCN_la_4ch', (50, 176, 208)),
                                         ('FCN_la_4ch_seg4', (50, 176, 208)), ('FCN_sa', (4, 208, 256)), ('UNet_ao', (6, 256, 256))])
def test_config3_shapes_vs_c_oracle(engines, parity_log, model, shape):
    """BASELINE config 3: short-axis N = 10 (the reference's own per-frame call), the long-axis
    models at 162x204 -> 176x208 with the T frames as the batch, one 208x256 shape; plus the aortic U-Net at its
    fixed 256x256."""
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS[model]
    img = cine_phantom(*shape, seed=sum(shape))
    if arch.kind == 1:
        img = ((img - 0.3) / 0.25).astype(np.float32)                   # z-score-like range (deploy_network_ao.py:93-94)
    _grade_vs_c_oracle(engines(model), arch, synthetic_params(arch, 1234), img, parity_log)


@pytest.mark.parametrize('shape', [(1, 1024, 1024), (3, 272, 304), (2, 16, 16), (5, 48, 400)])
def test_unusual_sizes_vs_c_oracle(engines, parity_log, shape):
    """Sizes the reference never pads to but accepts (any multiple of 16, deploy_network.py:97): a 1024 x 1024 slice (2048
    stride-2 tiles per image: the straight-line producer with 8x16 tiles), shapes whose pyramids no tiling divides, the
    smallest legal image (one pixel at level 4) and a very elongated one."""
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['FCN_sa']
    _grade_vs_c_oracle(engines('FCN_sa'), arch, synthetic_params(arch, 1234), cine_phantom(*shape, seed=shape[1] + shape[2]), parity_log, chunk=2)


def test_full_batch64_slices_equal_single_slice_runs(engines):
    """At the bench batch every workgroup of the stride-2 layers has enough stages for the straight-line producer
    (loads two stages ahead, last halo column / rows below the image through buffer-descriptor ranges); a slice run
    alone takes the generic producer (per-lane range tests).  Same tiling, same consumer arithmetic: the logits of a
    slice must be bit-identical either way -- first, last and a middle slice of the batch, borders included."""
    from ukbb_cardiac_amd.phantom import cine_phantom
    eng = engines('FCN_sa')
    img = cine_phantom(64, 192, 208, seed=77)
    full = eng.run(img, want_logits=True, want_prob=False)
    for i in (0, 31, 63):
        one = eng.run(img[i:i + 1], want_logits=True, want_prob=False)
        assert np.array_equal(one['logits'][0], full['logits'][i]), i
        assert np.array_equal(one['pred'][0], full['pred'][i]), i


def test_full_batch64_properties(engines):
    """N=64 x 192 x 208 (the bench workload): shift-of-batch invariance and
    label histogram consistency between device pred and device logits."""
    from ukbb_cardiac_amd.phantom import uniform_slices
    eng = engines('FCN_sa')
    img = uniform_slices(64, 192, 208, seed=1)
    a = eng.run(img, want_logits=True, want_prob=False)
    assert np.array_equal(np.argmax(a['logits'], -1).astype(np.int32), a['pred'])
    rolled = np.roll(img, 7, axis=0)
    b = eng.run(rolled, want_logits=False, want_prob=False)
    assert np.array_equal(np.roll(a['pred'], 7, axis=0), b['pred'])
    assert np.isfinite(a['logits']).all()


def test_error_paths(engines):
    from ukbb_cardiac_amd import _lib
    eng = engines('FCN_sa')
    with pytest.raises(_lib.UkbbFcnError):
        eng.run(np.zeros((1, 30, 32, 1), np.float32))       # not a multiple of 16
    with pytest.raises(ValueError):
        eng.run(np.zeros((1, 32, 32, 2), np.float32))


def test_session_mirror(engines):
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Session
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['FCN_sa']
    img = cine_phantom(2, 32, 48, seed=11)
    with Session(arch=arch, params=synthetic_params(arch, 1234)) as sess:
        prob, pred = sess.run(['prob:0', 'pred:0'], feed_dict={'image:0': img, 'training:0': False})
        assert prob.shape == (2, 32, 48, 4) and pred.shape == (2, 32, 48) and pred.dtype == np.int32
        g = np.load(os.path.join(GOLD, 'fcn_sa_2x32x48.npz'))
        assert not np.any((pred != g['pred64']) & (g['margin64'] > NEAR_TIE))
        only = sess.run('prob:0', feed_dict={'image:0': img, 'training:0': False})
        assert np.array_equal(only, prob)
        with pytest.raises(KeyError):
            sess.run(['nope:0'], feed_dict={'image:0': img})
        with pytest.raises(ValueError):
            sess.run(['pred:0'], feed_dict={'image:0': img, 'training:0': True})


# ---- aortic U-Net (network_ao.py:18-64, BASELINE config 5 fp32 leg) -------------------
def test_unet_golden(engines):
    g = np.load(os.path.join(GOLD, 'unet_ao_2x64x96.npz'))
    out = engines('UNet_ao').run(g['image'], want_logits=True)
    ref = g['logits64']
    assert np.abs(out['logits'] - ref).max() <= LOGIT_RTOL * np.abs(ref).max()
    bad = out['pred'] != g['pred64']
    assert not np.any(bad & (g['margin64'] > NEAR_TIE)) and bad.sum() <= 1
    assert np.abs(out['prob'] - g['prob64']).max() <= 1e-4


def test_unet_256_decoder_activations(engines):
    """Fixed 256x256 aortic input size (deploy_network_ao.py:105); every decoder level
    (transposed conv + skip concat + convs) against the fp64 oracle."""
    from oracle import fcn_oracle as O
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['UNet_ao']
    img = ((cine_phantom(1, 256, 256, seed=51) - 0.3) / 0.25).astype(np.float32)
    eng = engines('UNet_ao')
    out = eng.run(img, want_logits=True)
    ref, net = O.UNet(img, synthetic_params(arch, 1234), 3, n_block=arch.n_block, dtype=np.float64, return_net=True)
    for l in (3, 2, 1, 0):
        a = eng.activation('up%d' % l).reshape(net['conv%d_up' % l].shape)
        assert np.abs(a - net['conv%d_up' % l]).max() <= 1e-4 * max(1.0, np.abs(net['conv%d_up' % l]).max()), l
    assert np.abs(out['logits'] - ref).max() <= LOGIT_RTOL * np.abs(ref).max()
    bad = out['pred'] != O.argmax_pred(ref)
    assert not np.any(bad & (O.top2_margin(ref) > NEAR_TIE)) and bad.sum() <= 2


def test_aortic_pipeline_on_device(engines):
    """deploy_network_ao.py 'UNet' branch end to end on a (X,Y,1,T) cine, device forward
    vs the same host loop driven by the C oracle."""
    from oracle import c_oracle
    from ukbb_cardiac_amd import pipeline
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.weights import pack_flat, synthetic_params
    arch = MODELS['UNet_ao']
    flat = pack_flat(arch, synthetic_params(arch, 1234))
    rng = np.random.default_rng(3)
    vol = (1000.0 * rng.gamma(2.0, 1.0, size=(200, 180, 1, 4))).astype(np.float32)
    eng = engines('UNet_ao')
    dev = pipeline.aortic_prob_sequence(vol.copy(), lambda b: eng.run(b), batch_slices=3)

    def cpu_forward(b):
        _, pr, pd = c_oracle.forward(arch, flat, b, want_logits=False, want_prob=True)
        return {'prob': pr, 'pred': pd}
    ref = pipeline.aortic_prob_sequence(vol.copy(), cpu_forward, batch_slices=4)
    assert dev.shape == (200, 180, 1, 4, 3) and np.abs(dev - ref).max() <= 1e-4
    assert (np.argmax(dev, -1) != np.argmax(ref, -1)).mean() < 1e-4


@pytest.mark.parametrize('shape', [(200, 180, 1, 7), (96, 120, 2, 5)])
def test_aortic_unet_device_pipeline_equals_host_pipeline(engines, shape):
    """Frame-wise 'UNet' sequences with z-score / pack / argmax on the GPU (device_pipeline.aortic_unet_sequence_device) against
    the numpy mirror of deploy_network_ao.py:92-128,189 driven by the same engine: identical labels, identical class counts."""
    from ukbb_cardiac_amd import pipeline
    from ukbb_cardiac_amd.device_pipeline import aortic_unet_sequence_device
    eng = engines('UNet_ao')
    rng = np.random.default_rng(shape[0])
    vol = np.asfortranarray(np.round(1000.0 * rng.gamma(2.0, 1.0, size=shape)).astype(np.float32))
    keep = vol.copy()
    pred, aux = aortic_unet_sequence_device(vol, eng, batch_slices=4, return_aux=True)
    assert np.array_equal(vol, keep)
    prob = pipeline.aortic_prob_sequence(vol, lambda b: eng.run(b), batch_slices=3)
    want = np.argmax(prob, axis=-1).astype(np.int32)
    assert pred.dtype == np.int32 and np.array_equal(pred, want)
    for c in range(3):
        assert np.array_equal(aux['counts'][:, c], (want == c).sum(axis=(0, 1, 2)))
    assert len(np.unique(pred)) > 1


def test_sa_pipeline_on_device(engines):
    """deploy_network.py sequence mode on an un-padded (X,Y,Z,T) volume: device labels
    == C-oracle labels through the same host loop (pads 7/7 and 2/2: SURVEY 8(c))."""
    from oracle import c_oracle
    from ukbb_cardiac_amd import pipeline
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.weights import pack_flat, synthetic_params
    arch = MODELS['FCN_sa']
    flat = pack_flat(arch, synthetic_params(arch, 1234))
    rng = np.random.default_rng(4)
    vol = (1000.0 * rng.gamma(2.0, 1.0, size=(162, 204, 2, 3))).astype(np.float32)
    eng = engines('FCN_sa')
    dev = pipeline.segment_sequence(vol.copy(), lambda b: eng.run(b, want_prob=False), batch_slices=4)
    ref = pipeline.segment_sequence(vol.copy(), lambda b: {'pred': c_oracle.forward(arch, flat, b, want_logits=False)[2]},
                                    batch_slices=6)
    assert dev.dtype == np.float64 and dev.shape == vol.shape
    assert (dev != ref).sum() <= 2          # only fp32 near-ties may differ between two fp32 evaluations


# ---- BASELINE config 5: aortic U-Net, bf16 MFMA path vs fp32 -------------------------------
def test_unet_bf16_dice_vs_fp32():
    """bf16 operands / fp32 accumulation on the conv stack: per-class Dice
    (common/image_utils.py:171-175) against the fp32 path and the fp64 oracle, and logits error."""
    from oracle import fcn_oracle as O
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.image_utils import np_categorical_dice
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['UNet_ao']
    params = synthetic_params(arch, 1234)
    img = ((cine_phantom(3, 256, 256, seed=61) - 0.3) / 0.25).astype(np.float32)
    with Engine(arch, params) as eng:
        f32 = eng.run(img, want_logits=True)
        eng.set_precision('bf16')
        b16 = eng.run(img, want_logits=True)
        names, cfgs = eng.kernel_names(), eng.kernel_configs()
        # every conv / transposed conv on the bf16-storage tilings (ConvConfig::pc == 5, ids 230-299 and 320-325; pc == 6, the
        # weight-stationary ids 400-), the stem and the tail as the fused launches of kernels_stem.hip / kernels_tail.hip (no tiling
        # id): nothing left in fp32.  20 launches: conv0_0 + conv0_1, 17 convs / transposed convs, up0_0 + up0_1 + logits
        assert names[0] == 'conv0_0+conv0_1' and names[-1] == 'up0_0+up0_1+logits' and cfgs[0] == -1 and cfgs[-1] == -1
        assert len(cfgs) == 20 and all((230 <= c < 330 and not 300 <= c < 310) or 400 <= c < 430 for c in cfgs[1:-1]), 'bf16 tilings not selected: %s' % list(zip(names, cfgs))
        assert sum(400 <= c < 430 for c in cfgs) >= 10, 'weight-stationary tilings not in the plan: %s' % list(zip(names, cfgs))
        prob16 = eng.run(img, want_logits=True, want_prob=True)
        assert np.array_equal(prob16['logits'], b16['logits']) and np.array_equal(np.argmax(prob16['prob'], -1), b16['pred'])
        eng.set_precision('fp32')
        again = eng.run(img, want_logits=True)
    assert np.array_equal(again['logits'], f32['logits'])                 # switching back is exact
    scale = np.abs(f32['logits']).max()
    rel = np.abs(b16['logits'] - f32['logits']).max() / scale
    assert 1e-5 < rel < 5e-2, rel                                         # really bf16, and sane
    ref = O.argmax_pred(O.UNet(img[:1], params, 3, n_block=arch.n_block, dtype=np.float64))
    for k in (1, 2):                                                      # classes 1, 2 as in SURVEY 8(d) config 5
        assert np_categorical_dice(b16['pred'], f32['pred'], k) >= 0.98
        assert np_categorical_dice(b16['pred'][:1], ref, k) >= 0.98
    assert (b16['pred'] != f32['pred']).mean() < 0.02


def test_unet_bf16_dice_at_the_bench_batch(parity_log):
    """BASELINE config 5 at its stated size: N = 100 frames of 256 x 256 (deploy_network_ao.py:105-107 pads every aortic frame to
    that), bf16 path (bf16 operands and bf16 activations in HBM, first layer and logits fused) against the fp32 path of the same
    engine: per-class Dice (common/image_utils.py:171-175) >= 0.98 on classes 1, 2 over the batch and on every single frame
    whose class is populated; the fp32 path itself is pinned to the oracles by the other tests of this file."""
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.image_utils import np_categorical_dice
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['UNet_ao']
    img = ((cine_phantom(100, 256, 256, seed=5) - 0.3) / 0.25).astype(np.float32)
    with Engine(arch, synthetic_params(arch, 1234)) as eng:
        f32 = eng.run(img, want_prob=False)['pred']
        eng.set_precision('bf16')
        b16 = eng.run(img, want_prob=False)['pred']
        again = eng.run(img, want_prob=False)['pred']
    assert np.array_equal(b16, again)                                     # bit-deterministic
    dice = {k: float(np_categorical_dice(b16, f32, k)) for k in (1, 2)}
    worst = {k: min(float(np_categorical_dice(b16[i], f32[i], k)) for i in range(100) if (f32[i] == k).sum() > 500) for k in (1, 2)}
    parity_log(model='UNet_ao', shape=[100, 256, 256], oracle='fp32 path of the same engine', dice_class1=dice[1], dice_class2=dice[2],
               worst_frame_dice_class1=worst[1], worst_frame_dice_class2=worst[2], label_disagreement=float((b16 != f32).mean()))
    assert dice[1] >= 0.98 and dice[2] >= 0.98, dice
    assert worst[1] >= 0.97 and worst[2] >= 0.97, worst
    assert len(np.unique(b16)) == 3


# ---- UKBB_PREC_F32X3: fp32 results from three bf16 pieces per operand (FCN head), graded exactly like the fp32 path -----

@pytest.fixture(scope='module')
def engines_x3():
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.weights import synthetic_params
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Engine(MODELS[name], synthetic_params(MODELS[name], 1234))
            cache[name].set_precision('f32x3')
        return cache[name]
    yield get
    for e in cache.values():
        e.close()


@pytest.mark.parametrize('tag', FCN_GOLDENS)
def test_f32x3_fcn_golden(engines_x3, engines, tag, parity_log):
    """Same goldens, same tolerances as test_fcn_golden; and the mode really takes another path (logits differ in the last bits)."""
    g = np.load(os.path.join(GOLD, tag + '.npz'))
    out = engines_x3(str(g['model'])).run(g['image'], want_logits=True, want_prob=True, want_pred=True)
    ref = g['logits64']
    scale = np.abs(ref).max()
    err = np.abs(out['logits'] - ref).max()
    assert err <= LOGIT_RTOL * scale
    bad = out['pred'] != g['pred64']
    parity_log(model=str(g['model']) + ' f32x3', shape=list(g['image'].shape[:3]), oracle='numpy fp64 (committed golden)',
               max_logits_err=float(err), logits_scale=float(scale), rel_err=float(err / scale), pixels=int(bad.size),
               label_flips=int(bad.sum()), flips_away_from_tie=int((bad & (g['margin64'] > NEAR_TIE)).sum()),
               near_tie_pixels=int((g['margin64'] <= NEAR_TIE).sum()))
    assert not np.any(bad & (g['margin64'] > NEAR_TIE))
    assert bad.sum() <= max(1, int((g['margin64'] <= NEAR_TIE).sum()))
    assert np.array_equal(np.argmax(out['prob'], -1).astype(np.int32), out['pred'])
    if g['image'].shape[1] >= 32:                              # big enough for the producer/consumer head kernel
        plain = engines(str(g['model'])).run(g['image'], want_logits=True)['logits']
        assert np.abs(plain - out['logits']).max() <= 1e-5 * scale


def test_f32x3_full_batch64_vs_c_oracle(engines_x3, parity_log):
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import uniform_slices
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['FCN_sa']

    def log(**kw):
        kw['model'] = kw.get('model', '') + ' f32x3'
        parity_log(**kw)
    _grade_vs_c_oracle(engines_x3('FCN_sa'), arch, synthetic_params(arch, 1234), uniform_slices(64, 192, 208, seed=1), log)


@pytest.mark.parametrize('model,shape', [('FCN_la_2ch', (50, 176, 208)), ('FCN_la_4ch_seg4', (50, 176, 208)), ('FCN_sa', (4, 208, 256))])
def test_f32x3_config3_shapes_vs_c_oracle(engines_x3, parity_log, model, shape):
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS[model]

    def log(**kw):
        kw['model'] = kw.get('model', '') + ' f32x3'
        parity_log(**kw)
    _grade_vs_c_oracle(engines_x3(model), arch, synthetic_params(arch, 1234), cine_phantom(*shape, seed=17).astype(np.float32), log)


def test_f32x3_is_deterministic_and_switchable(engines_x3):
    from ukbb_cardiac_amd.phantom import uniform_slices
    eng = engines_x3('FCN_sa')
    img = uniform_slices(3, 96, 112, seed=4)
    a = eng.run(img, want_logits=True)
    b = eng.run(img, want_logits=True)
    assert np.array_equal(a['logits'], b['logits']) and np.array_equal(a['pred'], b['pred'])
    eng.set_precision('fp32')
    c = eng.run(img, want_logits=True)
    eng.set_precision('f32x3')
    d = eng.run(img, want_logits=True)
    assert np.array_equal(a['logits'], d['logits'])
    assert not np.array_equal(a['logits'], c['logits'])        # another instruction sequence ...
    assert np.abs(a['logits'] - c['logits']).max() <= 1e-5 * np.abs(c['logits']).max()   # ... the same numbers to fp32 accuracy


# ---- r03 -------------------------------------------------------------------------------------------------------------------

def test_two_handles_interleaved_on_two_streams():
    """launch state is per device, never per process.  Two engines of different models created with an
    explicit device=0 (the nearest a one-GPU box gets to two devices: two handles, two workspaces), forwards interleaved on
    two HIP streams with every large-LDS kernel family in play (head, Winograd, producer/consumer convs, U-Net): results
    equal those of each engine running alone."""
    import torch
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    dev = torch.device('cuda', 0)
    xs = {'FCN_sa': torch.from_numpy(cine_phantom(6, 96, 112, seed=5)).to(dev),
          'UNet_ao': torch.from_numpy(cine_phantom(3, 64, 96, seed=6)).to(dev)}
    alone = {}
    for name, x in xs.items():
        with Engine(MODELS[name], synthetic_params(MODELS[name], 1234), device=0) as e:
            n, h, w = x.shape[:3]
            lg = torch.empty((n, h, w, MODELS[name].n_class), dtype=torch.float32, device=dev)
            pd = torch.empty((n, h, w), dtype=torch.int32, device=dev)
            e.run_device(x.data_ptr(), n, h, w, logits_ptr=lg.data_ptr(), pred_ptr=pd.data_ptr())
            torch.cuda.synchronize()
            alone[name] = (lg.cpu().numpy(), pd.cpu().numpy())
    engs = {name: Engine(MODELS[name], synthetic_params(MODELS[name], 1234), device=0) for name in xs}
    streams = {name: torch.cuda.Stream(dev) for name in xs}
    try:
        outs = {name: [] for name in xs}
        for rep in range(4):
            for name, x in xs.items():
                n, h, w = x.shape[:3]
                lg = torch.empty((n, h, w, MODELS[name].n_class), dtype=torch.float32, device=dev)
                pd = torch.empty((n, h, w), dtype=torch.int32, device=dev)
                engs[name].run_device(x.data_ptr(), n, h, w, logits_ptr=lg.data_ptr(), pred_ptr=pd.data_ptr(),
                                      stream=streams[name].cuda_stream)
                outs[name].append((lg, pd))
        torch.cuda.synchronize()
        for name in xs:
            for lg, pd in outs[name]:
                assert np.array_equal(lg.cpu().numpy(), alone[name][0]) and np.array_equal(pd.cpu().numpy(), alone[name][1]), name
    finally:
        for e in engs.values():
            e.close()


@pytest.mark.parametrize('name', ['FCN_sa', 'FCN_la_4ch_seg4', 'UNet_ao'])
def test_pred_is_the_lowest_index_argmax_of_the_float32_probabilities(name):
    """train_network.py:198-199 / network_ao.py:159-160: pred = argmax(softmax(logits)) over the float32 PROBABILITIES.  With
    the last layer scaled to 1e-9 every logit difference is far below one ulp of exp's argument at 1.0: all probabilities
    round to the same float and the reference's label is class 0 everywhere, while the logits still differ (their argmax
    would scatter over the classes).  Also at 1e-7 -- a mix of ties and non-ties -- pred == np.argmax(prob) exactly."""
    import copy
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS[name]
    img = cine_phantom(2, 64, 80, seed=9)
    for scale, expect_all_zero in ((1e-9, True), (1e-7, False)):
        params = copy.deepcopy(synthetic_params(arch, 1234))
        params['logits']['kernel'] = (params['logits']['kernel'] * scale).astype(np.float32)
        params['logits']['bias'] = (params['logits']['bias'] * scale).astype(np.float32)
        with Engine(arch, params) as e:
            out = e.run(img, want_logits=True, want_prob=True, want_pred=True)
            only = e.run(img, want_logits=False, want_prob=False, want_pred=True)       # the pred-only path of the bench
        assert np.array_equal(out['pred'], np.argmax(out['prob'], axis=-1).astype(np.int32))
        assert np.array_equal(only['pred'], out['pred'])
        by_logits = np.argmax(out['logits'], axis=-1)
        if expect_all_zero:
            assert not out['pred'].any() and by_logits.any()


def test_small_batch_plan_is_arithmetic_neutral():
    """a handle that only ever sees small batches (the reference's sess.run of one frame's 10 slices,
    deploy_network.py:103-111) takes finer work items on the deep levels -- Winograd items of 32 instead of 64 output channels,
    one Cout block per wave in the stride-2 convs -- so that more CUs work.  Those siblings compute every output with the same
    arithmetic: the logits of slices 0..9 from a fresh N = 10 handle equal, bit for bit, those of the same slices inside the
    N = 64 bench batch on another handle (whose plan uses the large-batch tilings)."""
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.phantom import uniform_slices
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['FCN_sa']
    params = synthetic_params(arch, 1234)
    img = uniform_slices(64, 192, 208, seed=1)
    with Engine(arch, params) as big:
        ref = big.run(img, want_logits=True, want_prob=False)
        big_cfgs = big.kernel_configs()
        tail = big.run(img[:10], want_logits=True, want_prob=False)      # a large-batch handle keeps its plan for tail batches
        assert big.kernel_configs() == big_cfgs
    with Engine(arch, params) as small:
        got = small.run(img[:10], want_logits=True, want_prob=False)
        small_cfgs = small.kernel_configs()
    assert small_cfgs != big_cfgs and 301 in small_cfgs and 141 in small_cfgs, (small_cfgs, big_cfgs)
    assert np.array_equal(got['logits'], ref['logits'][:10]) and np.array_equal(got['pred'], ref['pred'][:10])
    assert np.array_equal(tail['logits'], ref['logits'][:10])


@pytest.mark.parametrize('shape', [(1, 16, 16), (2, 48, 400), (3, 272, 304), (5, 64, 16), (17, 32, 48), (2, 512, 512)])
def test_unet_bf16_at_unusual_sizes(shape):
    """The bf16-storage plan away from the tuned 256 x 256: maps smaller than a tile, tiles that do not divide the map, one
    16-pixel column, batches on both sides of the small-batch threshold -- logits within 5 % of the fp32 path's scale (bf16
    rounding of every activation, as at 256 x 256), no NaN, and the fused stem / tail launches in use."""
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['UNet_ao']
    n, h, w = shape
    img = ((cine_phantom(n, h, w, seed=h + w) - 0.3) / 0.25).astype(np.float32)
    with Engine(arch, synthetic_params(arch, 1234)) as eng:
        f32 = eng.run(img, want_logits=True)
        eng.set_precision('bf16')
        b16 = eng.run(img, want_logits=True)
        names = eng.kernel_names()
    assert names[0] == 'conv0_0+conv0_1' and names[-1] == 'up0_0+up0_1+logits' and len(names) == 20      # r04: fused stem and tail at every size
    assert np.isfinite(b16['logits']).all()
    rel = np.abs(b16['logits'] - f32['logits']).max() / np.abs(f32['logits']).max()
    assert rel < 5e-2, rel
    assert (b16['pred'] != f32['pred']).mean() < 0.05
    assert np.array_equal(np.argmax(b16['prob'], -1), b16['pred'])
