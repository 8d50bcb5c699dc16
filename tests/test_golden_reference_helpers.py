"""Golden vectors produced by the REFERENCE's own numpy function bodies
(tests/golden/make_golden.py, section 1) against (a) the oracle restatement and
(b) the product host code.  Bit-exact."""
import os

import numpy as np
import pytest

from oracle import fcn_oracle as O
from ukbb_cardiac_amd import image_utils as P

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ref_numpy_helpers.npz'))


@pytest.mark.parametrize('sz', [3, 7, 15, 31])
def test_linear_kernels(sz):
    assert np.array_equal(O.linear_1d(sz), G['linear_1d_%d' % sz])
    assert np.array_equal(O.linear_2d(sz), G['linear_2d_%d' % sz])


def test_linear_known_answers_from_reference_text():
    # common/network.py:117-124
    assert O.linear_1d(3).tolist() == [0.5, 1.0, 0.5]
    assert O.linear_1d(7).tolist() == [0.25, 0.5, 0.75, 1.0, 0.75, 0.5, 0.25]
    with pytest.raises(NotImplementedError):
        O.linear_1d(4)


@pytest.mark.parametrize('impl', [O, P])
def test_rescale_intensity(impl):
    v = G['rescale_in'].copy()
    out = impl.rescale_intensity(v, (1, 99))
    assert out.dtype == G['rescale_out'].dtype and np.array_equal(out, G['rescale_out'])
    assert np.array_equal(v, G['rescale_in_after'])            # in-place clip quirk
    assert abs(out.min()) < 1e-6 and abs(out.max() - 1.0) < 1e-6   # clip value is rounded to f32 first
    v = G['rescale_in'].copy()
    assert np.array_equal(impl.rescale_intensity(v, (2.0, 98.0)), G['rescale_out_2_98'])


@pytest.mark.parametrize('impl', [O, P])
def test_normalise_intensity(impl):
    out = impl.normalise_intensity(G['normalise_in'].copy(), 10.0)
    assert out.dtype == G['normalise_out'].dtype and np.array_equal(out, G['normalise_out'])


@pytest.mark.parametrize('impl', [O, P])
def test_dice(impl):
    got = np.array([impl.np_categorical_dice(G['dice_a'], G['dice_b'], k) for k in range(4)])
    assert np.array_equal(got.astype(np.float32), G['dice_k'])


def test_pad_arithmetic_known_answers():
    # common/deploy_network.py:97-99
    assert O.pad_to_multiple(162, 204) == (176, 208, 7, 7, 2, 2)
    assert O.pad_to_multiple(192, 208) == (192, 208, 0, 0, 0, 0)
    assert O.pad_to_multiple(163, 205) == (176, 208, 6, 7, 1, 2)
    # common/deploy_network_ao.py:105-107
    assert O.pad_to_fixed(240, 196) == (256, 256, 8, 8, 30, 30)


def test_aortic_window_known_answers():
    # common/deploy_network_ao.py:130-158
    w = O.aortic_window_weights(5, 0.1)
    assert len(w) == 9 and w[4] == 1.0 and np.allclose(w, [(1 - abs(t - 4) / 5.0) ** 0.1 for t in range(9)])
    assert O.aortic_window_indices(0, 50) == [46, 47, 48, 49, 0, 1, 2, 3, 4]
    assert O.aortic_window_indices(49, 50) == [45, 46, 47, 48, 49, 0, 1, 2, 3]


def test_es_rule():
    # common/deploy_network.py:127-130
    pred = np.zeros((4, 4, 2, 3))
    pred[:2, :, :, 0] = 1; pred[:1, :, :, 1] = 1; pred[:3, :, :, 2] = 1
    assert O.pick_es_frame(pred, 'sa') == 1
    assert O.pick_es_frame(pred, 'la_4ch', seg4=True) == 1
    assert O.pick_es_frame(pred, 'la_2ch') == 2
    assert O.pick_es_frame(pred, 'la_4ch') == 2
