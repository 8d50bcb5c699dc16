"""measures.py (counts -> clinical measures) against the formulas of the evaluation scripts applied to the
label volume itself (short_axis/eval_ventricular_volume.py:40-71, aortic/eval_aortic_area.py:60-78)."""
import numpy as np

from ukbb_cardiac_amd.measures import aortic_areas, ventricular_volumes


def test_ventricular_volumes_equal_the_volume_based_formulas():
    rng = np.random.default_rng(3)
    seg = rng.integers(0, 4, size=(40, 36, 5, 12)).astype(np.float64)
    seg[..., 7][seg[..., 7] == 1] = 0                                   # frame 7 has the smallest LV cavity
    pixdim = np.array([1, 1.8269, 1.8269, 10.0, 0.0305, 0, 0, 0], np.float32)
    counts = np.stack([[np.sum(seg[..., t] == c) for c in range(4)] for t in range(seg.shape[3])])
    got = ventricular_volumes(counts, pixdim)
    vpp = pixdim[1:4][0] * pixdim[1:4][1] * pixdim[1:4][2] * 1e-3
    vol_t = np.sum(seg == 1, axis=(0, 1, 2)) * vpp
    es = int(np.argmin(vol_t))
    assert got['ES_frame'] == es == 7
    assert got['LVEDV'] == np.sum(seg[:, :, :, 0] == 1) * vpp
    assert got['LVESV'] == np.sum(seg[:, :, :, es] == 1) * vpp
    assert got['LVEDM'] == np.sum(seg[:, :, :, 0] == 2) * vpp * 1.05
    assert got['RVESV'] == np.sum(seg[:, :, :, es] == 3) * vpp
    hr = 60.0 / (seg.shape[3] * pixdim[4])
    assert got['LVEF'] == (got['LVEDV'] - got['LVESV']) / got['LVEDV'] * 100
    assert got['LVCO'] == (got['LVEDV'] - got['LVESV']) * hr * 1e-3


def test_aortic_areas():
    rng = np.random.default_rng(4)
    seg = rng.integers(0, 3, size=(30, 30, 1, 9))
    pixdim = np.array([1, 1.6, 1.6, 6.0, 0.01, 0, 0, 0], np.float32)
    counts = np.stack([[np.sum(seg[..., t] == c) for c in range(3)] for t in range(9)])
    got = aortic_areas(counts, pixdim, central_pp=40.0)
    A = np.sum(seg == 1, axis=(0, 1, 2)) * (pixdim[1] * pixdim[2])
    assert got['AAo']['max area'] == A.max() and got['AAo']['min area'] == A.min()
    assert got['AAo']['distensibility'] == (A.max() - A.min()) / (A.min() * 40.0) * 1e3
