"""The C restatement (oracle/fcn_oracle.c) against the numpy fp64 oracle and the
committed goldens."""
import os
import time

import numpy as np
import pytest

from oracle import c_oracle, fcn_oracle as O
from ukbb_cardiac_amd.arch import MODELS
from ukbb_cardiac_amd.weights import pack_flat, synthetic_params

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('tag', ['fcn_sa_2x32x48', 'fcn_sa_1x192x208', 'fcn_seg4_1x80x112', 'fcn_la4ch_2x48x16',
                                 'fcn_la2ch_1x176x208'])
def test_c_fcn_vs_golden(tag):
    g = np.load(os.path.join(GOLD, tag + '.npz'))
    arch = MODELS[str(g['model'])]
    flat = pack_flat(arch, synthetic_params(arch, 1234))
    lg, pr, pd = c_oracle.forward(arch, flat, g['image'], want_prob=True)
    ref = g['logits64']
    assert np.abs(lg - ref).max() <= 1e-3 * np.abs(ref).max()
    bad = pd != g['pred64']
    assert not np.any(bad & (g['margin64'] > 1e-4))
    assert bad.sum() <= int((g['margin64'] <= 1e-4).sum())
    assert np.allclose(pr.sum(-1), 1.0, atol=1e-5)


def test_c_unet_vs_golden():
    g = np.load(os.path.join(GOLD, 'unet_ao_2x64x96.npz'))
    arch = MODELS['UNet_ao']
    flat = pack_flat(arch, synthetic_params(arch, 1234))
    lg, pr, pd = c_oracle.forward(arch, flat, g['image'], want_prob=True)
    ref = g['logits64']
    assert np.abs(lg - ref).max() <= 1e-3 * np.abs(ref).max()
    assert np.array_equal(pd, g['pred64'])
    assert np.abs(pr - g['prob64']).max() <= 1e-4


def test_c_oracle_rejects_unpadded_shape():
    arch = MODELS['FCN_sa']
    flat = pack_flat(arch, synthetic_params(arch, 1234))
    with pytest.raises(RuntimeError):
        c_oracle.forward(arch, flat, np.zeros((1, 30, 32, 1), np.float32))
