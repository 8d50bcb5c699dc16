"""What the engine may be run beside (r06; profiles/r06_notes.md section 10).

* No kernel depends on LDS it did not write: with every CU's LDS filled with NaN patterns in front of EVERY launch (UKBB_DEBUG_POISON_LDS) all models
  and precisions return identical bits.
* Every plan is right beside another stream's work: two engines on two streams, batches in flight on both, every label map equal to the single-stream
  result -- the use INTEGRATION.md section 4 describes (one handle per stream, one thread per handle).  The bf16-storage U-Net is the case that FAILED
  (85-98 % of forwards) until round 6 padded the 16-byte buffer stores of kernels_ws.hip: hipcc leaves a VALU write of the store's data registers in
  the next issue slot when the store's soffset is an SGPR (tests/test_store_hazard.py checks the ISA for that on the CPU side).
* The half-batch chains (UKBB_SPLIT_FROM; the default for the fp32 U-Net) change no bit of a forward, fp32 or bf16.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _engine(model, seed=1234):
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS[model]
    return arch, Engine(arch, synthetic_params(arch, seed))


@pytest.mark.parametrize('model,prec,shape', [('FCN_sa', 'fp32', (6, 96, 112)), ('FCN_la_4ch_seg4', 'fp32', (3, 176, 208)), ('UNet_ao', 'fp32', (4, 128, 144)),
                                              ('UNet_ao', 'bf16', (10, 304, 272)), ('UNet_ao', 'bf16', (3, 64, 96)), ('FCN_sa', 'f32x3', (4, 192, 208))])
def test_no_kernel_reads_lds_it_did_not_write(model, prec, shape, monkeypatch):
    from ukbb_cardiac_amd.phantom import uniform_slices
    n, h, w = shape
    img = ((uniform_slices(n, h, w, seed=3)[..., 0] - 0.3) / 0.25).astype(np.float32)
    monkeypatch.delenv('UKBB_DEBUG_POISON_LDS', raising=False)
    arch, eng = _engine(model)
    with eng:
        if prec != 'fp32':
            eng.set_precision(prec)
        ref = eng.run(img, want_logits=True)
        for pat in ('7FC07FC0', 'FFFFFFFF', '7F800000'):                     # bf16 NaN pairs, all ones, fp32 +inf
            monkeypatch.setenv('UKBB_DEBUG_POISON_LDS', pat)
            out = eng.run(img, want_logits=True)
            for k in ('logits', 'prob', 'pred'):
                assert np.array_equal(out[k], ref[k]), (model, prec, pat, k)
        monkeypatch.delenv('UKBB_DEBUG_POISON_LDS', raising=False)


def test_unet_lstm_cine_does_not_read_foreign_lds(monkeypatch):
    from ukbb_cardiac_amd.phantom import cine_phantom
    monkeypatch.delenv('UKBB_DEBUG_POISON_LDS', raising=False)
    frames = ((cine_phantom(13, 48, 64, seed=2)[..., 0] - 0.3) / 0.25).astype(np.float32)
    arch, eng = _engine('UNet-LSTM_ao')
    with eng:
        for prec in ('fp32', 'bf16'):
            eng.set_precision(prec)
            p0, l0 = eng.run_cine(frames)
            monkeypatch.setenv('UKBB_DEBUG_POISON_LDS', '7FC07FC0')          # (poisons in front of the U-Net launches of the cine; the ConvLSTM launches follow them)
            p1, l1 = eng.run_cine(frames)
            monkeypatch.delenv('UKBB_DEBUG_POISON_LDS', raising=False)
            assert np.array_equal(p0, p1, equal_nan=True) and np.array_equal(l0, l1), prec


@pytest.mark.parametrize('model,prec,shape,split', [('FCN_sa', 'fp32', (32, 192, 208), None), ('UNet_ao', 'fp32', (10, 304, 272), None), ('UNet_ao', 'fp32', (10, 304, 272), '0'),
                                                    ('UNet_ao', 'bf16', (10, 304, 272), None), ('UNet_ao', 'bf16', (24, 256, 256), '1'), ('FCN_sa', 'bf16', (32, 192, 208), None),
                                                    ('UNet-LSTM_ao', 'fp32', (12, 96, 128), None), ('UNet-LSTM_ao', 'bf16', (20, 128, 160), None)])
def test_plans_are_right_beside_another_streams_work(model, prec, shape, split):
    import subprocess
    n, h, w = shape
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''), PREC=prec)
    env.pop('UKBB_SPLIT_FROM', None)
    if split is not None:
        env['UKBB_SPLIT_FROM'] = split
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'two_stream_check.py'), model, str(n), str(h), str(w), '40' if model.startswith('UNet-LSTM') else '120'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == 'OK', r.stdout[-2000:]


def test_half_batch_chains_change_no_bit_of_a_forward(monkeypatch):
    """UKBB_SPLIT_FROM=k (default 1 for the fp32 U-Net, 0 otherwise): U-Net levels >= k as two half-batch launches per op on two streams -- odd and
    even batches, a batch below the split threshold; fp32 and bf16 storage."""
    from ukbb_cardiac_amd.phantom import cine_phantom
    for prec in ('fp32', 'bf16'):
        for n, H, W in ((9, 64, 96), (16, 128, 128), (3, 48, 80)):
            img = ((cine_phantom(n, H, W, seed=n)[..., 0] - 0.3) / 0.25).astype(np.float32)
            outs = {}
            for tag in ('0', '1', '2', None):
                if tag is None:
                    monkeypatch.delenv('UKBB_SPLIT_FROM', raising=False)
                else:
                    monkeypatch.setenv('UKBB_SPLIT_FROM', tag)
                arch, eng = _engine('UNet_ao', 77)
                with eng:
                    eng.set_precision(prec)
                    outs[tag] = eng.run(img, want_logits=True)
            for tag in ('1', '2', None):
                for k in ('logits', 'prob', 'pred'):
                    assert np.array_equal(outs['0'][k], outs[tag][k]), (prec, n, H, W, tag, k)
