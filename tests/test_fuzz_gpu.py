"""A slice of the randomised parity sweeps inside ``pytest -m gpu``: the generators of tools/fuzz_parity.py
(models x odd shapes x batch sizes x inputs, one case in five on ``weights.threshold_params`` -- label maps with the statistics of a
trained model) and tools/fuzz_lstm.py (UNet-LSTM ``forward_seq``), seeded, graded exactly like tests/test_gpu_parity.py: logits
within 1e-3 of the oracle's scale, label maps identical except where the fp64 restatement's own top-2 margin is below 1e-4.
The long sweeps (1 500 / 80 cases) stay tools whose output is committed under profiles/."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

pytestmark = pytest.mark.gpu

FCN_CASES, LSTM_CASES, SEED = 56, 12, 20261004


def _fcn_cases():
    import fuzz_parity
    rng = np.random.default_rng(SEED)
    return [fuzz_parity.draw_case(rng, i) for i in range(FCN_CASES)]


@pytest.mark.parametrize('chunk', range(4))
def test_seeded_fuzz_cases_against_the_c_oracle(chunk, parity_log):
    import fuzz_parity
    cases = _fcn_cases()                                       # the same list in every chunk: case i is case i whatever runs first
    worst, flips, away_all, px, thr = 0.0, 0, 0, 0, 0
    for i in range(chunk, FCN_CASES, 4):
        c = cases[i]
        ok, rel, fl, away, npx = fuzz_parity.grade_case(c)
        assert ok, 'case %d: %s seed %d %dx%dx%d: relative logits error %.2e, %d label flips away from a tie' % (
            i, c['name'], c['wseed'], c['n'], c['h'], c['w'], rel, away)
        worst = max(worst, rel); flips += fl; away_all += away; px += npx; thr += c['threshold_model']
    assert flips <= max(4, 40 * px // 1000000)
    parity_log(cases=len(range(chunk, FCN_CASES, 4)), threshold_model_cases=int(thr), worst_rel_logits_err=worst, label_flips=flips,
               away_from_tie=away_all, pixels=px)


def test_the_seeded_list_covers_every_model_and_the_trained_like_statistics():
    cases = _fcn_cases()
    assert {c['name'] for c in cases} == {'FCN_sa', 'FCN_la_2ch', 'FCN_la_4ch', 'FCN_la_4ch_seg4', 'UNet_ao'}
    assert sum(c['threshold_model'] for c in cases) >= 3 and any(min(c['h'], c['w']) == 16 for c in cases)
    assert any(c['n'] >= 5 for c in cases) and any(c['h'] * c['w'] > 160 * 160 for c in cases)


def test_seeded_fuzz_cases_of_the_unet_lstm_against_the_fp64_restatement(parity_log):
    import fuzz_lstm
    rng = np.random.default_rng(SEED + 1)
    worst, flips, px = 0.0, 0, 0
    for i in range(LSTM_CASES):
        c = fuzz_lstm.draw_case(rng)
        ok, rel, fl, away, npx = fuzz_lstm.grade_case(c)
        assert ok, 'case %d: seed %d %dx9x%dx%d: relative logits error %.2e, %d label flips away from a tie' % (i, c['wseed'], c['n'], c['h'], c['w'], rel, away)
        worst = max(worst, rel); flips += fl; px += npx
    parity_log(cases=LSTM_CASES, worst_rel_logits_err=worst, label_flips=flips, pixels=px)
