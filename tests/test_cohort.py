"""BASELINE.json configs[3]: the synthetic cohort generated on the device from seed = subject id, through the
real device subject pipeline (ukbb_cardiac_amd/synthetic_cohort.py).

CPU: the numpy twin of the device generator against a pure-Python restatement of the same integer recipe (known answers), the
sharding of the subject list.  GPU: device generator == numpy twin bit for bit; >= 200 full-size subjects through the pipeline with
three sampled subjects' label volumes and ES pick graded against ``oracle.fcn_oracle.deploy_sequence`` driving the C oracle
(common/deploy_network.py:83-131), every other subject's ES pick against the pick recomputed from its own label volume, device
memory before / after, and a second pass over the same subject ids bit-identical to the first.
"""
import json
import os
import time

import numpy as np
import pytest

from ukbb_cardiac_amd import synthetic_cohort as sc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEAR_TIE = 1e-4


def _voxel_py(seed, i):
    m = (1 << 64) - 1
    z = (seed * 0x9E3779B97F4A7C15 + i) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    z ^= z >> 31
    return ((z & 0xFFF) * ((z >> 12) & 0xFFF)) / 2048.0


@pytest.mark.parametrize('seed', [0, 1, 999, 2 ** 40 + 3])
def test_numpy_twin_of_the_generator_known_answers(seed):
    shape = (5, 7, 3, 2)
    v = sc.synth_volume_host(seed, shape)
    assert v.dtype == np.float32 and v.flags.f_contiguous and v.shape == shape
    flat = v.reshape(-1, order='F')
    for i in (0, 1, 2, 17, flat.size - 1):
        assert float(flat[i]) == _voxel_py(seed, i)                  # a * b < 2^24 and the 2^-11 scale: exact in float32
    assert 0.0 <= float(flat.min()) and float(flat.max()) < 8192.0
    if seed:
        assert not np.array_equal(v, sc.synth_volume_host(seed - 1, shape))


def test_generator_statistics_make_the_percentile_clip_matter():
    v = sc.synth_volume_host(3, (64, 64, 4, 4))
    lo, med, hi = np.percentile(v, (1, 50, 99))
    assert lo < 30 and 1000 < med < 2200 and 6000 < hi < 7800 and v.max() > hi * 1.1      # heavy right tail like an MR magnitude


def test_cohort_split_is_complete_and_disjoint():
    from ukbb_cardiac_amd.shard import subjects_for_shard
    ids = list(range(1000))
    for g in (1, 2, 4, 8):
        parts = [subjects_for_shard(ids, r, g) for r in range(g)]
        assert sorted(sum(parts, [])) == ids and max(map(len, parts)) - min(map(len, parts)) <= 1


# ---- GPU -------------------------------------------------------------------------------------------------------------

def _engine():
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['FCN_sa']
    params = synthetic_params(arch, 1234)
    return arch, params, Engine(arch, params, device=0)


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1, 3, 4, 1027, 192 * 208 * 10])
def test_device_generator_equals_numpy_twin(n):
    import torch
    from ukbb_cardiac_amd import _lib
    for seed in (0, 5, 123456789):
        t = torch.full((n + 4,), -1.0, dtype=torch.float32, device='cuda:0')
        _lib.check(_lib.lib.ukbb_fcn_synth_volume(seed, n, t.data_ptr(), 0), 'synth')
        got = t.cpu().numpy()
        want = sc.synth_volume_host(seed, (n, 1, 1, 1)).reshape(-1)
        assert np.array_equal(got[:n], want) and np.all(got[n:] == -1.0)           # nothing written behind the volume
    with pytest.raises(_lib.UkbbFcnError):
        _lib.check(_lib.lib.ukbb_fcn_synth_volume(1, 0, t.data_ptr(), 0), 'synth')


@pytest.mark.gpu
def test_config3_cohort_of_full_size_subjects_through_the_device_pipeline(parity_log):
    import torch
    from oracle import c_oracle, fcn_oracle as O
    from ukbb_cardiac_amd.device_pipeline import pick_ed_es_from_counts
    from ukbb_cardiac_amd.weights import pack_flat
    arch, params, eng = _engine()
    flat = pack_flat(arch, params)
    n_subj, sampled = 208, (3, 101, 207)
    X, Y, Z, T = sc.SHAPE
    checks = {'es_from_labels': 0}

    def on_result(sid, r):
        # every subject: the counts the ES pick uses are the counts of the label volume that came back (deploy_network.py:125-130)
        assert r.labels.shape == sc.SHAPE and r.labels.dtype == np.uint8 and r.image is None
        c1 = (r.labels == 1).sum(axis=(0, 1, 2))
        assert np.array_equal(c1, r.counts[:, 1]) and int(r.counts.sum()) == X * Y * Z * T
        checks['es_from_labels'] += 1
    with eng:
        sc.run_cohort(eng, range(2))                                           # warm-up (plans, allocator)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        rec = sc.run_cohort(eng, range(n_subj), keep=sampled, on_result=on_result)
        pipe = rec.pop('pipeline')
        del pipe
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
        # a second pass over the sampled ids alone (other neighbours in flight, other slots): identical bits
        again = sc.run_cohort(eng, sampled, keep=sampled)
    assert rec['subjects'] == n_subj and rec['slices'] == n_subj * Z * T and checks['es_from_labels'] == n_subj
    assert abs(free0 - free1) < 50e6, 'device memory moved by %.1f MB over %d subjects' % ((free0 - free1) / 1e6, n_subj)
    rate = rec['slices'] / rec['seconds']
    report = {'subjects': n_subj, 'slices_per_s': round(rate, 1), 'device_memory_delta_mb': round((free0 - free1) / 1e6, 2), 'sampled': {}}
    t_or = time.time()
    for sid in sampled:
        lab, counts, clip = rec['kept'][sid]
        lab2, counts2, clip2 = again['kept'][sid]
        assert np.array_equal(lab, lab2) and np.array_equal(counts, counts2) and clip == clip2
        vol = sc.synth_volume_host(sid)                                        # the volume the GPU generated, rebuilt from integers
        lo, hi = np.percentile(vol, (1, 99))
        assert (lo, hi) == clip                                                 # exact order statistics + numpy's interpolation
        ref_pred, _, ed, es = O.deploy_sequence(vol.copy(), lambda b: c_oracle.forward(arch, flat, b, want_logits=False)[2], 'sa')
        bad = lab != ref_pred
        nbad = int(bad.sum())
        assert nbad <= 40 * bad.size // 1000000, 'subject %d: %d label disagreements in %d voxels' % (sid, nbad, bad.size)
        away = 0
        for z, t in np.argwhere(bad.any(axis=(0, 1)))[:8]:                     # fp64 arbitration: each must sit on a numerical tie
            sl = np.clip(vol[:, :, z, t], lo, hi)
            net_in = ((sl.astype(np.float32).astype(np.float64) - np.float64(lo)) / (np.float64(hi) - np.float64(lo))).astype(np.float32)
            ref64 = O.build_FCN(net_in[None, :, :, None], params, arch.n_class, dtype=np.float64)
            away += int((bad[:, :, z, t] & (O.top2_margin(ref64)[0] > NEAR_TIE)).sum())
        assert away == 0, 'subject %d: %d label disagreements away from a numerical tie' % (sid, away)
        got_es = pick_ed_es_from_counts(counts, 'sa')[1]
        assert got_es == rec['es_frames'][sid]
        ref_c1 = np.sum(ref_pred == 1, axis=(0, 1, 2))
        if np.sort(ref_c1)[1] - np.sort(ref_c1)[0] > nbad:                     # the pick is not inside the tie noise
            assert got_es == es
        report['sampled'][str(sid)] = {'label_disagreements': nbad, 'away_from_tie': away, 'es_frame': int(got_es), 'oracle_es_frame': int(es)}
    report['c_oracle_seconds'] = round(time.time() - t_or, 1)
    parity_log(**{k: v for k, v in report.items() if k != 'sampled'}, sampled=json.dumps(report['sampled']))
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'cohort_report.json'), 'w') as f:
        json.dump(report, f, indent=1, sort_keys=True)
