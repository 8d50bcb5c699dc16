"""BASELINE.json configs[3] / SURVEY.md 8(d) config 4: a cohort of synthetic short-axis subjects (192x208x10x50 each)
"generated on device from seed = subject id", every subject through the REAL device stages of the sequence loop
(common/deploy_network.py:83-131) -- exact percentiles (``ukbb_fcn_select_kth``), clip / rescale / pad / transpose
(``ukbb_fcn_rescale_pack``), the FCN forward in 128-slice batches, label unpack + per-frame class counts
(``ukbb_fcn_unpack_labels``), uint8 label volume back to pinned host memory -- i.e. ``SubjectPipeline`` with the H2D copy of the
file's voxels replaced by ``ukbb_fcn_synth_volume`` on the copy-in stream (host gzip excluded and stated).

Subject i goes to rank ``i mod G`` (``shard.subjects_for_shard``); ranks share nothing, there is no collective.

``synth_volume_host`` is the numpy twin of the device generator (integer arithmetic only, bit-identical), so a test can hand
the oracle the very volume the GPU segmented without reading it back.
"""
import time

import numpy as np

from . import _lib
from .shard import subjects_for_shard

SHAPE = (192, 208, 10, 50)                                   # BASELINE.json configs[0] / configs[3]
_GOLDEN = 0x9E3779B97F4A7C15
_M64 = (1 << 64) - 1


def synth_volume_host(seed, shape=SHAPE):
    """What ``ukbb_fcn_synth_volume(seed, n, ...)`` writes, as a Fortran-ordered float32 (X,Y,Z,T) array: voxel i (x fastest)
    = a * b / 2048 with a, b the two low 12-bit fields of splitmix64's finaliser of seed * 0x9E3779B97F4A7C15 + i."""
    n = int(np.prod(shape))
    with np.errstate(over='ignore'):
        z = np.uint64((int(seed) * _GOLDEN) & _M64) + np.arange(n, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    a = (z & np.uint64(0xFFF)).astype(np.uint32)
    b = ((z >> np.uint64(12)) & np.uint64(0xFFF)).astype(np.uint32)
    v = (a * b).astype(np.float32) * np.float32(1.0 / 2048.0)          # a * b < 2^24: exact
    return v.reshape(shape, order='F')


def device_fill(seed):
    """fill(d_ptr, n, stream) callable for ``SubjectPipeline.submit_generated``."""
    def fill(d_ptr, n, stream):
        _lib.check(_lib.lib.ukbb_fcn_synth_volume(int(seed), int(n), d_ptr, stream), 'ukbb_fcn_synth_volume')
    return fill


def run_cohort(engine, subject_ids, shape=SHAPE, rank=0, world=1, keep=(), batch_slices=128, depth=3, on_result=None):
    """Segments the subjects of ``subject_ids`` this rank owns (i mod world == rank), ``depth - 1`` in flight.

    Returns a dict: subjects, slices, seconds (wall, first submit to last result collected, device synchronised), es_frames
    {subject: ES frame picked from the per-frame class counts, deploy_network.py:125-130}, kept {subject: (labels uint8 (X,Y,Z,T),
    counts [T, n_class], clip)} for the ids in ``keep``, and the device's free memory before / after (bytes)."""
    import torch
    from .device_pipeline import pick_ed_es_from_counts
    from .subject_pipeline import SubjectPipeline
    mine = subjects_for_shard(list(subject_ids), rank, world)
    dev = torch.device('cuda', engine.device)
    keep = set(keep)
    pipe = SubjectPipeline(engine, shape, batch_slices=batch_slices, depth=depth, extra_inputs=0, pinned_inputs=False)
    X, Y, Z, T = shape
    es, kept = {}, {}

    def take(sid):
        r = pipe.collect(copy=sid in keep)
        es[sid] = pick_ed_es_from_counts(r.counts, 'sa')[1]
        if sid in keep:
            kept[sid] = (r.labels, r.counts, r.clip)
        if on_result is not None:
            on_result(sid, r)
        r.done()
    torch.cuda.synchronize(dev)
    free0 = torch.cuda.mem_get_info(dev)[0]
    order = []
    t0 = time.perf_counter()
    for sid in mine:
        if pipe.pending() >= depth - 1:
            take(order.pop(0))
        pipe.submit_generated(shape, device_fill(sid))
        order.append(sid)
    while order:
        take(order.pop(0))
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    free1 = torch.cuda.mem_get_info(dev)[0]
    return {'subjects': len(mine), 'slices': len(mine) * Z * T, 'seconds': dt, 'es_frames': es, 'kept': kept,
            'free_before': int(free0), 'free_after': int(free1), 'pipeline': pipe}


def stage_times(engine, shape=SHAPE, seed=0, batch_slices=128, repeats=5):
    """One subject at a time, nothing overlapped: HIP-event time of each device stage (ms, median over ``repeats``).  What the
    pipelined cohort hides behind the network is visible here: {'generate', 'percentiles', 'pack', 'network', 'unpack', 'labels_d2h'}."""
    import ctypes as C
    import torch
    from .device_pipeline import lerp_like_numpy, percentile_ranks
    from .pipeline import pad_amounts
    dev = torch.device('cuda', engine.device)
    X, Y, Z, T = shape
    n = X * Y * Z * T
    X2, Y2, x_pre, _, y_pre, _ = pad_amounts(X, Y)
    nsl, px = Z * T, X2 * Y2
    s = torch.cuda.Stream(dev)
    d_vol = torch.empty(n, dtype=torch.float32, device=dev)
    d_batch = torch.empty(nsl * px, dtype=torch.float32, device=dev)
    d_pred = torch.empty(nsl * px, dtype=torch.int32, device=dev)
    d_lab = torch.empty(n, dtype=torch.uint8, device=dev)
    d_cnt = torch.empty(T * 16, dtype=torch.int64, device=dev)
    pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    engine.reserve(min(batch_slices, nsl), X2, Y2)
    names = ['generate', 'percentiles', 'pack', 'network', 'unpack', 'labels_d2h']
    acc = {k: [] for k in names}
    cs = s.cuda_stream
    for _ in range(repeats + 1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        with torch.cuda.stream(s):
            ev[0].record(s)
            _lib.check(_lib.lib.ukbb_fcn_synth_volume(int(seed), n, d_vol.data_ptr(), cs), 'ukbb_fcn_synth_volume')
            ev[1].record(s)
            ranks, gammas = [], []
            for q in (1, 99):
                k, k1, g = percentile_ranks(n, q)
                ranks += [k, k1]
                gammas.append(g)
            r = (C.c_uint64 * 4)(*ranks)
            out = np.empty(4, np.float32)
            _lib.check(_lib.lib.ukbb_fcn_select_kth(d_vol.data_ptr(), n, r, 4, _lib.f32ptr(out), cs), 'ukbb_fcn_select_kth')
            lo, hi = (lerp_like_numpy(out[2 * i], out[2 * i + 1], gammas[i]) for i in range(2))
            ev[2].record(s)
            _lib.check(_lib.lib.ukbb_fcn_rescale_pack(d_vol.data_ptr(), X, Y, Z, T, 1, X, X * Y, X * Y * Z, float(lo), float(hi),
                                                      X2, Y2, x_pre, y_pre, d_batch.data_ptr(), cs), 'ukbb_fcn_rescale_pack')
            ev[3].record(s)
            for i in range(0, nsl, batch_slices):
                m = min(batch_slices, nsl - i)
                engine.run_device(d_batch.data_ptr() + 4 * i * px, m, X2, Y2, pred_ptr=d_pred.data_ptr() + 4 * i * px, stream=cs)
            ev[4].record(s)
            _lib.check(_lib.lib.ukbb_fcn_unpack_labels(d_pred.data_ptr(), X, Y, Z, T, X2, Y2, x_pre, y_pre, engine.arch.n_class,
                                                       d_lab.data_ptr(), d_cnt.data_ptr(), cs), 'ukbb_fcn_unpack_labels')
            ev[5].record(s)
            pin.copy_(d_lab, non_blocking=True)
            ev[6].record(s)
        s.synchronize()
        for i, k in enumerate(names):
            acc[k].append(ev[i].elapsed_time(ev[i + 1]))
    return {k: round(float(np.median(v[1:])), 4) for k, v in acc.items()}
