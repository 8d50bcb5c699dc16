"""Multi-GPU path = batch split with no collective.  World-size-2 gloo test on
CPU: two ranks share one data_dir, each runs the deployment loop on its shard
with a stub forward; together they cover every subject exactly once."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ukbb_cardiac_amd.shard import shard_from_env, subjects_for_shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_properties():
    subs = ['s%03d' % i for i in range(11)]
    for g in (1, 2, 4, 8):
        parts = [subjects_for_shard(subs, r, g) for r in range(g)]
        flat = sorted(sum(parts, []))
        assert flat == subs                                    # complete, disjoint
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        subjects_for_shard(subs, 2, 2)


def test_shard_from_env(monkeypatch):
    monkeypatch.delenv('RANK', raising=False); monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('UKBB_SHARD_INDEX', raising=False); monkeypatch.delenv('UKBB_NUM_SHARDS', raising=False)
    assert shard_from_env() == (0, 1)
    monkeypatch.setenv('RANK', '3'); monkeypatch.setenv('WORLD_SIZE', '8')
    assert shard_from_env() == (3, 8)
    monkeypatch.setenv('UKBB_SHARD_INDEX', '1'); monkeypatch.setenv('UKBB_NUM_SHARDS', '2')
    assert shard_from_env() == (1, 2)


def _worker(rank, world, data_dir, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from test_host_pipeline import stub_forward
    from ukbb_cardiac_amd import deploy_network as DN
    F, _ = DN.define_flags().parse(['--data_dir', data_dir, '--num_shards', str(world), '--shard_index', str(rank)])
    done = DN.run(F, stub_forward, log=lambda *_: None)
    gathered = [None] * world
    dist.all_gather_object(gathered, done)                    # bookkeeping only; no data-path collective
    n = torch.tensor([len(done)], dtype=torch.int64)
    dist.all_reduce(n)
    if rank == 0:
        flat = sorted(sum(gathered, []))
        assert flat == sorted(os.listdir(data_dir)), (flat, os.listdir(data_dir))
        assert int(n.item()) == len(flat)
        assert not set(gathered[0]) & set(gathered[1])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_deploy(tmp_path):
    from test_host_pipeline import _write_subject
    from ukbb_cardiac_amd import nifti
    for i in range(5):
        _write_subject(tmp_path, 'subj%02d' % i, 'sa', (20, 28, 2, 3), 10 + i)
    port = 29500 + (os.getpid() % 400)
    mp.spawn(_worker, args=(2, str(tmp_path), port), nprocs=2, join=True)
    for i in range(5):
        seg = nifti.load(str(tmp_path / ('subj%02d' % i) / 'seg_sa.nii.gz'))
        assert seg.data.shape == (20, 28, 2, 3)


def test_launch_reports_killed_and_failed_workers(tmp_path, capfd):
    """A worker killed by a signal (negative Popen returncode, e.g. a GPU fault aborting with SIGABRT) or exiting
    non-zero must fail the launcher -- with skip-if-exists a silent 0 would look like a finished run."""
    from ukbb_cardiac_amd import shard
    assert shard.exit_status(0) == 0 and shard.exit_status(3) == 3 and shard.exit_status(-6) == 134 and shard.exit_status(-11) == 139
    script = tmp_path / 'w.py'
    script.write_text(
        'import os, signal, sys\n'
        'i = int(os.environ["UKBB_SHARD_INDEX"]); n = int(os.environ["UKBB_NUM_SHARDS"])\n'
        'assert n == 3 and os.environ["HIP_VISIBLE_DEVICES"] == str(i) and "RANK" not in os.environ\n'
        'mode = sys.argv[1]\n'
        'if mode == "kill" and i == 1: os.kill(os.getpid(), signal.SIGABRT)\n'
        'if mode == "fail" and i == 2: sys.exit(7)\n')
    assert shard.launch(3, [str(script), 'ok']) == 0
    assert shard.launch(3, [str(script), 'kill']) == 128 + 6
    assert 'shard 1 of 3' in capfd.readouterr().err
    assert shard.launch(3, [str(script), 'fail']) == 7
    assert 'shard 2 of 3' in capfd.readouterr().err


def test_shards_per_gpu_and_default_device(tmp_path, monkeypatch):
    from ukbb_cardiac_amd import deploy_network as DN, deploy_network_ao as DA, shard
    script = tmp_path / 'w.py'
    script.write_text(
        'import os, sys\n'
        'i = int(os.environ["UKBB_SHARD_INDEX"])\n'
        'open(os.path.join(sys.argv[1], "s%d" % i), "w").write(os.environ["HIP_VISIBLE_DEVICES"] + " " + os.environ["UKBB_NUM_SHARDS"])\n')
    assert shard.launch(2, [str(script), str(tmp_path)], shards_per_gpu=2) == 0
    assert [open(tmp_path / ('s%d' % i)).read() for i in range(4)] == ['0 4', '0 4', '1 4', '1 4']
    with pytest.raises(SystemExit):
        shard.main(['--gpus', '2', '--bogus', '1', '--', 'x.py'])
    # torchrun: every rank sees all GPUs -> bind to LOCAL_RANK; shard.launch workers see one GPU -> device 0
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'UKBB_SHARD_INDEX', 'UKBB_NUM_SHARDS'):
        monkeypatch.delenv(k, raising=False)
    assert shard.default_device() == 0
    monkeypatch.setenv('RANK', '5'); monkeypatch.setenv('WORLD_SIZE', '8'); monkeypatch.setenv('LOCAL_RANK', '5')
    for mod in (DN, DA):
        F, _ = mod.define_flags().parse(['--data_dir', 'x'])
        assert (F.device, F.shard_index, F.num_shards) == (5, 5, 8)
        F, _ = mod.define_flags().parse(['--data_dir', 'x', '--device', '2'])
        assert F.device == 2
    monkeypatch.setenv('UKBB_SHARD_INDEX', '1'); monkeypatch.setenv('UKBB_NUM_SHARDS', '2')
    F, _ = DN.define_flags().parse(['--data_dir', 'x'])
    assert (F.device, F.shard_index, F.num_shards) == (0, 1, 2)


# ---- work stealing on top of the static split (claim files) ------------------------------------------------------------------

def test_claim_files_are_exclusive_and_stale_ones_are_swept(tmp_path, monkeypatch):
    import subprocess
    import time
    from ukbb_cardiac_amd import nifti, shard
    d = str(tmp_path)
    assert shard.try_claim(d, 'seg_sa') and not shard.try_claim(d, 'seg_sa')          # O_EXCL: one winner (our own live claim is not stale)
    assert shard.try_claim(d, 'seg_la_2ch')                                            # another sequence of the same subject is another claim
    assert open(shard.claim_path(d, 'seg_sa')).read().split() == [nifti._host_tag(), str(os.getpid())]
    shard.release_claim(d, 'seg_sa')
    shard.release_claim(d, 'seg_sa')                                                   # idempotent
    assert shard.try_claim(d, 'seg_sa')
    shard.release_claim(d, 'seg_sa')
    # a claim of a process that no longer exists, same host and pid namespace: swept at once
    p = subprocess.Popen([sys.executable, '-c', 'pass'])
    p.wait()
    open(shard.claim_path(d, 'seg_sa'), 'w').write('%s %d\n' % (nifti._host_tag(), p.pid))
    assert shard.try_claim(d, 'seg_sa')
    shard.release_claim(d, 'seg_sa')
    # a claim of a LIVE process is respected
    open(shard.claim_path(d, 'seg_sa'), 'w').write('%s %d\n' % (nifti._host_tag(), os.getppid()))
    assert not shard.try_claim(d, 'seg_sa')
    # a claim written on another host (pid means nothing here): respected while young, swept by age
    open(shard.claim_path(d, 'seg_sa'), 'w').write('hdeadbeef00 %d\n' % p.pid)
    assert not shard.try_claim(d, 'seg_sa')
    old = time.time() - shard.CLAIM_MAX_AGE_S - 10
    os.utime(shard.claim_path(d, 'seg_sa'), (old, old))
    assert shard.try_claim(d, 'seg_sa')
    shard.release_claim(d, 'seg_sa')
    # an empty claim (its writer died between create and write) only counts as abandoned after a minute
    open(shard.claim_path(d, 'seg_sa'), 'w').close()
    assert not shard.try_claim(d, 'seg_sa')
    os.utime(shard.claim_path(d, 'seg_sa'), (time.time() - 120, time.time() - 120))
    assert shard.try_claim(d, 'seg_sa')


def test_stealing_order_and_cpu_sets():
    from ukbb_cardiac_amd import shard
    subs = ['s%02d' % i for i in range(10)]
    for g in (2, 3, 8):
        for r in range(g):
            o = shard.stealing_order(subs, r, g)
            own = subjects_for_shard(subs, r, g)
            assert sorted(o) == subs and o[:len(own)] == own                           # whole list, own static share first and in order
            nxt = subjects_for_shard(subs, (r + 1) % g, g)
            assert o[len(own):len(own) + len(nxt)] == nxt[::-1]                        # then the neighbour's share from its tail
    assert shard.split_cpus(range(8), 3) == [[0, 1, 2], [3, 4, 5], [6, 7]]
    assert shard.split_cpus([5, 3, 9], 4) == [[3, 5, 9]] * 4                           # fewer CPUs than workers: everyone gets all of them
    assert sum(shard.split_cpus(range(256), 8), []) == list(range(256))
    assert 1 <= shard.io_threads_for(32, 8) <= 16 and shard.io_threads_for(1, 8) == 1
    q = shard.ClaimQueue('/nonexistent', subs, 1, 2, 'seg_sa', stealing=False)
    assert list(q) == subjects_for_shard(subs, 1, 2) and q.take('s01') and not q.held  # static mode: no files touched


def _stealing_worker(rank, world, data_dir, port):
    import time
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from test_host_pipeline import stub_forward
    from ukbb_cardiac_amd import deploy_network as DN
    F, _ = DN.define_flags().parse(['--data_dir', data_dir, '--num_shards', str(world), '--shard_index', str(rank), '--output_csv',
                                    os.path.join(os.path.dirname(data_dir), 'sa.csv')])
    assert F.work_stealing

    def forward(batch):                                        # rank 0 drew the large subjects AND is slow: the case static i mod G loses on
        if rank == 0:
            time.sleep(0.02 * batch.shape[0])
        return stub_forward(batch)
    dist.barrier()
    done = DN.run(F, forward, log=lambda *_: None)
    gathered = [None] * world
    dist.all_gather_object(gathered, done)                    # bookkeeping only; no data-path collective
    if rank == 0:
        subs = sorted(os.listdir(data_dir))
        assert sorted(sum(gathered, [])) == subs, gathered     # complete ...
        assert not set(gathered[0]) & set(gathered[1])         # ... and disjoint
        static1 = subjects_for_shard(subs, 1, 2)
        assert set(static1) <= set(gathered[1]) and len(gathered[1]) > len(static1), gathered     # the fast worker took over part of the slow one's share
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_deploy_with_unequal_subjects_steals_work(tmp_path):
    from test_host_pipeline import _write_subject
    from ukbb_cardiac_amd import measures, nifti
    data = tmp_path / 'data'
    data.mkdir()
    for i in range(12):                                        # even indices (rank 0's share): 24 frames; odd: 3
        _write_subject(data, 'subj%02d' % i, 'sa', (20, 28, 2, 24 if i % 2 == 0 else 3), 10 + i)
    port = 29500 + (os.getpid() % 400) + 7
    mp.spawn(_stealing_worker, args=(2, str(data), port), nprocs=2, join=True)
    for i in range(12):
        d = data / ('subj%02d' % i)
        assert nifti.load(str(d / 'seg_sa.nii.gz')).data.shape == (20, 28, 2, 24 if i % 2 == 0 else 3)
        assert not [f for f in os.listdir(d) if f.startswith('.claim') or '.tmp.' in f]           # nothing left behind
    # the two workers' CSV parts merge into one table with every subject exactly once, stolen or not
    assert measures.merge_shard_csv(str(tmp_path / 'sa.csv'), 2)
    rows = open(tmp_path / 'sa.csv').read().strip().split('\n')
    assert [r.split(',')[0] for r in rows[1:]] == ['subj%02d' % i for i in range(12)]


_DYING_WORKER = '''
import os, sys, time
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
from test_host_pipeline import stub_forward
from ukbb_cardiac_amd import deploy_network as DN
rank = int(os.environ['UKBB_SHARD_INDEX'])
F, _ = DN.define_flags().parse(['--data_dir', sys.argv[1]])
assert (F.num_shards, F.shard_index, F.work_stealing) == (2, rank, True)
calls = [0]
def forward(batch):
    calls[0] += 1
    if rank == 0 and sys.argv[2] == 'die' and calls[0] == 4:
        os._exit(9)                                            # mid-subject (its second), claim file left behind, no cleanup of any kind
    if rank == 1:
        time.sleep(0.05)                                       # still busy with its own share when worker 0 dies
    return stub_forward(batch)
DN.run(F, forward, log=lambda *_: None)
'''


def test_a_worker_that_dies_mid_cohort_loses_nothing_but_its_exit_status(tmp_path, capfd):
    from test_host_pipeline import _write_subject
    from ukbb_cardiac_amd import nifti, shard
    data = tmp_path / 'data'
    data.mkdir()
    for i in range(8):
        _write_subject(data, 'subj%02d' % i, 'sa', (20, 28, 2, 3), 40 + i)
    script = tmp_path / 'w.py'
    script.write_text(_DYING_WORKER.format(root=ROOT))
    rc = shard.launch(2, [str(script), str(data), 'die'])
    assert rc == 9 and 'shard 0 of 2' in capfd.readouterr().err                      # the launcher reports the dead worker ...
    done = [i for i in range(8) if os.path.exists(data / ('subj%02d' % i) / 'seg_sa.nii.gz')]
    assert done == list(range(8)), done                                              # ... and the survivor drained the whole list, the orphan included
    for i in range(8):
        assert not [f for f in os.listdir(data / ('subj%02d' % i)) if f.startswith('.claim') or '.tmp.' in f]
        assert nifti.load(str(data / ('subj%02d' % i) / 'seg_sa.nii.gz')).data.shape == (20, 28, 2, 3)
    before = {i: os.path.getmtime(data / ('subj%02d' % i) / 'seg_sa.nii.gz') for i in range(8)}
    assert shard.launch(2, [str(script), str(data), 'ok']) == 0                       # the rerun is a no-op
    assert before == {i: os.path.getmtime(data / ('subj%02d' % i) / 'seg_sa.nii.gz') for i in range(8)}
