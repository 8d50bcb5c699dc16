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
