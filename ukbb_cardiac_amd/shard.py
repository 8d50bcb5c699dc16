"""Subject-granular sharding across the GPUs of one node.

The path has no exchange step: every (subject, frame, slice) is independent
(``common/deploy_network.py:58,103``), so multi-GPU is a batch split with no
collective.  Subject ``i`` of ``sorted(os.listdir(data_dir))`` goes to shard
``i mod num_shards``; the reference's skip-if-output-exists
(``deploy_network.py:62-67``) keeps reruns idempotent, so shards may also share
a directory with a crashed earlier run.

``python -m ukbb_cardiac_amd.shard --gpus 8 -- ukbb_cardiac_amd/deploy_network.py --seq_name sa ...``
starts one worker process per GPU (``HIP_VISIBLE_DEVICES=i``) and waits; it exits non-zero if any worker
failed or was killed.  Under ``torch.distributed.run`` the deploy scripts shard by RANK / WORLD_SIZE and bind
to GPU LOCAL_RANK instead (``default_device``).
"""
import os
import subprocess
import sys
from typing import List, Sequence


def shard_of(index: int, num_shards: int) -> int:
    return index % num_shards


def subjects_for_shard(subjects: Sequence[str], shard_index: int, num_shards: int) -> List[str]:
    if num_shards < 1 or not 0 <= shard_index < num_shards:
        raise ValueError('bad shard %d of %d' % (shard_index, num_shards))
    return [s for i, s in enumerate(subjects) if shard_of(i, num_shards) == shard_index]


def shard_from_env(default_index=0, default_count=1):
    """torchrun-style environment (RANK / WORLD_SIZE) or UKBB_SHARD_INDEX / UKBB_NUM_SHARDS."""
    idx = os.environ.get('UKBB_SHARD_INDEX', os.environ.get('RANK'))
    cnt = os.environ.get('UKBB_NUM_SHARDS', os.environ.get('WORLD_SIZE'))
    return (int(idx) if idx is not None else default_index, int(cnt) if cnt is not None else default_count)


def exit_status(returncode: int) -> int:
    """Shell-style status of a worker: a worker killed by signal n has returncode -n (a GPU fault aborts with
    -6 / -11) and must count as a failure, 128 + n, never as 'smaller than 0 = fine'."""
    return returncode if returncode >= 0 else 128 - returncode


def default_device() -> int:
    """HIP ordinal a worker binds to when --device is not given: torchrun's LOCAL_RANK (all ranks of a node see
    all GPUs), else 0 (shard.launch narrows each worker to one GPU with HIP_VISIBLE_DEVICES)."""
    if 'UKBB_SHARD_INDEX' in os.environ:
        return 0
    return int(os.environ.get('LOCAL_RANK', 0))


def launch(gpus: int, argv: Sequence[str], shards_per_gpu: int = 1) -> int:
    """One worker process per shard, ``shards_per_gpu`` consecutive shards pinned to each GPU
    (``HIP_VISIBLE_DEVICES``; 1 in production, > 1 to oversubscribe a device in tests).  Returns 0 only if every
    worker exited 0; otherwise the first failing worker's status, after naming every shard that failed.  Nothing is
    restarted in place: rerun the same command and skip-if-output-exists resumes the missing subjects."""
    n = gpus * shards_per_gpu
    procs = []
    for i in range(n):
        env = dict(os.environ)
        env['HIP_VISIBLE_DEVICES'] = str(i // shards_per_gpu)
        env['UKBB_SHARD_INDEX'] = str(i)
        env['UKBB_NUM_SHARDS'] = str(n)
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):         # the UKBB_* pair above is authoritative for the workers
            env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env))
    rc = 0
    for i, p in enumerate(procs):
        code = exit_status(p.wait())
        if code:
            print('shard %d of %d (GPU %d) failed with status %d%s: its subjects are incomplete -- rerun to resume'
                  % (i, n, i // shards_per_gpu, code, ' (killed by signal %d)' % (code - 128) if code > 128 else ''),
                  file=sys.stderr)
            rc = rc or code
    csv_path = _flag_value(argv, 'output_csv')
    if csv_path and rc == 0 and n > 1:                          # the workers wrote <path>.shard<i>-of-<n>: one table, sorted subjects
        from .measures import merge_shard_csv
        merge_shard_csv(csv_path, n)
    return rc


def _flag_value(argv, name):
    """Value of --name VALUE / --name=VALUE in a worker command line, or None."""
    for i, a in enumerate(argv):
        if a in ('--' + name, '-' + name) and i + 1 < len(argv):
            return argv[i + 1]
        for pre in ('--' + name + '=', '-' + name + '='):
            if a.startswith(pre):
                return a[len(pre):]
    return None


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    usage = 'usage: python -m ukbb_cardiac_amd.shard --gpus N [--shards_per_gpu M] -- script.py [flags...]'
    if '--' not in argv or len(argv) < 3 or argv[0] != '--gpus':
        sys.exit(usage)
    head, rest = argv[:argv.index('--')], argv[argv.index('--') + 1:]
    gpus, per = int(head[1]), 1
    if len(head) == 4 and head[2] == '--shards_per_gpu':
        per = int(head[3])
    elif len(head) != 2:
        sys.exit(usage)
    if gpus < 1 or per < 1 or not rest:
        sys.exit(usage)
    sys.exit(launch(gpus, rest, per))


if __name__ == '__main__':
    main()
