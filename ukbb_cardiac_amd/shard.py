"""Subject-granular sharding across the GPUs of one node.

The path has no exchange step: every (subject, frame, slice) is independent
(``common/deploy_network.py:58,103``), so multi-GPU is a batch split with no
collective.  Subject ``i`` of ``sorted(os.listdir(data_dir))`` goes to shard
``i mod num_shards``; the reference's skip-if-output-exists
(``deploy_network.py:62-67``) keeps reruns idempotent, so shards may also share
a directory with a crashed earlier run.

``python -m ukbb_cardiac_amd.shard --gpus 8 -- ukbb_cardiac_amd/deploy_network.py --seq_name sa ...``
starts one worker process per GPU (``HIP_VISIBLE_DEVICES=i``) and waits.
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


def launch(gpus: int, argv: Sequence[str]) -> int:
    procs = []
    for g in range(gpus):
        env = dict(os.environ)
        env['HIP_VISIBLE_DEVICES'] = str(g)
        env['UKBB_SHARD_INDEX'] = str(g)
        env['UKBB_NUM_SHARDS'] = str(gpus)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait())
    return rc


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if '--' not in argv or len(argv) < 3 or argv[0] != '--gpus':
        sys.exit('usage: python -m ukbb_cardiac_amd.shard --gpus N -- script.py [flags...]')
    gpus = int(argv[1])
    rest = argv[argv.index('--') + 1:]
    sys.exit(launch(gpus, rest))


if __name__ == '__main__':
    main()
