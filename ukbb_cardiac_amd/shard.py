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


# ---- work stealing on top of the static split: claim files ---------------------------------------------------------
# Static i mod G leaves the worker that drew the large subjects (or the slow disk, or a GPU shared with someone else) finishing
# last while the others idle.  With --work_stealing every worker walks the WHOLE sorted list -- its own share first, in order,
# then the other shares from their tails -- and takes a subject only after creating ``<subject>/.claim.<pre>_<seq>`` with
# O_CREAT | O_EXCL (atomic on local and NFSv3+ file systems): exactly one worker wins.  The reference's skip-if-output-exists
# (common/deploy_network.py:62-67) still decides what is done; the claim only says "someone is on it".  A claim whose writer is
# gone (worker killed mid-subject) is swept like nifti.py's tmp files: same host + pid namespace and the pid is dead -> removed at
# once; another host -> removed after CLAIM_MAX_AGE_S.  No claim is ever waited for.

CLAIM_MAX_AGE_S = 6 * 3600.0


def claim_path(subject_dir: str, what: str) -> str:
    return os.path.join(subject_dir, '.claim.' + what)


def _claim_is_stale(path: str) -> bool:
    import time
    from .nifti import _host_tag
    try:
        with open(path) as f:
            parts = f.read().split()
        age = time.time() - os.path.getmtime(path)
    except OSError:
        return False                                        # vanished (its owner finished) or unreadable: not ours to judge
    if len(parts) < 2 or not parts[1].isdigit():
        return age > 60.0                                   # a claim is written in one go; an empty one this old lost its writer
    tag, pid = parts[0], int(parts[1])
    if tag != _host_tag():
        return age > CLAIM_MAX_AGE_S
    if pid == os.getpid():
        return False
    try:
        os.kill(pid, 0)
        return False
    except ProcessLookupError:
        return True
    except OSError:
        return False                                        # exists but is not ours to signal


def try_claim(subject_dir: str, what: str) -> bool:
    """Atomically take ``what`` (e.g. 'seg_sa') of a subject.  True: this process owns it until ``release_claim``."""
    from .nifti import _host_tag
    path = claim_path(subject_dir, what)
    for attempt in range(2):
        try:
            fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o644)
        except FileExistsError:
            if attempt == 0 and _claim_is_stale(path):
                try:
                    os.remove(path)                         # several sweepers may race here: one remove wins, O_EXCL arbitrates the re-claim
                except OSError:
                    pass
                continue
            return False
        except OSError:
            return False                                    # read-only / vanished directory: leave the subject to whoever can write there
        with os.fdopen(fd, 'w') as f:
            f.write('%s %d\n' % (_host_tag(), os.getpid()))
        return True
    return False


def release_claim(subject_dir: str, what: str) -> None:
    try:
        os.remove(claim_path(subject_dir, what))
    except OSError:
        pass


def stealing_order(subjects: Sequence[str], shard_index: int, num_shards: int) -> List[str]:
    """This worker's walk over the whole list: its static share in order, then the other workers' shares -- nearest shard first,
    each from its TAIL (its owner works from the head, so thief and owner meet as late as possible)."""
    out = subjects_for_shard(subjects, shard_index, num_shards)
    for k in range(1, num_shards):
        out += reversed(subjects_for_shard(subjects, (shard_index + k) % num_shards, num_shards))
    return out


class ClaimQueue:
    """Iterates the subjects this worker should look at; ``take(name)`` / ``done(name)`` bracket the work on one.

    stealing=False: the static share, ``take`` always succeeds (the r01-r05 behaviour).  stealing=True: ``stealing_order``, ``take``
    = ``try_claim``; subjects found claimed by a live worker are remembered and offered once more at the end (``second_chance``):
    by then their owner has either finished them (output exists -> the caller's skip-if-exists drops them) or died (stale claim ->
    taken over)."""

    def __init__(self, data_dir, subjects, shard_index, num_shards, what, stealing):
        self.data_dir, self.what, self.stealing = data_dir, what, bool(stealing) and num_shards > 1
        self.static = subjects_for_shard(subjects, shard_index, num_shards)
        self.order = stealing_order(subjects, shard_index, num_shards) if self.stealing else list(self.static)
        self.held, self.busy_elsewhere, self.stolen = set(), [], []
        self._static_set = set(self.static)

    def __iter__(self):
        return iter(self.order)

    def take(self, name):
        if not self.stealing:
            return True
        if try_claim(os.path.join(self.data_dir, name), self.what):
            self.held.add(name)
            if name not in self._static_set:
                self.stolen.append(name)
            return True
        self.busy_elsewhere.append(name)
        return False

    def done(self, name):
        if name in self.held:
            self.held.discard(name)
            release_claim(os.path.join(self.data_dir, name), self.what)

    def second_chance(self):
        again, self.busy_elsewhere = self.busy_elsewhere, []
        return again

    def release_all(self):
        for name in list(self.held):
            self.done(name)


def split_cpus(cpus: Sequence[int], n: int) -> List[List[int]]:
    """n contiguous, near-equal parts of a CPU list (fewer CPUs than workers: everyone gets the whole list)."""
    cpus = sorted(cpus)
    if n < 1 or len(cpus) < n:
        return [list(cpus) for _ in range(max(n, 1))]
    q, r = divmod(len(cpus), n)
    out, at = [], 0
    for i in range(n):
        k = q + (1 if i < r else 0)
        out.append(cpus[at:at + k])
        at += k
    return out


def cgroup_cpu_limit():
    """CPU bandwidth the cgroup allows (cpu.max / cfs_quota), in CPUs, or None."""
    try:
        a, b = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        return None if a == 'max' else float(a) / float(b)
    except Exception:
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        p = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return q / p if q > 0 and p > 0 else None
    except Exception:
        return None


def io_threads_for(cpus_in_set: int, n_workers: int) -> int:
    """Default --io_threads of one worker: what it can really run at once -- its CPU set, or its 1/n share of the cgroup quota when
    that is smaller -- between 1 and 16 (a short-axis subject costs ~0.24 s of a core in inflate + deflate against 10 ms of GPU:
    threads beyond the CPUs only add contention)."""
    lim = cgroup_cpu_limit()
    share = cpus_in_set if lim is None else min(cpus_in_set, lim / max(1, n_workers))
    return int(max(1, min(16, share)))


def apply_cpu_set_from_env() -> List[int]:
    """Worker side, before the first GPU call: bind this process (and the threads it will start) to UKBB_CPU_SET ("3,4,5,...").
    Returns the set applied ([] = none asked for / not supported)."""
    txt = os.environ.get('UKBB_CPU_SET', '')
    if not txt or not hasattr(os, 'sched_setaffinity'):
        return []
    cpus = [int(c) for c in txt.split(',') if c.strip().isdigit()]
    try:
        os.sched_setaffinity(0, cpus)
        return cpus
    except OSError:
        return []


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
    # per-worker CPU set = a contiguous 1/n of what this process may run on (the workers' reader / writer threads inherit it), and the
    # default --io_threads that fits it; the worker applies the set itself before its first GPU call (apply_cpu_set_from_env)
    try:
        cpu_sets = split_cpus(sorted(os.sched_getaffinity(0)), n)
    except AttributeError:
        cpu_sets = [[] for _ in range(n)]
    for i in range(n):
        env = dict(os.environ)
        env['HIP_VISIBLE_DEVICES'] = str(i // shards_per_gpu)
        env['UKBB_SHARD_INDEX'] = str(i)
        env['UKBB_NUM_SHARDS'] = str(n)
        if cpu_sets[i] and 'UKBB_CPU_SET' not in os.environ:
            env['UKBB_CPU_SET'] = ','.join(map(str, cpu_sets[i]))
            env.setdefault('UKBB_IO_THREADS', str(io_threads_for(len(cpu_sets[i]), n)))
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
