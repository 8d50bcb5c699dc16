#!/usr/bin/env python3
"""Headline benchmark: short-axis FCN inference, batch = 64 synthetic 192x208
slices per GPU, fp32 (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one forward of the hot path over one batch of 64 slices already
resident in HBM, label map (int32) left in HBM.  One process per GPU, batch
split only, no data-path collective (weak scaling); the process group is used
for the barrier and the max-over-ranks time only.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
BATCH, H, W = 64, 192, 208


def kernel_source_sha():
    """Fingerprint of the kernel + engine sources: a committed PMC traffic file is only quoted when it was
    collected from the same sources (tools/pmc_traffic.py stamps it)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'ukbb_cardiac_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h', '.cpp')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(path, kernel, batch):
    """HBM bytes per launch of `kernel` from a rocprofv3 --pmc summary (tools/pmc_traffic.py), or (None, why).
    A file is quoted only if it says it was collected from these very kernel sources at this batch size."""
    import glob
    cands = [path] if path else sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), reverse=True)
    sha = kernel_source_sha()
    for c in cands:
        try:
            tj = json.load(open(c))
        except Exception:
            continue
        if tj.get('kernel_source_sha') != sha or tj.get('batch', BATCH) != batch:
            continue
        hit = tj.get('engine_kernels', {}).get(kernel)
        if hit:
            return hit['hbm_bytes'], '%s (%s, separate rocprofv3 --pmc passes of this build, kernel_source_sha %s)' % (
                os.path.relpath(c, ROOT), hit.get('gpu_kernel', kernel), sha)
    return None, 'no PMC traffic file matches kernel_source_sha %s (collect with tools/run_pmc.sh)' % sha


def _timed(fn, slices_per_call, target_seconds, max_calls):
    fn()                                                               # warm-up: thread pool, allocator, page-in
    calls = 0
    t0 = time.perf_counter()
    while True:
        fn()
        calls += 1
        dt = time.perf_counter() - t0
        if dt >= target_seconds or calls >= max_calls:
            return slices_per_call * calls / dt, slices_per_call * calls, dt


def cpu_quota():
    """What this process may actually use of the host's CPUs: the scheduler affinity mask and the cgroup CPU bandwidth limit
    (a GPU box can show 128-256 logical CPUs while the container is capped at a fraction of them)."""
    q = {'logical_cpus': os.cpu_count()}
    try:
        q['sched_affinity_cpus'] = len(os.sched_getaffinity(0))
    except Exception:
        q['sched_affinity_cpus'] = None
    q['cgroup_cpu_max'] = None
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
        except Exception:
            continue
        if path.endswith('cpu.max'):
            q['cgroup_cpu_max'] = ' '.join(txt)
            if txt and txt[0] != 'max' and len(txt) > 1 and float(txt[1]) > 0:
                q['cgroup_cpus'] = round(float(txt[0]) / float(txt[1]), 2)
        else:
            try:
                period = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                q['cgroup_cpu_max'] = '%s %d' % (txt[0], int(period))
                if float(txt[0]) > 0 and period > 0:
                    q['cgroup_cpus'] = round(float(txt[0]) / period, 2)
            except Exception:
                pass
        break
    return q


def smi_sclk_mhz():
    """Best effort: the shader clock rocm-smi reports right now (None when the tool or the permission is missing)."""
    import re
    import shutil
    import subprocess
    # Under a profiler (rocprofv3 preloads a library that initialises the GPU in every child) a `#!/usr/bin/env python3` tool is an exec
    # from a GPU-initialised process, which this pool refuses: no probe there, and everywhere else the tool is started with this
    # interpreter directly, in an environment without the preload.
    if any('rocprof' in os.environ.get(k, '').lower() for k in ('LD_PRELOAD', 'HSA_TOOLS_LIB', 'ROCP_TOOL_LIBRARIES')):
        return None
    tool = shutil.which('rocm-smi')
    if tool is None:
        return None
    env = {k: v for k, v in os.environ.items() if k not in ('LD_PRELOAD', 'HSA_TOOLS_LIB')}
    try:
        r = subprocess.run([sys.executable, tool, '--showclocks'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10, env=env)
        m = re.search(r'sclk clock level:?\s*\d*:?\s*\(?(\d+)\s*Mhz', r.stdout, flags=re.I)
        return int(m.group(1)) if m else None
    except Exception:
        return None


def smi_showuse():
    """`rocm-smi --showuse` as this container sees it (text), or why it could not be run.  Printed beside `sustained` so that a reader
    of the driver's own GPU-busy samples can tell an idle GPU from a tool that reports nothing inside the container."""
    import shutil
    import subprocess
    if any('rocprof' in os.environ.get(k, '').lower() for k in ('LD_PRELOAD', 'HSA_TOOLS_LIB', 'ROCP_TOOL_LIBRARIES')):
        return 'skipped under a profiler'
    tool = shutil.which('rocm-smi')
    if tool is None:
        return 'rocm-smi not on PATH'
    env = {k: v for k, v in os.environ.items() if k not in ('LD_PRELOAD', 'HSA_TOOLS_LIB')}
    try:
        r = subprocess.run([sys.executable, tool, '--showuse'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=15, env=env)
        lines = [l.strip() for l in r.stdout.splitlines() if 'use' in l.lower() or 'busy' in l.lower() or 'error' in l.lower() or 'warn' in l.lower()]
        return ' | '.join(lines)[:400] or r.stdout.strip()[-400:]
    except Exception as e:
        return repr(e)[-200:]


def sustained_probe(step, n, dev_index, seconds=10.0, est_ms=1.3):
    """Reported beside the headline: the SAME step looped for >= `seconds` of GPU time (a cohort runs for hours, the headline
    region is a 26 ms burst), with the shader clock the chip held DURING the loop: a one-wave probe (ukbb_fcn_clock_probe:
    s_memtime against the 100 MHz s_memrealtime) on a stream of its own while the steps are queued on the bench stream."""
    import torch
    from ukbb_cardiac_amd import _lib
    side = torch.cuda.Stream(torch.device('cuda', dev_index))
    chunk = max(8, int(0.25 / (est_ms * 1e-3)))                       # about a quarter second of steps per enqueue round
    clocks, smi = [], []
    done = 0
    # `rocm-smi --showuse` from a thread while the steps keep being queued: what a sampler sees of this loop from inside the container
    import threading
    use = {}

    def sample_use():
        time.sleep(min(2.0, seconds / 3))
        use['text'] = smi_showuse()
    th = threading.Thread(target=sample_use, daemon=True)
    th.start()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while True:
        for _ in range(chunk):
            step()
        done += chunk
        # the probe's wave runs next to the queued steps; its host-side wait returns after ~0.2 ms, the bench stream stays full
        try:
            clocks.append(_lib.clock_probe_mhz(dev_index, side.cuda_stream, 200))
        except Exception:
            pass
        if time.perf_counter() - t0 >= seconds:
            break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s = smi_sclk_mhz()                                                 # right behind the loop (the tool itself takes ~0.1-1 s)
    th.join(timeout=20)
    clocks.sort()
    med = clocks[len(clocks) // 2] if clocks else None
    return {'value': round(n * done / dt, 1), 'unit': 'slices/s', 'steps': done, 'seconds': round(dt, 3), 'ms_per_step': round(dt / done * 1e3, 4),
            'shader_clock_mhz_in_loop': None if med is None else round(med, 1),
            'shader_clock_samples_mhz': [round(c) for c in clocks],
            'rocm_smi_sclk_mhz_after_loop': s,
            'rocm_smi_showuse_during_loop': use.get('text'),
            'note': 'same step, same resident batch, looped for %.1f s; clock = median of one-wave s_memtime / s_memrealtime probes that ran '
                    'beside the queued steps' % dt}


def usable_cpus():
    """CPUs this process can really run on at once: min(scheduler affinity, cgroup bandwidth quota, logical CPUs)."""
    q = cpu_quota()
    c = [q['logical_cpus'] or 1]
    if q.get('sched_affinity_cpus'):
        c.append(q['sched_affinity_cpus'])
    if q.get('cgroup_cpus'):
        c.append(max(1, int(round(q['cgroup_cpus']))))
    return max(1, min(c))


def physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        return os.cpu_count() or 1


def cpu_leg(which, target_seconds=None):
    """Runs in a CHILD process of rank 0 (fresh thread pools, no GPU context; two OpenMP runtimes spinning in one
    process cost an order of magnitude on the 256-thread GPU hosts) and prints one JSON object."""
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import uniform_slices
    from ukbb_cardiac_amd.weights import pack_flat, synthetic_params
    if target_seconds is None:
        target_seconds = float(os.environ.get('UKBB_CPU_LEG_SECONDS', '6.0'))
    arch = MODELS['FCN_sa']
    params = synthetic_params(arch, 1234)
    img64 = uniform_slices(BATCH, H, W, seed=1)
    if which == 'c':
        # one thread count per child process (OMP_NUM_THREADS is read when the OpenMP runtime starts): cpu_baseline() sweeps
        from oracle import c_oracle
        flat = pack_flat(arch, params)
        img8 = img64[:8]
        rc, nc, tc = _timed(lambda: c_oracle.forward(arch, flat, img8, want_logits=False), 8, target_seconds, 400)
        print(json.dumps({'value': round(rc, 2), 'unit': 'slices/s', 'cores': c_oracle.num_threads(),
                          'sample': '%d slices (batches of 8) through oracle/fcn_oracle.c (plain C, OpenMP, unfused) on %d OpenMP threads, %.1f s'
                                    % (nc, c_oracle.num_threads(), tc)}))
        return
    import torch
    from oracle.torch_oracle import TorchFCN
    net = TorchFCN(params, arch)
    # thread count: all logical CPUs is what SURVEY.md 8(d) names, but on the SMT hosts of the GPU boxes oneDNN is
    # several times slower there than at the physical core count; take the fastest of a short ascending sweep
    logical, phys = os.cpu_count() or 1, physical_cores()
    usable = usable_cpus()
    cands = sorted({c for c in (max(1, usable // 2), usable, 2 * usable, 8, 16, 32, 64, phys // 2, phys, logical) if 1 <= c <= logical})
    # swept on the timed batch itself (r02 swept on 4 slices and picked a count that was not the fastest at N = 64)
    probe, best_t, best_n, sweep = img64, None, None, {}
    net(img64[:4])
    for c in cands:
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        net(probe)
        dt = time.perf_counter() - t0
        sweep[c] = round(BATCH / dt, 2)
        if best_t is None or dt < best_t:
            best_t, best_n = dt, c
        elif dt > 1.5 * best_t:
            break
    torch.set_num_threads(best_n)
    img10 = img64[:10]
    r64, n64, t64 = _timed(lambda: net(img64), BATCH, target_seconds, 50)
    r10, n10, t10 = _timed(lambda: net(img10), 10, target_seconds, 50)          # 50 calls = one 500-slice subject
    print(json.dumps({
        'value': round(r64, 2), 'unit': 'slices/s', 'cores': best_n, 'kind': 'port',
        'sample': '%d slices of %dx%d in batches of %d through oracle/torch_oracle.py TorchFCN (torch %s CPU, fp32, %d threads = '
                  'fastest of the sweep %s on a host with %d logical / %d physical cores; op-by-op restatement of '
                  'common/network.py build_FCN standing in for the reference\'s TF-CPU graph, TensorFlow is not installable '
                  'here), %.1f s' % (n64, H, W, BATCH, torch.__version__, best_n, json.dumps(sweep), logical, phys, t64),
        'reference_call_pattern': {'value': round(r10, 2), 'unit': 'slices/s',
                                   'sample': '%d sess.run-shaped calls of N=10 slices (deploy_network.py:103-111: one call per '
                                             'frame, 50 per subject), same torch-CPU graph and threads, %.1f s' % (n10 // 10, t10)}}))


def cpu_baseline():
    """The CPU leg SURVEY.md 8(d) specifies, on this host's cores, on a bounded sample of the same workload:
    the torch-CPU restatement of the reference graph (oracle/torch_oracle.py: conv / batch_norm / relu op by op,
    fp32) -- the stand-in for the reference's TF-CPU deploy_network.py, which cannot be installed here -- timed (a) at
    the bench batch N = 64 and (b) in the reference's own call pattern, one sess.run per frame with N = 10 slices
    (deploy_network.py:103-111); plus (c) the plain-C OpenMP port (oracle/fcn_oracle.c) as a second figure.  Each leg is
    a child process of rank 0 (never an exec of this GPU-initialised process)."""
    import subprocess

    def child(which, env_extra):
        env = dict(os.environ)
        env.update(env_extra)
        env['HIP_VISIBLE_DEVICES'] = ''                                 # the CPU legs must not touch the GPU
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-leg', which], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=600)
        if r.returncode != 0:
            return {'error': (r.stderr or r.stdout)[-400:]}
        return json.loads(r.stdout.strip().splitlines()[-1])
    torch_leg = child('torch', {})
    # C/OpenMP port: thread counts around what the container may use (quota / 2, quota, 2 x quota), one short child each, then the
    # fastest count timed for the full sample; `cores` is the thread count that leg really ran with (r05 reported the 128 threads of
    # a 16-CPU quota as "cores")
    usable = usable_cpus()
    sweep = {}
    for c in sorted({max(1, usable // 2), usable, min(2 * usable, os.cpu_count() or usable)}):
        leg = child('c', {'OMP_NUM_THREADS': str(c), 'OMP_DYNAMIC': 'FALSE', 'UKBB_CPU_LEG_SECONDS': '2.0'})
        sweep[c] = leg.get('value', leg.get('error'))
    ok_counts = [c for c, v in sweep.items() if isinstance(v, (int, float))]
    best_c = max(ok_counts, key=lambda c: sweep[c]) if ok_counts else usable
    c_leg = child('c', {'OMP_NUM_THREADS': str(best_c), 'OMP_DYNAMIC': 'FALSE'})
    c_leg['thread_sweep_slices_per_s'] = sweep
    c_leg['usable_cpus'] = usable
    # the headline CPU figure is the FASTEST leg measured (a slow baseline would flatter any GPU/CPU ratio); all legs are kept
    legs = {'torch_cpu': torch_leg, 'c_port': c_leg}
    ok = {k: v for k, v in legs.items() if 'value' in v}
    if not ok:
        return {'error': 'no CPU leg finished', 'legs': legs}
    name = max(ok, key=lambda k: ok[k]['value'])
    best = ok[name]
    return {'value': best['value'], 'unit': best['unit'], 'cores': best['cores'], 'kind': 'port',
            'sample': 'fastest of the CPU legs (%s): %s' % (name, best['sample']), 'legs': legs,
            # what the host let this job use: a weak CPU figure on a 128-core box is usually a container quota
            'cpu_quota': cpu_quota()}


def inflight_probe(arch, params, x, n, steps):
    """Reported beside the headline, never as it: the same K steps with TWO batches in flight (two engines = two
    activation workspaces, two HIP streams, steps alternating).  Each of the ~14 launches of a step pays ~10 us of fixed
    time (dispatch, per-XCD L2 write-back between dependent kernels, first-load latency, tail; profiles/r02_notes.md);
    a second stream fills those holes.  The headline stays single-stream so that its per-kernel HIP-event / rocprof
    durations are those of a kernel that owns the GPU."""
    import torch
    from ukbb_cardiac_amd.engine import Engine
    engs = [Engine(arch, params, device=x.device.index) for _ in range(2)]
    streams = [torch.cuda.Stream(x.device) for _ in range(2)]
    preds = [torch.empty((n, H, W), dtype=torch.int32, device=x.device) for _ in range(2)]
    for e in engs:
        e.reserve(n, H, W)

    def step(i):
        engs[i & 1].run_device(x.data_ptr(), n, H, W, pred_ptr=preds[i & 1].data_ptr(), stream=streams[i & 1].cuda_stream)
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e in engs:
        e.close()
    return {'value': round(n * steps / dt, 1), 'unit': 'slices/s', 'ms_per_step': round(dt / steps * 1e3, 4),
            'note': 'same workload, two batches in flight on two HIP streams (two workspaces); not the headline'}


def other_configs_probe(device):
    """Reported beside the headline, never as it: BASELINE.json configs[4] (aortic U-Net, N = 100 x 256x256, fp32 and the bf16
    MFMA path with per-class Dice against fp32, common/image_utils.py:171-175) and the reference's own call shape N = 10
    (deploy_network.py:103-111) of the headline model -- each a few forwards, inputs resident in HBM."""
    import numpy as np
    import torch
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.image_utils import np_categorical_dice
    from ukbb_cardiac_amd.phantom import cine_phantom, uniform_slices
    from ukbb_cardiac_amd.weights import synthetic_params

    def rate(eng, x, n, h, w, pred, k=10):
        for _ in range(3):
            eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            eng.run_device(x.data_ptr(), n, h, w, pred_ptr=pred.data_ptr())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k
    out = {}
    arch = MODELS['UNet_ao']
    n = 100
    img = ((cine_phantom(n, 256, 256, seed=5) - 0.3) / 0.25).astype(np.float32)
    x = torch.from_numpy(img).to(device)
    pred = torch.empty((n, 256, 256), dtype=torch.int32, device=device)
    with Engine(arch, synthetic_params(arch, 1234), device=device.index) as eng:
        # timed regions of >= 50 ms: one host synchronisation costs 0.3-0.5 ms, which a 10-forward region of the bf16 path (10 ms) showed as +4 % (r06: 1.10 vs 1.05 ms)
        t32 = rate(eng, x, n, 256, 256, pred, k=20)
        p32 = pred.cpu().numpy().copy()
        eng.set_precision('bf16')
        t16 = rate(eng, x, n, 256, 256, pred, k=50)
        p16 = pred.cpu().numpy()
        launches = len(eng.kernel_names())
    from ukbb_cardiac_amd.arch import fcn_macs_per_slice
    m3u, m1u = fcn_macs_per_slice(arch, 256, 256)
    flop5 = 2.0 * (m3u + m1u) * n
    # algorithmic HBM bytes of the bf16 plan (tools/unet_roofline.py: every stored bf16 map written once and read once per consumer, image
    # in, labels out; no halos, no weights); measured traffic only from a counter file collected from these very sources
    lv = [256 * 256 * 16, 128 * 128 * 32, 64 * 64 * 64, 32 * 32 * 128, 16 * 16 * 256]
    alg5 = 256 * 256 * 4 * 2 + lv[0] * 2 * 3 + sum(lv[l] * 2 * 2 + lv[l] * 2 * (3 if l < 4 else 2) for l in range(1, 5)) + \
        sum(lv[l] * 2 * 2 * (3 if l > 0 else 1) for l in range(3, -1, -1))
    alg5 *= n
    tr5, tr5_src = None, 'no counter file of this build (tools/profile_unet.sh)'
    try:
        import glob
        for c in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_unet_traffic.json')), reverse=True):
            tj = json.load(open(c))
            if tj.get('kernel_source_sha') == kernel_source_sha():
                tr5, tr5_src = tj['hbm_bytes_per_forward'], os.path.relpath(c, ROOT)
                break
    except Exception:
        pass
    out['config5_aortic_unet'] = {
        'roofline': {'mfma': {'dtype': 'bf16', 'achieved': round(flop5 / t16 / 1e12, 1), 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(flop5 / t16 / 2.5e15, 4),
                              'flop_per_forward': flop5, 'note': 'algorithmic FLOPs of the 100-slice forward / wall time of the bf16 forward'},
                     'hbm': {'achieved': round(alg5 / t16 / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(alg5 / t16 / 8e12, 4),
                             'algorithmic_bytes_per_forward': alg5, 'traffic': tr5, 'traffic_source': tr5_src,
                             'frac_measured_traffic': None if tr5 is None else round(tr5 / t16 / 8e12, 4)},
                     'bound': 'neither roof binds: per-kernel table in profiles/'},
        'workload': 'UNet_ao (network_ao.py:18-64), N = 100 x 256x256 resident in HBM, int32 labels out',
        'fp32': {'value': round(n / t32, 1), 'unit': 'slices/s', 'ms_per_step': round(t32 * 1e3, 4)},
        'bf16': {'value': round(n / t16, 1), 'unit': 'slices/s', 'ms_per_step': round(t16 * 1e3, 4), 'dtype': 'bf16 MFMA operands, fp32 accumulation, '
                 'bf16 activations in HBM', 'kernel_launches': launches,
                 'dice_vs_fp32': {'class1': round(float(np_categorical_dice(p16, p32, 1)), 4), 'class2': round(float(np_categorical_dice(p16, p32, 2)), 4)},
                 'label_disagreement': round(float((p16 != p32).mean()), 5)}}
    # the reference's DEFAULT aortic model (demo_pipeline.py:116-117): one slice position, 100 frames, circular 9-frame windows
    arch = MODELS['UNet-LSTM_ao']
    F = 100
    prob = torch.empty((F, 256, 256, 3), dtype=torch.float32, device=device)
    with Engine(arch, synthetic_params(arch, 1234), device=device.index) as eng:
        def cine(k):
            for _ in range(k):
                eng.run_cine_device(x.data_ptr(), F, 256, 256, prob.data_ptr(), pred.data_ptr())
            torch.cuda.synchronize()
        cine(2)
        t0 = time.perf_counter()
        cine(5)
        tl = (time.perf_counter() - t0) / 5
        pl32 = pred.cpu().numpy().copy()
        # UKBB_PREC_BF16 on the same handle: bf16-storage U-Net plan, ConvLSTM as direct convs on the bf16 matrix instruction (cell state fp32)
        eng.set_precision('bf16')
        cine(2)
        t0 = time.perf_counter()
        cine(5)
        tl16 = (time.perf_counter() - t0) / 5
        pl16 = pred.cpu().numpy()
        # algorithmic HBM bytes of the bf16 ConvLSTM part per pixel and cine: 16 steps x (gx 64 x 2 + h in 16 x 2 + c in / out 2 x 16 x 4 + h out 16 x 2), the x pass
        # (x 16 x 2 in, gx 128 x 2 + 2 x (c1 16 x 4 + h1 16 x 2) out), the output pass (18 hidden maps x 16 x 2 in, prob + labels 16 out)
        bytes16 = (16 * 320 + 480 + 592) * float(F * 256 * 256)
        lstm_bf16 = {'value': round(F / tl16, 1), 'unit': 'frames/s', 'ms_per_cine': round(tl16 * 1e3, 3),
                     'roofline': {'bound': 'hbm', 'peak': 8000.0, 'unit': 'GB/s', 'achieved': round(bytes16 / tl16 / 1e9, 1), 'frac': round(bytes16 / tl16 / 8e12, 4),
                                  'algorithmic_bytes_lstm_part': bytes16, 'traffic': None,
                                  'note': 'ConvLSTM part only over the whole cine time (the bf16 U-Net features take ~1.3 ms of it); counters: profiles/r*_unet_lstm_bf16*'},
                     'dice_vs_fp32': {'class1': round(float(np_categorical_dice(pl16, pl32, 1)), 4), 'class2': round(float(np_categorical_dice(pl16, pl32, 2)), 4)},
                     'label_disagreement': round(float((pl16 != pl32).mean()), 5)}
    # reference-graph FLOPs with the features computed once per frame (the reference recomputes the U-Net for each of the 9 window positions):
    # U-Net + per window 18 gate convs (3x3, 32 -> 64) + 9 output convs -- an "effective" figure, not a utilisation (the x half of the gate
    # conv is hoisted out of the time loop and the rest runs as Winograd F(2x4))
    hw = 256 * 256
    flop_l = F * (2.0 * (m3u + m1u) + 2 * 9 * hw * (9 * 32 * 64) * 2.0 + 9 * hw * 32 * 3 * 2.0)
    # algorithmic HBM bytes of the ConvLSTM part as this implementation lays it out (fp32): per time step gx 64 ch + h in 16 + c in / out 16 + 16 +
    # h out 16 (16 steps), the x pass (x in 16, gx out 2 x 64, c1 / h1 out 2 x 32), the output pass (18 hidden maps in, prob + labels out)
    px = F * hw * 4.0
    bytes_l = 16 * (64 + 16 + 32 + 16) * px + (16 + 128 + 64) * px + (18 * 16 + 4) * px
    def lstm_traffic(tag):
        # measured HBM bytes per cine of the ConvLSTM kernels from a counter file of these very sources (tools/profile_lstm.sh), or (None, why)
        import glob
        for c in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_unet_lstm%s_traffic.json' % tag)), reverse=True):
            try:
                tj = json.load(open(c))
                if tj.get('kernel_source_sha') == kernel_source_sha():
                    return tj['hbm_bytes_per_cine_lstm_kernels'], os.path.relpath(c, ROOT)
            except Exception:
                pass
        return None, 'no counter file of this build (tools/profile_lstm.sh)'
    tr32, tr32_src = lstm_traffic('')
    tr16, tr16_src = lstm_traffic('_bf16')
    lstm_bf16['roofline']['traffic'] = tr16
    lstm_bf16['roofline']['traffic_source'] = tr16_src
    out['unet_lstm_cine'] = {
        'workload': 'UNet-LSTM_ao (network_ao.py:255-399 + the window tiling of deploy_network_ao.py:129-183): one slice position, 100 frames '
                    'of 256x256 resident in HBM, fp32; prob [F,H,W,3] + int32 labels out',
        'value': round(F / tl, 1), 'unit': 'frames/s', 'ms_per_cine': round(tl * 1e3, 3), 'bf16': lstm_bf16,
        'roofline': {'bound': 'hbm', 'peak': 8000.0, 'unit': 'GB/s',
                     'achieved': round(bytes_l / tl / 1e9, 1), 'frac': round(bytes_l / tl / 8e12, 4),
                     'algorithmic_bytes_lstm_part': bytes_l, 'traffic': tr32, 'traffic_source': tr32_src,
                     'note': 'ConvLSTM part only (the U-Net features of the 100 frames add ~4 ms of MFMA-bound work to the same wall time, so this '
                             'is a lower bound of the rate the LSTM kernels reach); per-kernel times and counter traffic: profiles/r*_unet_lstm_*',
                     'effective_tflops_reference_graph_features_once': round(flop_l / tl / 1e12, 1)}}
    arch = MODELS['FCN_sa']
    x10 = torch.from_numpy(uniform_slices(10, H, W, seed=1)).to(device)
    pred10 = torch.empty((10, H, W), dtype=torch.int32, device=device)
    with Engine(arch, synthetic_params(arch, 1234), device=device.index) as eng:
        t10 = rate(eng, x10, 10, H, W, pred10, k=30)
    out['reference_call_shape_n10'] = {'workload': 'FCN_sa, N = 10 x 192x208 (one sess.run of the reference), own handle', 'value': round(10 / t10, 1),
                                       'unit': 'slices/s', 'ms_per_step': round(t10 * 1e3, 4)}
    return out


def config4_cohort_probe(dev_index, rank, world, n_subjects, barrier=None):
    """BASELINE.json configs[3]: `n_subjects` synthetic 192x208x10x50 subjects generated on the device from
    seed = subject id, each through the real device stages of common/deploy_network.py:83-131 (exact percentiles, clip / rescale /
    pad / transpose, forward in 128-slice batches, label unpack + per-frame counts, uint8 labels to pinned host memory; the ES frame
    is picked from the counts) -- ukbb_cardiac_amd/synthetic_cohort.py.  Subject i runs on rank i mod G, no collective: `barrier`
    (reached by every rank whatever happened before it) only lines the ranks' clocks up.  Returns this rank's record."""
    import torch
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.synthetic_cohort import SHAPE, run_cohort, stage_times
    from ukbb_cardiac_amd.weights import synthetic_params
    arch = MODELS['FCN_sa']
    dev = torch.device('cuda', dev_index)
    eng, err = None, None
    try:
        eng = Engine(arch, synthetic_params(arch, 1234), device=dev_index)
        warm = run_cohort(eng, range(4), rank=0, world=1)                    # allocations, plans, first touch of the pinned buffers
        del warm
        torch.cuda.synchronize(dev)
    except Exception as e:
        err = repr(e)[-300:]
    if barrier:
        barrier()
    if err:
        return {'error': err}
    X, Y, Z, T = SHAPE
    try:
        free_before = torch.cuda.mem_get_info(dev)[0]
        reserved_before = torch.cuda.memory_reserved(dev)
        t0 = time.perf_counter()
        rec = run_cohort(eng, range(n_subjects), rank=rank, world=world)
        dt = time.perf_counter() - t0                                        # from the common start to THIS rank's last result
        del rec['pipeline']
        torch.cuda.synchronize(dev)
        free_after = torch.cuda.mem_get_info(dev)[0]
        stages = stage_times(eng) if rank == 0 else None
    except Exception as e:
        return {'error': repr(e)[-300:]}
    finally:
        eng.close()
    out = {'workload': '%d synthetic subjects of %dx%dx%dx%d (%d slices each) generated on the device from seed = subject id '
                       '(ukbb_fcn_synth_volume), subject i on rank i mod %d; per subject: select_kth (exact 1 / 99 percentiles) -> rescale_pack -> '
                       'FCN_sa forward in 128-slice batches -> unpack_labels + per-frame class counts -> uint8 label volume to pinned host memory, '
                       'ES frame from the counts; 2 subjects in flight on 3 streams (SubjectPipeline); host gzip / file I/O excluded'
                       % (n_subjects, X, Y, Z, T, Z * T, world),
           'subjects_this_rank': rec['subjects'], 'seconds_this_rank': round(dt, 4),
           'device_free_bytes_before': int(free_before), 'device_free_bytes_after': int(free_after),
           'device_memory_delta_mb': round((free_before - free_after) / 1e6, 2),
           # the pipeline's device buffers are torch tensors: freed ones stay in torch's caching allocator, so the HIP-level free memory is
           # compared with the pipeline alive on both sides and torch's reserved pool is reported beside it
           'torch_reserved_bytes_before': int(reserved_before), 'torch_reserved_bytes_after': int(torch.cuda.memory_reserved(dev))}
    if stages:
        tot = sum(stages.values())
        out['stage_ms_one_subject_unoverlapped'] = stages
        out['network_fraction_of_device_time'] = round(stages['network'] / tot, 4)
        out['network_only_slices_per_s'] = round(Z * T / (stages['network'] * 1e-3), 1)
    return out


def drop_in_hipsession_probe(dev_index):
    """The literal drop-in surface: INTEGRATION.md's `HipSession` stub (executed verbatim from the file, as
    tests/test_deploy_gpu.py::test_integration_md_hipsession_snippet does), called the way common/deploy_network.py:103-116 calls
    its session -- 50 x sess.run(['prob:0', 'pred:0'], {'image:0': pageable float32 [10,192,208,1]}) per subject -- through
    ukbb_fcn_forward_host (H2D + forward + D2H of prob AND pred, synchronous)."""
    import re
    import tempfile
    import numpy as np
    from ukbb_cardiac_amd import _lib
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import uniform_slices
    from ukbb_cardiac_amd.weights import save_blob, synthetic_params
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    stub = [b for b in re.findall(r'```python\n(.*?)```', text, re.S) if 'class HipSession' in b]
    ns = {}
    exec(compile(stub[0], 'INTEGRATION.md', 'exec'), ns)
    arch = MODELS['FCN_sa']
    with tempfile.TemporaryDirectory() as td:
        mp = os.path.join(td, 'FCN_sa')
        save_blob(mp + '.ukbbw', arch, synthetic_params(arch, 1234))
        vol = uniform_slices(500, H, W, seed=1).reshape(50, 10, H, W, 1)                # one subject: 50 frames x 10 slices
        with ns['HipSession'](mp, lib=_lib.LIB_PATH, device=dev_index) as sess:
            for t in range(3):
                sess.run(['prob:0', 'pred:0'], feed_dict={'image:0': vol[t], 'training:0': False})
            t0 = time.perf_counter()
            for t in range(50):
                prob, pred = sess.run(['prob:0', 'pred:0'], feed_dict={'image:0': vol[t], 'training:0': False})
            dt = time.perf_counter() - t0
            t0 = time.perf_counter()
            for t in range(50):
                pred = sess.run('pred:0', feed_dict={'image:0': vol[t], 'training:0': False})
            dt_stub_pred = time.perf_counter() - t0
    return {'workload': "INTEGRATION.md HipSession.run(['prob:0','pred:0']) 50 x N = 10 x 192x208 from pageable numpy (deploy_network.py:103-116 "
                        "call pattern), ukbb_fcn_forward_host: H2D + forward + D2H of prob (6.4 MB) and pred (1.6 MB) per call, synchronous",
            'value': round(500 / dt, 1), 'unit': 'slices/s', 'ms_per_call': round(dt / 50 * 1e3, 3),
            'stub_asked_for_pred_only': {'value': round(500 / dt_stub_pred, 1), 'unit': 'slices/s',
                                         'note': 'same stub, fetches = "pred:0": the stub still produces and copies prob (it always passes both buffers)'}}


def f32x3_probe(eng, x, n, steps, head_index):
    """Reported beside the headline, never as it: the same K steps with UKBB_PREC_F32X3 (include/ukbb_fcn.h): the FCN head's
    out0 / out1 products from three bf16 pieces per fp32 operand on the dense matrix cores, fp32 accumulation.  Same logits error
    against the fp64 oracle and the same label maps as the fp32 path (tests/test_gpu_parity.py test_f32x3_*), but not the
    instruction the metric names (fp32 MFMA), hence its own field."""
    import torch
    pred = torch.empty((n, H, W), dtype=torch.int32, device=x.device)
    eng.set_precision('f32x3')
    try:
        for _ in range(4):
            eng.run_device(x.data_ptr(), n, H, W, pred_ptr=pred.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.run_device(x.data_ptr(), n, H, W, pred_ptr=pred.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        eng.set_timing(True)
        for _ in range(5):
            eng.run_device(x.data_ptr(), n, H, W, pred_ptr=pred.data_ptr())
        ms, cnt = eng.kernel_times()
        eng.set_timing(False)
        head_us = ms[head_index] / max(1, cnt[head_index]) * 1e3
    finally:
        eng.set_precision('fp32')
    return {'value': round(n * steps / dt, 1), 'unit': 'slices/s', 'ms_per_step': round(dt / steps * 1e3, 4), 'head_us': round(head_us, 1),
            'note': 'same workload with ukbb_fcn_set_precision(UKBB_PREC_F32X3): head products from three bf16 pieces per fp32 '
                    'operand (6 bf16 MFMAs per fp32 MFMA\'s worth, fp32 accumulate); parity identical to fp32; not the headline'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-f32x3-probe', action='store_true', help='skip the extra K steps in UKBB_PREC_F32X3 mode reported as out["f32x3"]')
    ap.add_argument('--pmc-traffic', default=None,
                    help='JSON written by tools/pmc_traffic.py (rocprofv3 --pmc passes of this build); default: the newest '
                         'profiles/r*_pmc_traffic.json whose kernel_source_sha matches the sources in this tree')
    ap.add_argument('--no-kernel-events', action='store_true',
                    help='do not bracket kernels with HIP events in the timed region (roofline becomes null)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the config-5 (aortic U-Net fp32 / bf16) and N = 10 probes')
    ap.add_argument('--cohort-subjects', type=int, default=1000,
                    help='BASELINE configs[3]: synthetic 500-slice subjects run through the device subject pipeline after the timed region '
                         '(other_configs.config4_cohort; every rank takes i mod G; 0 = skip)')
    ap.add_argument('--inflight-probe', action='store_true',
                    help='after the timed region also measure the same steps with two batches in flight on two streams '
                         '(extra field two_batches_in_flight; off by default so that a rocprofv3 trace of the default command '
                         'holds single-stream launches only)')
    ap.add_argument('--sustained-seconds', type=float, default=10.0,
                    help='as the LAST thing of the run (after the side probes and the CPU legs) loop the same step for this long and report '
                         'it as out["sustained"] with the shader clock observed during the loop (0 = skip)')
    ap.add_argument('--rehearsal', action='store_true',
                    help='allow several ranks on one GPU (the 1-GPU rehearsal of the N > 1 launch); without it two ranks that report '
                         'the same PCI bus id make the run fail')
    ap.add_argument('--cpu-leg', choices=['torch', 'c'], default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_leg:
        return cpu_leg(args.cpu_leg)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        if args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d '
                     '(WORLD_SIZE is %d)' % (args.gpus, args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU: the HIP path has no CPU fallback')
    # More ranks than visible GPUs only happens in the 1-GPU rehearsal of the N > 1 launch (tests/test_deploy_gpu.py):
    # ranks then share devices and synchronise over gloo (RCCL refuses two ranks on one GPU); the line says so.
    ndev = torch.cuda.device_count()
    oversub = world > ndev
    dev_index = local_rank % ndev if oversub else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if oversub:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from ukbb_cardiac_amd.arch import MODELS, fcn_macs_per_slice
    from ukbb_cardiac_amd.engine import Engine
    from ukbb_cardiac_amd.phantom import uniform_slices
    from ukbb_cardiac_amd.weights import synthetic_params

    arch = MODELS['FCN_sa']
    params = synthetic_params(arch, 1234)
    eng = Engine(arch, params, device=dev_index)
    n = args.batch
    x = torch.from_numpy(uniform_slices(n, H, W, seed=1 + rank)).to(dev)
    pred = torch.empty((n, H, W), dtype=torch.int32, device=dev)
    eng.reserve(n, H, W)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.run_device(x.data_ptr(), n, H, W, pred_ptr=pred.data_ptr(), stream=stream)

    def barrier():
        if world > 1:
            dist.barrier()

    use_events = not args.no_kernel_events
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    warm_avg = None
    if use_events:
        # untimed survey pass with every kernel bracketed: finds the dominant kernel and
        # fills roofline_detail; the timed region below brackets ONLY the dominant kernel
        # (2 event records per step), which does not perturb it measurably.
        eng.set_timing(True)
        eng.kernel_times(reset=True)
        for _ in range(max(3, args.warmup)):
            step()
        ms, cnt = eng.kernel_times(reset=True)
        warm_avg = [m / max(c, 1) for m, c in zip(ms, cnt)]
        dom = max(range(len(warm_avg)), key=lambda i: warm_avg[i])
        eng.set_timing_kernel(dom)
        eng.kernel_times(reset=True)

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if oversub else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t.item())
    # per-rank evidence for whoever reads the N > 1 line: which device each rank ran on and how long ITS K steps took
    props = torch.cuda.get_device_properties(dev_index)
    bus = None
    if all(hasattr(props, k) for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id')):
        bus = '%04x:%02x:%02x' % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    mine = {'rank': rank, 'local_rank': local_rank, 'device_index': dev_index, 'device_name': props.name, 'pci_bus_id': bus,
            'uuid': str(getattr(props, 'uuid', '')) or None, 'visible_devices': ndev, 'pid': os.getpid(),
            'ms_per_step': round(elapsed / args.steps * 1e3, 4)}
    ranks = [mine]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    ids = [r['pci_bus_id'] or r['uuid'] or 'dev%d' % r['device_index'] for r in ranks]
    shared = len(set(ids)) < len(ids)
    if shared and not args.rehearsal:
        if rank == 0:
            print('bench.py: %d ranks but only %d distinct GPUs %s -- pass --rehearsal if that is intended (not a scaling measurement)'
                  % (world, len(set(ids)), sorted(set(ids))), file=sys.stderr, flush=True)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(3)

    m3, m1 = fcn_macs_per_slice(arch, H, W)
    roofline = None
    detail = None
    if use_events:
        ms, cnt = eng.kernel_times(reset=True)
        eng.set_timing(False)
        names, macs, xmacs = eng.kernel_names(), eng.kernel_macs(), eng.kernel_mfma_macs()
        pmacs = eng.kernel_mfma_macs_issued()          # incl. padding slots of partly filled tiles / Winograd regions
        dom_avg = ms[dom] / max(cnt[dom], 1)          # measured inside the timed region
        avg = list(warm_avg)
        tf = lambda mac, t_ms: 2.0 * mac / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0
        issued = tf(xmacs[dom], dom_avg)              # multiplies this kernel really sends to the matrix pipe
        credited = tf(macs[dom], dom_avg)             # reference-graph FLOPs attributed to it
        traffic, traffic_src = pmc_traffic(args.pmc_traffic, names[dom], n)
        roofline = {'bound': 'mfma', 'kernel': names[dom], 'achieved': round(issued, 2),
                    'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(issued / PEAK_FP32_MFMA_TFLOPS, 4),
                    'traffic': traffic, 'traffic_source': traffic_src,
                    'avg_launch_us': round(dom_avg * 1e3, 2), 'launches_timed': int(cnt[dom]),
                    # achieved / frac count the FLOPs the launch EXECUTES on the matrix pipe (= SQ_INSTS_MFMA x FLOP per
                    # instruction in profiles/): the head runs only the level-0 slice of the 160->64 conv at full
                    # resolution, the other four slices run at low resolution inside the sqg kernels (and are counted
                    # there), Winograd layers run 16/36 of the direct multiplies.
                    'flop_per_launch': 2.0 * xmacs[dom], 'executed_incl_padding_flop_per_launch': 2.0 * pmacs[dom],
                    # the same launch priced with the FLOPs of the reference graph's layers it replaces (can exceed 1;
                    # not a utilisation)
                    'effective_frac_reference_graph': round(credited / PEAK_FP32_MFMA_TFLOPS, 4),
                    'reference_graph_flop_per_launch': 2.0 * macs[dom]}
        is3 = [nm.startswith('conv') and nm != 'conv0_0' for nm in names]
        t3 = sum(a for a, f in zip(avg, is3) if f)
        mac3 = sum(m for m, f in zip(macs, is3) if f)
        xmac3 = sum(m for m, f in zip(xmacs, is3) if f)
        tall = sum(avg)
        detail = {
            'note': 'per-kernel survey from an untimed pass with all kernels bracketed by HIP events; frac = FLOPs the kernel '
                    'executes on the matrix pipe / peak (the utilisation); effective_frac_reference_graph = FLOPs of the '
                    'reference-graph layers it replaces / peak (Winograd layers and the head execute fewer multiplies than '
                    'the reference graph, so this one can exceed 1)',
            'conv3x3_mfma_stack': {'tflops': round(tf(xmac3, t3), 2), 'frac': round(tf(xmac3, t3) / PEAK_FP32_MFMA_TFLOPS, 4),
                                   'effective_frac_reference_graph': round(tf(mac3, t3) / PEAK_FP32_MFMA_TFLOPS, 4),
                                   'us_per_step': round(t3 * 1e3, 1)},
            'all_kernels': {'tflops': round(tf(sum(xmacs), tall), 2),
                            'frac': round(tf(sum(xmacs), tall) / PEAK_FP32_MFMA_TFLOPS, 4),
                            'effective_frac_reference_graph': round(tf(sum(macs), tall) / PEAK_FP32_MFMA_TFLOPS, 4),
                            'us_per_step': round(tall * 1e3, 1)},
            'per_kernel_us': {nm: round(a * 1e3, 1) for nm, a in zip(names, avg)},
            'per_kernel_frac': {nm: round(tf(m, a) / PEAK_FP32_MFMA_TFLOPS, 3) for nm, m, a in zip(names, xmacs, avg)},
            # what the matrix pipe really issues: partly filled tiles and Winograd regions run all their slots (the 24x26 / 12x13 maps
            # of levels 3-4 fill 81 % / 87.5 % of their regions); = SQ_INSTS_MFMA x FLOP per instruction of the counter passes
            'per_kernel_executed_incl_padding_frac': {nm: round(tf(m, a) / PEAK_FP32_MFMA_TFLOPS, 3) for nm, m, a in zip(names, pmacs, avg)},
            'per_kernel_padding_overhead': {nm: round(p / m, 3) for nm, p, m in zip(names, pmacs, xmacs) if m > 0 and p > m * 1.001},
            'per_kernel_effective_frac_reference_graph': {nm: round(tf(m, a) / PEAK_FP32_MFMA_TFLOPS, 3) for nm, m, a in zip(names, macs, avg)},
        }

    # BASELINE configs[3] on every rank (subject i -> rank i mod G, no collective; a barrier lines the clocks up, the slowest rank's
    # time from that barrier to its last result is the cohort's time)
    cohort = None
    if args.cohort_subjects > 0 and not args.no_other_configs:
        mine_c = config4_cohort_probe(dev_index, rank, world, args.cohort_subjects, barrier if world > 1 else None)
        allc = [mine_c]
        if world > 1:
            allc = [None] * world
            dist.all_gather_object(allc, mine_c)
        cohort = allc[0]
        errs = [c['error'] for c in allc if 'error' in c]
        if errs:
            cohort = {'error': errs[0], 'ranks_failed': len(errs)}
        else:
            t_max = max(c['seconds_this_rank'] for c in allc)
            n_sl = args.cohort_subjects * 500
            cohort.update({'value': round(n_sl / t_max, 1), 'unit': 'slices/s', 'subjects_per_s': round(args.cohort_subjects / t_max, 2),
                           'seconds_max_over_ranks': round(t_max, 4), 'n_gpus': world})
            if world > 1:
                cohort['scaling'] = 'strong (one fixed cohort split over the ranks)'
                cohort['per_rank'] = [{k: c.get(k) for k in ('subjects_this_rank', 'seconds_this_rank', 'device_memory_delta_mb')} for c in allc]
            elif 'network_only_slices_per_s' in cohort:
                cohort['pipeline_efficiency_vs_network_only'] = round(cohort['value'] / cohort['network_only_slices_per_s'], 4)

    if rank == 0:
        slices = world * n * args.steps
        value = slices / elapsed_max
        flops_per_slice = 2.0 * (m3 + m1)
        out = {
            'metric': '192x208 SAX slices/sec', 'value': round(value, 1), 'unit': 'slices/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed_max / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'short-axis FCN (FCN_sa, 4 classes) inference, batch=%d synthetic 192x208 slices '
                                   'per GPU resident in HBM, int32 label map out (BASELINE.json configs[1])' % n,
                       'slices_per_gpu_per_step': n, 'height': H, 'width': W, 'weights': 'synthetic seed 1234',
                       'parallelism': 'batch split x%d, no collectives' % world + (
                           ' (REHEARSAL: %d ranks share %d GPU(s), gloo barrier; not a scaling measurement)' % (world, ndev) if oversub else '')},
            # whole step priced with the reference graph's 3.0965 GFLOP per slice: "effective", not a utilisation
            'e2e_effective_tflops_reference_graph': round(value * flops_per_slice / 1e12, 2),
            'e2e_effective_frac_reference_graph': round(value * flops_per_slice / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world), 4),
            'roofline': roofline,
            'backend': ('gloo' if oversub else 'nccl (RCCL)') if world > 1 else None,
            'ranks': ranks, 'per_rank_ms_per_step': [r['ms_per_step'] for r in ranks],
        }
        if detail:
            out['roofline_detail'] = detail
        if world == 1 and args.inflight_probe:
            out['two_batches_in_flight'] = inflight_probe(arch, params, x, n, args.steps)
        if world == 1 and not args.no_f32x3_probe:
            out['f32x3'] = f32x3_probe(eng, x, n, args.steps, eng.kernel_names().index('head'))
        if world == 1 and not args.no_other_configs:
            try:
                out['other_configs'] = other_configs_probe(dev)
            except Exception as e:                                    # never lose the headline line to a side probe
                out['other_configs'] = {'error': repr(e)[-300:]}
            try:
                out['other_configs']['drop_in_hipsession'] = drop_in_hipsession_probe(dev_index)
            except Exception as e:
                out['other_configs']['drop_in_hipsession'] = {'error': repr(e)[-300:]}
        if cohort is not None:
            out.setdefault('other_configs', {})['config4_cohort'] = cohort
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        # last, and long enough for an outside sampler at a 5 s period to land inside it at least once
        if world == 1 and args.sustained_seconds > 0:
            try:
                sus = sustained_probe(step, n, dev_index, args.sustained_seconds, elapsed_max / args.steps * 1e3)
                out['sustained'] = sus
                if roofline and sus.get('shader_clock_mhz_in_loop'):
                    ghz = sus['shader_clock_mhz_in_loop'] / 1e3
                    peak_obs = props.multi_processor_count * 4 * 64 * ghz / 1e3       # CUs x SIMDs x FLOP/clk x GHz -> TFLOP/s
                    roofline['peak_at_observed_clock'] = round(peak_obs, 1)
                    roofline['frac_at_observed_clock'] = round(roofline['achieved'] / peak_obs, 4)
                    roofline['observed_clock_mhz'] = sus['shader_clock_mhz_in_loop']
            except Exception as e:
                out['sustained'] = {'error': repr(e)[-300:]}
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
