#!/usr/bin/env python3
"""Static instruction mix of the hottest loop of every kernel in an AMDGPU .s file.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only X.hip -o X.s
    python tools/isa_loop_stats.py X.s [substring]

For each kernel: finds the outermost loop (label .. backward branch) holding the most MFMAs, and prints MFMA / VALU / LDS / VMEM / SALU counts.  On gfx950 fp32 MFMA and
VALU instructions serialise on a SIMD (tools/mfma_coissue.hip), so est = mfma_cycles + 5*VALU is
the loop's issue-bound estimate and mfma_cycles/est the ceiling of the matrix-pipe fraction.
"""
import re, sys, subprocess

def demangle(n):
    try:
        return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n

def classify(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_'): return 'salu'
    return None

def mfma_cycles(op):
    if '32x32x2' in op: return 64
    if '16x16x4' in op: return 32
    if '32x32x16' in op or '32x32x8' in op: return 32 if 'bf16' in op or 'f16' in op else 64
    if '16x16x32' in op or '16x16x16' in op: return 16
    return 32

def main():
    path = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ''
    lines = open(path).read().split('\n')
    kernels = []; cur = None
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m: cur = [m.group(1), i, None]; kernels.append(cur)
        if cur and 's_endpgm' in l and cur[2] is None: cur[2] = i
    for name, a, b in kernels:
        dn = demangle(name)
        if sub not in dn: continue
        body = lines[a:(b or len(lines))]
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r'^(\.LBB\d+_\d+):', l)
            if m: labels[m.group(1)] = i
        best = None
        for i, l in enumerate(body):
            m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                lo, hi = labels[m.group(1)], i
                cnt = {'mfma': 0, 'valu': 0, 'lds': 0, 'vmem': 0, 'salu': 0}; mc = 0
                for x in body[lo:hi + 1]:
                    t = x.split()
                    if not t or t[0].startswith((';', '.')): continue
                    c = classify(t[0])
                    if c: cnt[c] += 1
                    if c == 'mfma': mc += mfma_cycles(t[0])
                if cnt['mfma'] and (best is None or cnt['mfma'] > best[0]['mfma'] or
                                    (cnt['mfma'] == best[0]['mfma'] and hi - lo > best[2] - best[1])):
                    best = (cnt, lo, hi, mc)
        if best:
            cnt, lo, hi, mc = best
            est = mc + 5 * cnt['valu']
            print('%-110s mfma %4d (%6d cyc) valu %4d lds %4d vmem %3d salu %4d  ceiling %.2f' %
                  (dn[:110], cnt['mfma'], mc, cnt['valu'], cnt['lds'], cnt['vmem'], cnt['salu'], mc / est))

if __name__ == '__main__':
    main()
