"""Roofline figures of the bf16 aortic U-Net forward (BASELINE config 5) from what tools/profile_unet.sh collected:
    python tools/unet_roofline.py gpurun_out/r03_unet
Sums the hardware counters of every kernel launch of the counter passes (`bench_unet.py 100 bf16 3`: FETCH_SIZE, WRITE_SIZE,
SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES), divides by the number of forwards, and prices them with the step time of the
un-profiled run (unet.txt).  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction, MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(out):
    from ukbb_cardiac_amd.arch import MODELS, fcn_macs_per_slice
    n, H, W = 100, 256, 256
    m3, m1 = fcn_macs_per_slice(MODELS['UNet_ao'], H, W)
    flop = 2.0 * (m3 + m1) * n
    ms = {}
    for line in open(os.path.join(out, 'unet.txt')):
        m = re.match(r'UNet_ao (\w+): N=100 256x256: ([\d.]+) ms/step', line)
        if m:
            ms[m.group(1)] = float(m.group(2))
    tot, launches = defaultdict(float), defaultdict(int)
    forwards = None
    for log in sorted(glob.glob(os.path.join(out, 'pmc', 'pass*.log'))):
        for line in open(log):
            m = re.match(r'forwards_total=(\d+)', line)
            if m:
                forwards = int(m.group(1))
    for f in glob.glob(os.path.join(out, 'pmc', '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get('Kernel_Name', '')
                if 'ukbb::' not in k:
                    continue
                c = row['Counter_Name']
                tot[c] += float(row.get('Counter_Value', 0) or 0)
                launches[c] += 1
    if not forwards or 'FETCH_SIZE' not in tot:
        print('counter passes incomplete: forwards=%s counters=%s' % (forwards, sorted(tot)))
        return
    t = ms['bf16'] * 1e-3
    fetch, write = tot['FETCH_SIZE'] / forwards * 1024, tot['WRITE_SIZE'] / forwards * 1024
    hbm = 2 * fetch + write
    per_fwd = {c: launches[c] / forwards for c in launches}
    # algorithmic HBM bytes of the plan (bf16 activations read once / written once, pred out, image in; halos and weights excluded)
    lv = [H * W * 16, H * W // 4 * 32, H * W // 16 * 64, H * W // 64 * 128, H * W // 256 * 256]   # elements per map and level
    alg = H * W * 4                                             # image in
    alg += lv[0] * 2 * 3                                        # conv0 written, read by conv1_0 and by up0_0
    for l in range(1, 5):
        alg += lv[l] * 2 * 2                                    # conv{l}_0 written + read
        alg += lv[l] * 2 * (3 if l < 4 else 2)                  # conv{l}_1 written, read by the next level and (l < 4) by the decoder
    for l in range(3, -1, -1):
        # up{l}_t, up{l}_0, up{l}_1 written + read; at level 0 (r04: fused tail, kernels_tail.hip) only up0_t exists in HBM
        alg += lv[l] * 2 * 2 * (3 if l > 0 else 1)
    alg += H * W * 4                                            # pred out
    alg *= n
    print('aortic U-Net, UKBB_PREC_BF16, N = %d x %dx%d, %s ms per forward (un-profiled run), %d kernel launches per forward' % (
        n, H, W, ms['bf16'], round(per_fwd.get('FETCH_SIZE', 0))))
    print('matrix: %.1f GFLOP algorithmic per forward (%.1f M MAC per slice) / %.3f ms = %.1f TFLOP/s = %.3f of the 2500 TFLOP/s dense bf16 peak' % (
        flop / 1e9, (m3 + m1) / 1e6, ms['bf16'], flop / t / 1e12, flop / t / 2.5e15))
    if 'SQ_INSTS_MFMA' in tot:
        mf = tot['SQ_INSTS_MFMA'] / forwards
        print('        SQ_INSTS_MFMA per forward %.4g (x 32768 FLOP for v_mfma_f32_32x32x16_bf16 = %.1f GFLOP issued incl. zero-padded rows / taps)' % (
            mf, mf * 32768 / 1e9))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in tot and 'SQ_BUSY_CYCLES' in tot:
        print('        SQ_VALU_MFMA_BUSY_CYCLES / forward %.4g ; SQ_BUSY_CYCLES / forward %.4g' % (
            tot['SQ_VALU_MFMA_BUSY_CYCLES'] / forwards, tot['SQ_BUSY_CYCLES'] / forwards))
    print('HBM:    FETCH_SIZE %.1f MB x 2 (gfx950 correction) + WRITE_SIZE %.1f MB = %.1f MB per forward / %.3f ms = %.0f GB/s = %.3f of 8000 GB/s' % (
        fetch / 1e6, write / 1e6, hbm / 1e6, ms['bf16'], hbm / t / 1e9, hbm / t / 8e12))
    print('        algorithmic bytes of the plan (every stored map written once and read once per consumer, image in, labels out; no halos, no weights): '
          '%.1f MB per forward = %.2f MB per slice -> traffic / algorithmic = %.2f' % (alg / 1e6, alg / 1e6 / n, hbm / alg))
    if 'fp32' in ms:
        print('fp32 path of the same engine: %.2f ms per forward -> bf16 speed-up %.2fx' % (ms['fp32'], ms['fp32'] / ms['bf16']))
    # what bench.py quotes as config 5's measured traffic: stamped with the kernel sources it was collected from
    import json
    import bench
    with open(os.path.join(out, 'unet_traffic.json'), 'w') as f:
        json.dump({'kernel_source_sha': bench.kernel_source_sha(), 'hbm_bytes_per_forward': hbm, 'fetch_size_bytes': fetch, 'write_size_bytes': write,
                   'algorithmic_bytes_per_forward': alg, 'forwards_counted': forwards, 'ms_per_forward_unprofiled': ms['bf16'],
                   'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/bench_unet.py 100 bf16 3 (tools/profile_unet.sh); HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB units, gfx950 correction)'}, f)


if __name__ == '__main__':
    main(sys.argv[1])
