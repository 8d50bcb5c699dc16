"""Per-kernel table of one UNet-LSTM cine (100 frames of 256 x 256) from what tools/profile_lstm.sh collected:
    python tools/lstm_bytes.py gpurun_out/r05_unet_lstm
kernel_stats.csv (rocprofv3 --kernel-trace --stats) gives calls and average duration; the counter passes give, per launch, HBM bytes
= (2 * FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction, MI355X_MICROARCH.md HBM section) and SQ_INSTS_MFMA.  The algorithmic bytes of
an LSTM step are printed beside the measured ones (Wn windows x H x W pixels; 16 feature + 16 hidden channels in, fp32)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short                                          # noqa: E402


def cines(path):
    for line in open(path):
        m = re.match(r'cines_total=(\d+)', line)
        if m:
            return int(m.group(1))
    return None


def main(out):
    n_trace = cines(os.path.join(out, 'under_rocprof.txt'))
    stats = {}
    with open(os.path.join(out, 'kernel_stats.csv')) as fh:
        for row in csv.DictReader(fh):
            k = short(row['Name'])
            if k.startswith('__amd') or 'at::' in k:
                continue
            stats[k] = (int(row['Calls']), float(row['AverageNs']) / 1e3)
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(out, 'pmc', '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get('Kernel_Name', ''))
                c = row['Counter_Name']
                tot[k][c] += float(row.get('Counter_Value', 0) or 0)
                cnt[k][c] += 1
    print('%-44s %9s %9s %10s %10s %10s %12s' % ('kernel', 'calls/cine', 'avg us', 'ms/cine', 'HBM MB/launch', 'TB/s', 'MFMA/launch'))
    total_ms, total_hbm = 0.0, 0.0
    for k, (calls, us) in sorted(stats.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        per = calls / n_trace if n_trace else float('nan')
        mb = None
        if cnt[k].get('FETCH_SIZE') and cnt[k].get('WRITE_SIZE'):
            mb = (2 * tot[k]['FETCH_SIZE'] / cnt[k]['FETCH_SIZE'] + tot[k]['WRITE_SIZE'] / cnt[k]['WRITE_SIZE']) * 1024 / 1e6
        mf = tot[k]['SQ_INSTS_MFMA'] / cnt[k]['SQ_INSTS_MFMA'] if cnt[k].get('SQ_INSTS_MFMA') else None
        ms = per * us / 1e3
        total_ms += ms
        if mb is not None:
            total_hbm += mb * per
        print('%-44s %9.1f %9.1f %10.3f %10s %10s %12s' % (k[:44], per, us, ms, '%.1f' % mb if mb is not None else '-',
                                                            '%.2f' % (mb / us) if mb is not None else '-', '%.4g' % mf if mf is not None else '-'))
    print('sum of kernel time %.2f ms per cine; measured HBM bytes %.2f GB per cine' % (total_ms, total_hbm / 1e3))
    # what bench.py quotes as the cine's measured traffic: stamped with the kernel sources it was collected from
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    # the ConvLSTM launches by NAME: lstm_* kernels (cell / output / tile, the bf16 walker form) and the fused gate-conv forms of the
    # F(2x4) kernel, whose 4th template argument LS is 1 (x pass) or 2 (time step) -- wino24_pc_kernel<TBW, BF, WM, LS, ...>
    import re
    is_lstm = re.compile(r'(^|[^a-z])lstm_|wino24_pc_kernel<\s*\d+\s*,\s*\w+\s*,\s*\d+\s*,\s*[12]\s*[,>]|ws_main.*\bLS\b|lstm_ws')
    picked = [k for k in stats if is_lstm.search(k)]
    if not picked:
        sys.exit('lstm_bytes: no ConvLSTM kernel matched among %s -- kernel or template argument renamed? refusing to write lstm_traffic.json' % sorted(stats))
    lstm_mb = sum((2 * tot[k]['FETCH_SIZE'] / cnt[k]['FETCH_SIZE'] + tot[k]['WRITE_SIZE'] / cnt[k]['WRITE_SIZE']) * 1024 / 1e6 * (stats[k][0] / n_trace)
                  for k in picked if cnt[k].get('FETCH_SIZE') and cnt[k].get('WRITE_SIZE'))
    print('ConvLSTM kernels counted: %s' % '; '.join(k[:60] for k in picked))
    with open(os.path.join(out, 'lstm_traffic.json'), 'w') as f:
        json.dump({'kernel_source_sha': bench.kernel_source_sha(), 'hbm_bytes_per_cine': total_hbm * 1e6, 'hbm_bytes_per_cine_lstm_kernels': lstm_mb * 1e6,
                   'kernel_ms_per_cine': total_ms,
                   'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/bench_unet_lstm.py (tools/profile_lstm.sh); HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB units, gfx950 correction), '
                             'per-kernel means x launches per cine'}, f)
    Wn, HW = 100, 256 * 256
    m = Wn * HW * 4 / 1e6                                               # MB per channel of one step's maps
    print('algorithmic bytes of one LSTM step (MB): read x 16 ch %.0f + h 16 ch %.0f + c %.0f, write c %.0f + h %.0f = %.0f;'
          ' a gates round trip (64 ch written, then read) adds %.0f' % (16 * m, 16 * m, 16 * m, 16 * m, 16 * m, 80 * m, 128 * m))


if __name__ == '__main__':
    main(sys.argv[1])
