"""HBM bytes per launch per kernel from a tools/pmc_summary.py CSV -> JSON (profiles/rNN_pmc_traffic.json).

    python tools/pmc_traffic.py gpurun_out/pmc/summary.csv > profiles/r01_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced
reads (MI355X_MICROARCH.md, HBM section), hence hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# GPU kernels that are exactly one launch of the engine's plan (bench.py looks its dominant kernel up by plan name)
ENGINE_NAMES = (('fcn_head', 'head'), ('sqg_multi', 'sqg1-4'))


def main(path):
    out = {'_note': 'HBM traffic per launch from rocprofv3 --pmc passes of `python3 bench.py --no-cpu-baseline --no-f32x3-probe '
                    '--no-kernel-events --steps 3 --warmup 1` (tools/run_pmc.sh: FETCH_SIZE and WRITE_SIZE collected in '
                    'separate passes, never with --kernel-trace). Units are KB; on gfx950 FETCH_SIZE reports half the '
                    'bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.',
           'kernels': {}, 'engine_kernels': {}}
    from bench import BATCH, kernel_source_sha
    out['kernel_source_sha'] = kernel_source_sha()       # bench.py quotes this file only for the same kernel sources
    out['batch'] = BATCH
    with open(path) as fh:
        for row in csv.DictReader(fh):
            try:
                f, w = float(row['FETCH_SIZE']), float(row['WRITE_SIZE'])
            except (KeyError, ValueError):
                continue
            out['kernels'][row['kernel']] = {'fetch_size_kb': f, 'write_size_kb': w, 'hbm_bytes': (2 * f + w) * 1024}
            for prefix, plan_name in ENGINE_NAMES:
                if row['kernel'].startswith(prefix):
                    out['engine_kernels'][plan_name] = {'gpu_kernel': row['kernel'], 'hbm_bytes': (2 * f + w) * 1024}
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main(sys.argv[1])
