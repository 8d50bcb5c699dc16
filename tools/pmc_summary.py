"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv files under a
directory tree) into per-kernel means per launch.  python tools/pmc_summary.py DIR [DIR...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'^void\s+', '', name)
    name = re.sub(r'\([^()]*\)$', '', name)                      # the trailing argument list only: '(anonymous namespace)' has parentheses too
    return name.replace('(anonymous namespace)::', '').replace('ukbb::', '')


def main(dirs):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row.get('Kernel_Name', ''))
                    if not k or k.startswith('__amd') or 'at::' in k:
                        continue
                    c = row.get('Counter_Name')
                    v = float(row.get('Counter_Value', 0) or 0)
                    a = acc[k][c]
                    a[0] += v
                    a[1] += 1
    counters = sorted({c for k in acc for c in acc[k]})
    print('kernel,' + ','.join(counters))
    for k in sorted(acc):
        print(k.replace(',', ';') + ',' + ','.join('%.4g' % (acc[k][c][0] / acc[k][c][1]) if acc[k][c][1] else '' for c in counters))


if __name__ == '__main__':
    main(sys.argv[1:] or ['.'])
