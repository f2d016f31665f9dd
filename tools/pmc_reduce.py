"""Reduce a rocprofv3 --pmc counter_collection CSV: per kernel (name prefix) and counter, mean value per dispatch.

    python tools/pmc_reduce.py gpurun_out/pmc_sq [substr ...]
"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    want = sys.argv[2:] or ['k_layer', 'k_graph_update']
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r['Kernel_Name']
                key = next((w for w in want if w in k), None)
                if key is None:
                    continue
                if 'ILi' in k:
                    key += '<' + k.split('ILi')[1].split('E')[0] + '>'
                a = acc[(key, r['Counter_Name'])]
                a[0] += float(r['Counter_Value'])
                a[1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        print(f'{k:22s} {c:26s} n={n:5d} mean={v / n:16.1f}')


if __name__ == '__main__':
    main()
