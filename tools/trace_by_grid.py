"""Groups a rocprofv3 kernel_trace.csv by (kernel, grid, LDS): launches, mean duration, share of the total.
    python tools/trace_by_grid.py <dir with *kernel_trace.csv> [top N]"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    acc = defaultdict(lambda: [0, 0.0])
    total = 0.0
    for f in files:
        for r in csv.DictReader(open(f)):
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            grid = (r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'), r.get('Workgroup_Size_X'))
            k = (r['Kernel_Name'][:70], grid)
            acc[k][0] += 1
            acc[k][1] += d
            total += d
    print(f'total {total / 1e3:.2f} ms in {sum(v[0] for v in acc.values())} dispatches')
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f'{k[0]:70s} grid {str(k[1]):28s} n={v[0]:5d} mean {v[1] / v[0]:8.1f} us  {100 * v[1] / total:5.1f}%')


if __name__ == '__main__':
    main()
