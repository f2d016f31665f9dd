"""pmc_traffic.py - reduce rocprofv3 --pmc CSVs (one pass FETCH_SIZE, one pass WRITE_SIZE) to HBM bytes per
launch for each kernel, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section): both counters are
in KiB; FETCH_SIZE tallies 128-B requests at 64 B, so the read side is doubled.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def reduce_dir(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                k = row['Kernel_Name']
                tot[k] += float(row['Counter_Value'])
                cnt[k] += 1
    return tot, cnt


def short(name):
    return name.split('(')[0].replace('void ', '').strip()


def main():
    fdir, wdir, out = sys.argv[1:4]
    ft, fc = reduce_dir(fdir, 'FETCH_SIZE')
    wt, wc = reduce_dir(wdir, 'WRITE_SIZE')
    res = {}
    for k in sorted(set(ft) | set(wt), key=lambda k: -(2 * ft.get(k, 0) + wt.get(k, 0))):
        n = fc.get(k) or wc.get(k)
        fetch = 2.0 * 1024.0 * ft.get(k, 0.0) / max(fc.get(k, 1), 1)
        write = 1024.0 * wt.get(k, 0.0) / max(wc.get(k, 1), 1)
        res[short(k)] = {'launches': n, 'fetch_bytes_per_launch': fetch, 'write_bytes_per_launch': write,
                         'hbm_bytes_per_launch': fetch + write}
    meta = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of '
                      'python3 bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline',
            'corrections': 'KiB -> bytes (x1024); FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B)',
            'kernels': res}
    with open(out, 'w') as fh:
        json.dump(meta, fh, indent=1)
    for k, v in list(res.items())[:12]:
        print(f"{k[:60]:60s} n={v['launches']:6d} fetch {v['fetch_bytes_per_launch'] / 1e6:9.3f} MB  write "
              f"{v['write_bytes_per_launch'] / 1e6:9.3f} MB")


if __name__ == '__main__':
    main()
