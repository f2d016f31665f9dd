"""How well do the sub-batch streams of the sampler overlap?  Reads rocprofv3 --kernel-trace CSVs of a two-stream and a
one-stream bench pass and prints, per kernel class, the mean duration in each, and for the two-stream run the mean
duration of launches that started while ANOTHER queue's fused layer kernel was running against those that did not.
usage: overlap_probe.py two_stream_kernel_trace.csv [one_stream_kernel_trace.csv]"""
import csv
import sys
from collections import defaultdict


def load(path):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            name = r['Kernel_Name']
            short = name.split('(')[0].replace('void ', '')
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), short))
    rows.sort()
    return rows


def main():
    two = load(sys.argv[1])
    # keep the second half (steady state, dense steps come last)
    t0, t1 = two[0][0], two[-1][1]
    print(f'two-stream trace: {len(two)} dispatches over {(t1 - t0) / 1e6:.1f} ms, queues {sorted({r[2] for r in two})}')
    layers = [(s, e, q) for s, e, q, n in two if n.startswith('k_layer')]
    import bisect
    starts = [x[0] for x in layers]
    agg = defaultdict(lambda: [0, 0, 0, 0])
    for s, e, q, n in two:
        # is a k_layer of another queue running at time s?
        i = bisect.bisect_right(starts, s)
        under = any(ls <= s < le and lq != q for ls, le, lq in layers[max(0, i - 4):i])
        a = agg[n]
        if under:
            a[0] += e - s; a[1] += 1
        else:
            a[2] += e - s; a[3] += 1
    one = defaultdict(lambda: [0, 0])
    if len(sys.argv) > 2:
        for s, e, q, n in load(sys.argv[2]):
            one[n][0] += e - s; one[n][1] += 1
    print(f'{"kernel":34s} {"under other k_layer":>22s} {"not under":>18s} {"one-stream run":>18s}')
    tot_u = tot_n = 0
    for n, a in sorted(agg.items(), key=lambda kv: -(kv[1][0] + kv[1][2])):
        if a[1] + a[3] < 50:
            continue
        u = f'{a[0] / max(a[1], 1) / 1e3:8.1f} us x{a[1]:6d}'
        v = f'{a[2] / max(a[3], 1) / 1e3:8.1f} us x{a[3]:6d}'
        o = f'{one[n][0] / max(one[n][1], 1) / 1e3:8.1f} us x{one[n][1]:6d}' if one[n][1] else ''
        print(f'{n[:34]:34s} {u:>22s} {v:>18s} {o:>18s}')
    busy, cs, ce = 0, two[0][0], two[0][1]
    for s, e, _, _ in two[1:]:
        if s > ce:
            busy += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    tot = sum(e - s for s, e, _, _ in two)
    print(f'span {(t1 - t0) / 1e6:.1f} ms, union busy {busy / 1e6:.1f} ms, summed kernel time {tot / 1e6:.1f} ms '
          f'(overlap factor {tot / busy:.2f})')


if __name__ == '__main__':
    main()
