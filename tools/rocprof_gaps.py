"""GPU busy time vs span from a rocprofv3 --kernel-trace rocpd database: is a phase bound by the device or by
the host that feeds it?  Prints, for the dispatches between the fractions lo and hi of the trace, span, union of busy intervals, summed kernel
time (> union when streams overlap), idle time and the share of each queue.
usage: rocprof_gaps.py results.db [lo=0.5] [hi=1.0]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    rows = list(db.execute('select start, end, queue_id, name from kernels order by start'))
    rows = rows[int(len(rows) * lo):int(len(rows) * hi)]
    span = rows[-1][1] - rows[0][0]
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    for s, e, _, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(e - s for s, e, _, _ in rows)
    print(f'{len(rows)} dispatches: span {span / 1e6:.3f} ms, busy (union) {busy / 1e6:.3f} ms, summed {tot / 1e6:.3f} ms, '
          f'idle {100 * (1 - busy / span):.1f}%')
    q = {}
    for s, e, qi, _ in rows:
        q[qi] = q.get(qi, 0) + e - s
    for qi, v in q.items():
        print(f'  queue {qi}: {v / 1e6:.3f} ms')
    gaps = sorted(((rows[i + 1][0] - max(r[1] for r in rows[max(0, i - 8):i + 1]), rows[i][3][:50], rows[i + 1][3][:50])
                   for i in range(len(rows) - 1)), reverse=True)
    print('largest idle gaps (us, after kernel, before kernel):')
    for g in gaps[:12]:
        print(f'  {g[0] / 1e3:8.1f}  {g[1]}  ->  {g[2]}')


if __name__ == '__main__':
    main()
