"""Top kernels of a rocprofv3 --kernel-trace run stored as a rocpd sqlite database (the default output
format of rocprofv3 7.x): name, calls, total us, average us, share.  usage: rocprof_top.py results.db [N] [divide_calls_by]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    rows = list(db.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    tot = sum(r[2] for r in rows)
    print(f'total kernel time {tot / 1e3 / div:.3f} us-thousands per unit ({len(rows)} kernels)')
    for r in rows[:top]:
        print(f'{r[0][:90]:90s} calls {r[1] / div:8.1f}  total_us {r[2] / 1e3 / div:10.1f}  avg_us {r[3] / 1e3:8.2f}  {r[4]:5.1f}%')


if __name__ == '__main__':
    main()
