"""Timeline of ONE training step from a rocprofv3 kernel trace of tools/train_probe.py: the dispatches between the
last two optimizer launches, per hardware queue, with the gaps on the queue that carries the chain.
    python tools/train_timeline.py <dir with *kernel_trace.csv> [list]"""
import csv
import glob
import sys


def main():
    rows = []
    for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    adam = [i for i, r in enumerate(rows) if 'FusedAdam' in r['Kernel_Name']]
    # one optimizer step = a run of consecutive Adam launches; take the dispatches between the last two runs
    runs = []
    for i in adam:
        if runs and i - runs[-1][-1] <= 2:
            runs[-1].append(i)
        else:
            runs.append([i])
    # train_probe's last five steps run under the library's event timer (serialised): take a step of its timed loop
    k = -8 if len(runs) >= 9 else -2
    a, b = runs[k][-1] + 1, runs[k + 1][-1] + 1
    seg = rows[a:b]
    t0 = int(seg[0]['Start_Timestamp'])
    t1 = max(int(r['End_Timestamp']) for r in seg)
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f'step: {len(seg)} dispatches, wall {(t1 - t0) / 1e3:.1f} us, GPU busy (union) {busy / 1e3:.1f} us, '
          f'sum of durations {sum(e - s for s, e in iv) / 1e3:.1f} us')
    byq = {}
    for r in seg:
        byq.setdefault(r['Queue_Id'], []).append(r)
    for q, rs in byq.items():
        d = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
        gaps = [int(rs[i + 1]['Start_Timestamp']) - int(rs[i]['End_Timestamp']) for i in range(len(rs) - 1)]
        big = sorted(gaps, reverse=True)[:5]
        print(f'  queue {q}: {len(rs)} dispatches, {d / 1e3:.1f} us of kernels, gaps: total {sum(g for g in gaps if g > 0) / 1e3:.1f} us, '
              f'largest {[round(g / 1e3, 1) for g in big]}')
    if len(sys.argv) > 2:
        for r in seg:
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{r['Queue_Id']} {r['Kernel_Name'][:60]}")


if __name__ == '__main__':
    main()
