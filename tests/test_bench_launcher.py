"""CPU: bench.py's own launcher (`python bench.py --gpus N` outside torchrun) with two child ranks over gloo.

`--dry-run-cpu` skips the sampling (no GPU here) and prints the multi-rank bookkeeping of the JSON line; what is
checked is what the driver's N > 1 runs depend on: the children get RANK / WORLD_SIZE / MASTER_*, meet in ONE process
group, shard the generations by global graph id, rank 0 alone prints one JSON line that certifies the run (world size
and every rank's identity as gathered over the group, per-rank elapsed times), and a rank that dies takes the run
down with its exit code instead of leaving the others in the rendezvous."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_launcher_two_ranks_print_one_certified_line():
    p = _run(['--gpus', '2', '--dry-run-cpu', '--steps', '2', '--warmup', '1'])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['scaling'] == 'weak' and 'dry_run' in r and r['value'] is None
    mg = r['multi_gpu']
    assert mg['world_size'] == 2 and mg['backend'] == 'gloo'
    assert [x['rank'] for x in mg['ranks']] == [0, 1]
    assert len({x['pid'] for x in mg['ranks']}) == 2                      # two processes, not one counted twice
    assert len(mg['per_rank_elapsed_s']) == 2 and all(e > 0 for e in mg['per_rank_elapsed_s'])
    assert r['elapsed_max_s'] >= max(mg['per_rank_elapsed_s']) - 1e-3     # the line's time is the MAX over ranks
    assert r['config']['shards'] == [[0, 64], [64, 64]]                   # weak scaling: 64 graphs per rank, ids disjoint


def test_launcher_strong_scaling_shards_the_64_generations():
    p = _run(['--gpus', '2', '--dry-run-cpu', '--total-graphs', '64'])
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    assert r['scaling'] == 'strong' and r['config']['shards'] == [[0, 32], [32, 32]]


def test_launcher_reports_a_dead_rank_and_stops_the_others():
    t0 = time.time()
    p = _run(['--gpus', '2', '--dry-run-cpu', '--fail-rank', '1'])
    assert p.returncode == 7, (p.returncode, p.stderr[-2000:])
    assert 'rank 1 exited with code 7' in p.stderr
    assert time.time() - t0 < 90                                           # not rank 0's collective timeout (minutes)
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
