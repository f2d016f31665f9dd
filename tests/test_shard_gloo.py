"""N>1 path on CPU: two gloo processes run the sharding / reduction logic bench.py and the sampler use
(graphs are independent, rank r takes a contiguous block, RNG keyed by global graph id)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'graph-generation-edge_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    import torch.distributed as td
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    td.init_process_group('gloo', rank=rank, world_size=world)
    from edgehip import shard
    import helpers as H
    base, count = shard.shard_generations(total, rank, world)
    # the uniforms rank r draws for its local graph j are those of global graph base + j
    u_local = H.rng_node_uniforms(7, [5] * count, 3, gid_base=base)
    u_global = H.rng_node_uniforms(7, [5] * total, 3, gid_base=0)
    same = bool(np.array_equal(u_local, u_global[base * 5:(base + count) * 5]))
    mx = shard.max_over_ranks(1.0 + rank)
    counts = shard.gather_edge_counts([100 * (base + j) for j in range(count)])
    td.barrier()
    td.destroy_process_group()
    q.put((rank, base, count, same, mx, counts.tolist()))


@pytest.mark.parametrize('total', [7, 8])
def test_two_rank_sharding(total):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    covered = []
    for rank, base, count, same, mx, counts in got:
        assert same, 'shard RNG streams differ from the single-process streams'
        assert mx == 2.0
        assert counts == [100 * g for g in range(total)]
        covered += list(range(base, base + count))
    assert covered == list(range(total))
    assert abs(got[0][2] - got[1][2]) <= 1


def test_shard_edges():
    from edgehip import shard
    assert shard.shard_generations(0, 0, 4) == (0, 0)
    assert [shard.shard_generations(10, r, 4) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]
    with pytest.raises(ValueError):
        shard.shard_generations(4, 4, 4)
    assert shard.max_over_ranks(3.5) == 3.5


def _train_worker(rank, world, port, q):
    import torch.distributed as td
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    td.init_process_group('gloo', rank=rank, world_size=world)
    from edgehip import shard
    g = torch.Generator().manual_seed(100 + rank)
    params = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2))]
    params[0].grad = torch.randn(3, 5, generator=g)
    params[1].grad = torch.randn(7, generator=g) if rank == 0 else None       # missing on one rank: zeros
    params[2].requires_grad_(False)
    bucket = shard.allreduce_gradients(params)
    P = shard.sum_over_ranks(10 + rank)
    Pd = shard.sum_over_ranks_device(10 + rank, 'cpu')
    assert torch.is_tensor(Pd) and Pd.dtype == torch.float32 and float(Pd) == P
    t_all = shard.all_gather_cat(torch.tensor([rank * 2, rank * 2 + 1]))
    # the single per-step all-gather of the training step: [pair count, t, kl] of every rank
    Pg, tg, klg = shard.gather_step_scalars(746031 + rank, torch.tensor([5 + rank, 500 - rank]),
                                            torch.tensor([0.25 + rank, 1e-20 * (rank + 1)]))
    assert float(Pg) == 746031 * 2 + 1 and tg.dtype == torch.int64 and tg.tolist() == [5, 500, 6, 499]
    assert klg.dtype == torch.float32 and torch.equal(klg, torch.tensor([0.25, 1e-20, 1.25, 2e-20]))
    td.barrier()
    td.destroy_process_group()
    q.put((rank, params[0].grad.tolist(), params[1].grad.tolist(), params[2].grad, P, t_all.tolist(), bucket.numel()))


def test_two_rank_gradient_bucket_and_history_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted((q.get(timeout=120) for _ in ps), key=lambda x: x[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    g0 = torch.Generator().manual_seed(100)
    a0, b0 = torch.randn(3, 5, generator=g0), torch.randn(7, generator=g0)
    a1 = torch.randn(3, 5, generator=torch.Generator().manual_seed(101))
    for rank, ga, gb, gc, P, t_all, nb in got:
        assert torch.allclose(torch.tensor(ga), a0 + a1) and torch.allclose(torch.tensor(gb), b0)
        assert gc is None and P == 21.0 and t_all == [0, 1, 2, 3] and nb == 22


def _arena_worker(rank, world, port, q):
    import torch.distributed as td
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    td.init_process_group('gloo', rank=rank, world_size=world)
    from edgehip import shard, train
    params = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7))]
    arena = train.GradArena(params)
    arena.zero_and_attach()
    views_ok = all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, arena.views))
    # what the fused backward does: add straight into the views
    params[0].grad += float(rank + 1)
    params[1].grad += 10.0 * (rank + 1)
    shard.allreduce_flat(arena.flat)
    still = all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, arena.views))
    arena.zero_and_attach()
    zeroed = float(arena.flat.abs().sum()) == 0.0 and arena.matches(params)
    td.barrier()
    td.destroy_process_group()
    q.put((rank, views_ok and still and zeroed, arena.flat.numel()))


def _arena_values(rank, world, port, q):
    import torch.distributed as td
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    td.init_process_group('gloo', rank=rank, world_size=world)
    from edgehip import shard, train
    params = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7))]
    arena = train.GradArena(params)
    arena.zero_and_attach()
    params[0].grad += float(rank + 1)
    params[1].grad += 10.0 * (rank + 1)
    shard.allreduce_flat(arena.flat)
    td.barrier()
    td.destroy_process_group()
    q.put((rank, params[0].grad.tolist(), params[1].grad.tolist()))


@pytest.mark.parametrize('worker', ['layout', 'values'])
def test_two_rank_gradient_arena(worker):
    """The flat gradient arena of the fused training step: p.grad are views, the all-reduce is in place."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    fn = _arena_worker if worker == 'layout' else _arena_values
    ps = [ctx.Process(target=fn, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted((q.get(timeout=120) for _ in ps), key=lambda x: x[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for g in got:
        if worker == 'layout':
            assert g[1] is True and g[2] == 22
        else:
            assert g[1] == [[3.0] * 5] * 3 and g[2] == [30.0] * 7
