"""shard.py - how the path spreads over GPUs: one process per GPU, graphs are independent.

Sampling: rank r of G takes a contiguous block of the `num_generation` graphs and keys its
RNG streams with the GLOBAL graph id (`gid_base`), so the union of the shards is the same set of graphs a
single process would have sampled.  There is no collective on the data path; the only communication is
the timing reduction of the benchmark (MAX over ranks) and an optional gather of per-graph edge counts.
"""
import torch


def shard_generations(num_generation, rank, world):
    """(gid_base, count) of rank's block: blocks differ by at most one graph, earlier ranks get the
    remainder.  Reference: `model.sample(num_generation)` is a single call (experiment.py:67)."""
    if not (0 <= rank < world):
        raise ValueError(f'rank {rank} outside world {world}')
    if num_generation < 0:
        raise ValueError('num_generation must be >= 0')
    q, r = divmod(num_generation, world)
    count = q + (1 if rank < r else 0)
    base = rank * q + min(rank, r)
    return base, count


def max_over_ranks(value, device='cpu'):
    """MAX of a python float over the process group (RCCL on GPUs, gloo on CPU); identity if not initialised."""
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device='cpu' if td.get_backend() == 'gloo' else device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def gather_edge_counts(edges_per_graph, device='cpu'):
    """All ranks' per-graph edge counts in global graph order (int64 [num_generation]); shards may differ
    in length by one, so counts are padded to the longest shard for the all_gather."""
    import torch.distributed as td
    e = torch.as_tensor(edges_per_graph, dtype=torch.int64, device=device)
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return e
    world = td.get_world_size()
    n = torch.tensor([e.numel()], dtype=torch.int64, device=device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    td.all_gather(ns, n)
    m = int(max(int(x.item()) for x in ns))
    pad = torch.zeros(m, dtype=torch.int64, device=device)
    pad[:e.numel()] = e
    outs = [torch.zeros_like(pad) for _ in range(world)]
    td.all_gather(outs, pad)
    return torch.cat([o[:int(k.item())] for o, k in zip(outs, ns)])


# ---- training (data-parallel replicas) ---------------------------------------------
def _dist():
    import torch.distributed as td
    return td if (td.is_available() and td.is_initialized() and td.get_world_size() > 1) else None


def sum_over_ranks(value, device='cpu'):
    """SUM of a python number over the process group (e.g. the global number of pairs P that
    normalises elbo_bpd, diffusion/loss.py:10-12)."""
    td = _dist()
    if td is None:
        return value
    t = torch.tensor([float(value)], dtype=torch.float64, device='cpu' if td.get_backend() == 'gloo' else device)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return float(t.item())


def sum_over_ranks_device(value, device):
    """As sum_over_ranks, but the result stays on the device (0-d fp32 tensor) when a process group is up: the
    training step divides by it without a host sync.  Without a group: the python number."""
    td = _dist()
    if td is None:
        return value
    t = torch.tensor([float(value)], dtype=torch.float64, device='cpu' if td.get_backend() == 'gloo' else device)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return t[0].to(torch.float32).to(device)


def _host_staged(x):
    """gloo moves host memory only: under it a device tensor is staged through the host (CPU rehearsals of the
    multi-rank paths on a one-GPU box); RCCL takes device tensors as they are."""
    import torch.distributed as td
    return x.is_cuda and td.get_backend() == 'gloo'


def gather_step_scalars(num_entries, t, kl):
    """ONE all-gather per training step for everything the replicas exchange besides the gradients: the local pair
    count (elbo_bpd normalises by the GLOBAL one, diffusion/loss.py:10-12) and every rank's (t, kl) for the importance
    history (diffusion_binomial_active.py:250-254).  Returns (global pair count as a 0-d fp32 tensor or the python
    number without a group, t of all ranks int64, kl of all ranks fp32), equal lengths per rank assumed."""
    td = _dist()
    if td is None:
        return num_entries, t, kl
    B = t.numel()
    # fp64 payload: t < 2^53 and pair counts up to 2^53 stay exact
    msg = torch.empty(1 + 2 * B, dtype=torch.float64, device=t.device)
    msg[0] = float(num_entries)
    msg[1:1 + B] = t.to(torch.float64)
    msg[1 + B:] = kl.detach().to(torch.float64)
    staged = _host_staged(msg)
    src = msg.cpu() if staged else msg
    outs = [torch.empty_like(src) for _ in range(td.get_world_size())]
    td.all_gather(outs, src)
    allm = torch.stack(outs)
    if staged:
        allm = allm.to(t.device)
    return allm[:, 0].sum().to(torch.float32), allm[:, 1:1 + B].reshape(-1).to(torch.int64), \
        allm[:, 1 + B:].reshape(-1).to(torch.float32)


def all_gather_cat(x):
    """Concatenation of every rank's 1-D tensor in rank order (equal lengths): the (t, kl) pairs every
    replica must apply to Lt_history / Lt_count so the importance sampler stays identical across ranks
    (diffusion_binomial_active.py:250-254)."""
    td = _dist()
    if td is None:
        return x
    staged = _host_staged(x)
    src = x.cpu() if staged else x.contiguous()
    outs = [torch.empty_like(src) for _ in range(td.get_world_size())]
    td.all_gather(outs, src)
    out = torch.cat(outs)
    return out.to(x.device) if staged else out


def allreduce_gradients(params, bucket=None):
    """ONE flat fp32 bucket (1 123 650 floats = 4.49 MB for the reference network) summed over ranks with a
    single all-reduce (RCCL over xGMI: latency-bound at this size, so one message, not one per tensor),
    then scattered back into the .grad tensors.  Parameters without a gradient contribute zeros so the
    bucket layout is identical on every rank.  Returns the bucket (reusable on the next step)."""
    td = _dist()
    ps = [p for p in params if p.requires_grad]
    if td is None or not ps:
        return bucket
    n = sum(p.numel() for p in ps)
    dev = ps[0].device
    if bucket is None or bucket.numel() != n or bucket.device != dev:
        bucket = torch.empty(n, dtype=torch.float32, device=dev)
    o = 0
    for p in ps:
        k = p.numel()
        if p.grad is None:
            bucket[o:o + k].zero_()
        else:
            bucket[o:o + k].copy_(p.grad.reshape(-1))
        o += k
    if _host_staged(bucket):
        hb = bucket.cpu()
        td.all_reduce(hb, op=td.ReduceOp.SUM)
        bucket.copy_(hb)
    else:
        td.all_reduce(bucket, op=td.ReduceOp.SUM)
    o = 0
    for p in ps:
        k = p.numel()
        if p.grad is None:
            p.grad = bucket[o:o + k].view_as(p).clone()
        else:
            p.grad.copy_(bucket[o:o + k].view_as(p))
        o += k
    return bucket


def allreduce_flat(flat):
    """SUM of a flat gradient arena over the ranks, in place (train.GradArena: p.grad are views into it, so
    there is no pack / unpack)."""
    td = _dist()
    if td is not None:
        if _host_staged(flat):
            h = flat.cpu()
            td.all_reduce(h, op=td.ReduceOp.SUM)
            flat.copy_(h)
        else:
            td.all_reduce(flat, op=td.ReduceOp.SUM)
    return flat
