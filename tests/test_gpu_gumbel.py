"""GPU: the arithmetic Gumbel transform + patch list (include/edgehip.h "a3") against the host's own table.

Every random decision of the path is g(u1) + lp1 > g(u0) + lp0 with g(u) = -log(-log(u + 1e-30) + 1e-30)
(diffusion_base.py:175-180) in the HOST's arithmetic.  The device evaluates g itself and carries a list of the grid
points where its exact form differs from the host's.  Checked here, exhaustively where the claim is exhaustive:
  * the blob reproduces the host's g at ALL 2^24 grid points (exact form + patches == tables.gumbel_lut());
  * the two compiled copies of the transforms (state.hip / gnn.hip kernels) agree bit for bit on the whole grid;
  * eps_fast in the blob bounds |fast - host| on the whole grid;
  * decisions through the C ABI (edgehip_active_select_sample with a table that gives every node its own log-probs)
    equal the host's on engineered ties and near-ties (margins of 0, +-1, +-2 ulps), for every patched grid point
    and for random ones - the cases where an unpatched device logarithm decides differently.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
N24 = 1 << 24


def _blob():
    from edgehip import model, tables
    blob = model._gumbel_lut(torch.device('cuda'))
    w = blob.view(torch.int32)
    n = int(w[1].item())
    assert int(w[0].item()) == model.GUMBEL_MAGIC and blob.numel() == 4 + 2 * n
    keys = w[4:4 + n].long()
    vals = blob[4 + n:]
    eps = float(blob[2].item())
    return tables.gumbel_lut(), keys, vals, eps


def test_blob_reproduces_the_host_table_on_the_whole_grid():
    from edgehip import capi
    host, keys, vals, eps = _blob()
    hd = host.cuda()
    grids = []
    for which in (0, 1):
        ex, fa = torch.empty_like(hd), torch.empty_like(hd)
        capi.check(capi.lib.edgehip_gumbel_grid(ex.data_ptr(), fa.data_ptr(), 0, N24, which, capi.stream_ptr()), 'grid')
        grids.append((ex, fa))
    assert torch.equal(grids[0][0], grids[1][0]) and torch.equal(grids[0][1], grids[1][1])   # both compiled copies
    ex, fa = grids[0]
    n = int(keys.numel())
    assert n == int((ex != hd).sum().item())
    # how many depends on the HOST's math library: 15 862 where torch's MKL runs its Intel path (correctly rounded
    # almost everywhere), 2 380 501 on the GPU boxes' AMD EPYC hosts (MKL's generic path: 7.5 % of its fp32
    # logarithms are one ulp off the correctly rounded value).  Either way the list is exhaustive.
    assert n < N24 // 4, n
    assert bool((keys[1:] > keys[:-1]).all()) if n > 1 else True
    patched = ex.clone()
    patched[keys] = vals
    assert torch.equal(patched, hd)                                                          # ALL 2^24 grid points
    worst = float((fa - hd).abs().max().item())
    assert worst <= eps < 1e-4, (worst, eps)
    # a sub-range call addresses the same grid
    sub = torch.empty(1000, dtype=torch.float32, device='cuda')
    capi.check(capi.lib.edgehip_gumbel_grid(sub.data_ptr(), None, 12345, 1000, 0, capi.stream_ptr()), 'grid')
    assert torch.equal(sub, ex[12345:13345])
    assert capi.lib.edgehip_gumbel_grid(sub.data_ptr(), None, N24 - 10, 11, 0, None) == capi.ERR_ARG
    print(f'gumbel patch: {n} of {N24} grid points differ from the host, eps_fast {eps:.3e}')


def _decide_on_device(k0, k1, lp0, lp1, patch):
    """flags of edgehip_active_select_sample for one case per node: node i draws (u0, u1) = (k0, k1)[i] 2^-24 and reads
    its own log-probs (lp0, lp1)[i] from a table with one row per node (d0[i] - deg[i] = i selects row i)."""
    import edgehip
    from edgehip import capi
    M = len(k0)
    B, per = 16, (M + 15) // 16
    b = edgehip.GraphBatch([per] * B, degree=[np.zeros(per)] * B, device='cuda')
    Nn = b.num_nodes
    pad = Nn - M

    def padded(x, fill):
        return torch.cat([x, torch.full((pad,), fill, dtype=x.dtype)]) if pad else x
    u = torch.stack([padded(k0, 0).float() * 2.0 ** -24, padded(k1, 0).float() * 2.0 ** -24], 1).contiguous().cuda()
    post = torch.stack([padded(lp0, 0.0), padded(lp1, 0.0)], 1).contiguous().cuda()          # [1][Nn][2]
    d0 = torch.arange(Nn, dtype=torch.int32, device='cuda')
    deg = torch.zeros(Nn + 32, dtype=torch.int32, device='cuda')
    tg = torch.zeros(B, dtype=torch.int32, device='cuda')
    flags = torch.full((Nn + 32,), 7, dtype=torch.uint8, device='cuda')
    act = torch.zeros(Nn + 32, dtype=torch.int32, device='cuda')
    cnt = torch.zeros(B + 1, dtype=torch.int32, device='cuda')
    capi.check(capi.lib.edgehip_active_select_sample(C.byref(b.layout), d0.data_ptr(), deg.data_ptr(), tg.data_ptr(),
                                                     post.data_ptr(), Nn - 1, u.data_ptr(), None, capi.ptr(patch),
                                                     flags.data_ptr(), act.data_ptr(), cnt.data_ptr(), capi.stream_ptr()),
               'active_select_sample')
    f = flags[:M].cpu()
    assert int(cnt[:B].sum().item()) == int(flags[:Nn].sum().item())
    return f


def _ulp_steps(x, j):
    """x moved by j ulps (fp32)."""
    i = x.view(torch.int32).clone()
    neg = x < 0
    i = torch.where(neg, i - j, i + j)
    return i.view(torch.float32)


def test_decisions_equal_the_hosts_on_engineered_near_ties():
    from edgehip import model
    host, keys, vals, eps = _blob()
    blob = model._gumbel_lut(torch.device('cuda'))
    gen = torch.Generator().manual_seed(5)
    keys_c = keys.cpu()
    n = int(keys_c.numel())
    # every patched grid point as u1 (all of them when there are few, a sample otherwise), plus random grid points
    pk = keys_c if n <= 40_000 else keys_c[torch.randperm(n, generator=gen)[:40_000]]
    rk = torch.randint(0, N24, (60_000,), generator=gen)
    k1 = torch.cat([pk, rk])
    k0 = torch.randint(0, N24, (k1.numel(),), generator=gen)
    # half of the cases: the OTHER side is a patched point too
    if n:
        sel = torch.rand(k0.numel(), generator=gen) < 0.5
        k0 = torch.where(sel, keys_c[torch.randint(0, n, (k0.numel(),), generator=gen)], k0)
    total, differ_unpatched = 0, 0
    for lp0_kind in ('zero', 'random', 'large'):
        if lp0_kind == 'zero':
            lp0 = torch.zeros(k0.numel())
        elif lp0_kind == 'random':
            lp0 = -torch.rand(k0.numel(), generator=gen) * 5
        else:
            lp0 = torch.full((k0.numel(),), -92.10340118408203)          # log(1e-40): the reference's "cannot change"
        # lp1 that makes the HOST's sums tie, then moved by j ulps
        tie = (host[k0] + lp0) - host[k1]
        for j in (-2, -1, 0, 1, 2):
            lp1 = _ulp_steps(tie, j)
            want = ((host[k1] + lp1) > (host[k0] + lp0)).to(torch.uint8)
            got = _decide_on_device(k0, k1, lp0, lp1, blob)
            bad = (got != want).nonzero().flatten()
            assert bad.numel() == 0, (lp0_kind, j, int(bad.numel()), k0[bad[:4]].tolist(), k1[bad[:4]].tolist())
            total += int(want.numel())
            if lp0_kind == 'zero':
                differ_unpatched += int((_decide_on_device(k0, k1, lp0, lp1, None) != want).sum())
    # without the blob (device logarithms only) some of these near-ties do come out the other way: the cases are real
    assert n == 0 or differ_unpatched > 0
    print(f'{total} engineered near-tie decisions equal the host; without the patch list {differ_unpatched} of '
          f'{5 * k0.numel()} differ')


def test_decisions_on_ordinary_margins_and_at_the_ends_of_the_grid():
    from edgehip import model
    host, keys, vals, eps = _blob()
    blob = model._gumbel_lut(torch.device('cuda'))
    gen = torch.Generator().manual_seed(6)
    M = 100_000
    k0, k1 = torch.randint(0, N24, (M,), generator=gen), torch.randint(0, N24, (M,), generator=gen)
    lp0, lp1 = -torch.rand(M, generator=gen) * 8, -torch.rand(M, generator=gen) * 8
    want = ((host[k1] + lp1) > (host[k0] + lp0)).to(torch.uint8)
    assert torch.equal(_decide_on_device(k0, k1, lp0, lp1, blob), want)
    # the extremes of the grid: u = 0 (g = -4.2352) and u = 1 - 2^-24 (g = 16.6355)
    ke = torch.tensor([0, 0, N24 - 1, N24 - 1, 1, N24 - 2])
    kf = torch.tensor([N24 - 1, 0, 0, N24 - 1, N24 - 2, 1])
    z = torch.zeros(6)
    want = ((host[kf] + z) > (host[ke] + z)).to(torch.uint8)
    assert torch.equal(_decide_on_device(ke, kf, z, z, blob), want) and want.tolist() == [1, 0, 0, 0, 1, 0]
