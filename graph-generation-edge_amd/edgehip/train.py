"""train.py - the differentiable (training) form of TGNN_degree_guided.forward + the CE loss.

Each op is a HIP kernel pair (csrc/train.hip) behind a thin `torch.autograd.Function` (the
caller owns every buffer as a torch tensor and torch's autograd engine only ORDERS the backward calls
and sums gradients of tensors used twice).  The op sequence follows layers/layers.py:315-383 of the
reference, op for op and unfolded, so gradients land on the reference's own parameters.
"""
import ctypes as C

import torch
from torch.autograd import Function

from . import capi

byref = C.byref
lib = capi.lib
SILU, MISH = 1, 2


def _s():
    return capi.stream_ptr()


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _zeros(shape, like):
    return torch.zeros(shape, dtype=torch.float32, device=like.device)


class _Linear(Function):
    """torch.nn.Linear: y = x W^T + b  (fp32 MFMA forward, dX via the transposed weight, dW/db via the
    row-reduction MFMA kernel)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        M, K = x.shape
        N = w.shape[0]
        y = _empty((M, N), x)
        if M:
            capi.check(lib.edgehip_linear_fwd(x.data_ptr(), M, K, w.data_ptr(), b.data_ptr(), N, 0, y.data_ptr(), _s()),
                       'linear_fwd')
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = x.shape
        N = w.shape[0]
        dx = None
        dw, db = _zeros((N, K), x), _zeros((N,), x)
        if M:
            if ctx.needs_input_grad[0]:
                wt = _empty((K * N,), x)
                dx = _empty((M, K), x)
                capi.check(lib.edgehip_linear_bwd_input(dy.data_ptr(), M, N, w.data_ptr(), K, wt.data_ptr(), dx.data_ptr(),
                                                        _s()), 'linear_bwd_input')
            capi.check(lib.edgehip_linear_bwd_weight(dy.data_ptr(), x.data_ptr(), M, N, K, dw.data_ptr(), db.data_ptr(),
                                                     _s()), 'linear_bwd_weight')
        elif ctx.needs_input_grad[0]:
            dx = _zeros((M, K), x)
        return dx, dw, db


class _Act(Function):
    @staticmethod
    def forward(ctx, x, kind):
        x = x.contiguous()
        y = torch.empty_like(x)
        if x.numel():
            capi.check(lib.edgehip_act_fwd(kind, x.data_ptr(), x.numel(), y.data_ptr(), _s()), 'act_fwd')
        ctx.save_for_backward(x)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        if x.numel():
            capi.check(lib.edgehip_act_bwd(ctx.kind, x.data_ptr(), dy.data_ptr(), x.numel(), dx.data_ptr(), _s()), 'act_bwd')
        return dx, None


class _Embed(Function):
    """[embedding_t(d_t/dmax) | embedding_0(d0/dmax) | embedding_sel(active)]  layers.py:326-336."""

    @staticmethod
    def forward(ctx, wt, bt, w0, b0, wsel, L, deg, d0, flags, max_degree):
        X0 = _empty((L.num_nodes, 192), wt)
        capi.check(lib.edgehip_train_embed_fwd(byref(L), deg.data_ptr(), d0.data_ptr(), flags.data_ptr(), max_degree,
                                               wt.data_ptr(), bt.data_ptr(), w0.data_ptr(), b0.data_ptr(), wsel.data_ptr(),
                                               X0.data_ptr(), _s()), 'embed_fwd')
        ctx.args = (L, deg, d0, flags, max_degree)
        ctx.like = wt
        return X0

    @staticmethod
    def backward(ctx, dX0):
        L, deg, d0, flags, max_degree = ctx.args
        dX0 = dX0.contiguous()
        z = ctx.like
        dwt, dbt, dw0, db0, dwsel = _zeros((64, 1), z), _zeros((64,), z), _zeros((64, 1), z), _zeros((64,), z), _zeros((2, 64), z)
        capi.check(lib.edgehip_train_embed_bwd(byref(L), deg.data_ptr(), d0.data_ptr(), flags.data_ptr(), max_degree,
                                               dX0.data_ptr(), dwt.data_ptr(), dbt.data_ptr(), dw0.data_ptr(),
                                               db0.data_ptr(), dwsel.data_ptr(), _s()), 'embed_bwd')
        return dwt, dbt, dw0, db0, dwsel, None, None, None, None, None


class _Assemble(Function):
    """out[i] = [ h[i] + ctx[g(i)] | tail_graph[g(i)] or tail_rows[i] ]."""

    @staticmethod
    def forward(ctx, h, cx, tg, tr, L):
        h = h.contiguous()
        Kt = 64 if tg is not None else (tr.shape[1] if tr is not None else 0)
        out = _empty((L.num_nodes, 64 + Kt), h)
        capi.check(lib.edgehip_assemble_fwd(byref(L), h.data_ptr(), capi.ptr(cx), capi.ptr(tg), capi.ptr(tr), Kt,
                                            out.data_ptr(), _s()), 'assemble_fwd')
        ctx.L, ctx.Kt = L, Kt
        ctx.has = (cx is not None, tg is not None, tr is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        L, Kt = ctx.L, ctx.Kt
        dout = dout.contiguous()
        dh = _empty((L.num_nodes, 64), dout)
        dcx = _empty((L.num_graphs, 64), dout) if ctx.has[0] else None
        dtg = _empty((L.num_graphs, 64), dout) if ctx.has[1] else None
        dtr = _empty((L.num_nodes, Kt), dout) if ctx.has[2] else None
        capi.check(lib.edgehip_assemble_bwd(byref(L), dout.data_ptr(), Kt, dh.data_ptr(), capi.ptr(dcx), capi.ptr(dtg),
                                            capi.ptr(dtr), _s()), 'assemble_bwd')
        return dh, dcx, dtg, dtr, None


class _SegmentMean(Function):
    """scatter(mean) over `batch` (+ optional per-graph addend)  layers.py:342,367."""

    @staticmethod
    def forward(ctx, x, add, L):
        x = x.contiguous()
        out = _empty((L.num_graphs, 64), x)
        capi.check(lib.edgehip_segment_mean_fwd(byref(L), x.data_ptr(), capi.ptr(add), out.data_ptr(), _s()), 'segment_mean')
        ctx.L = L
        ctx.has_add = add is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        L = ctx.L
        dout = dout.contiguous()
        dx = _empty((L.num_nodes, 64), dout)
        capi.check(lib.edgehip_segment_mean_bwd(byref(L), dout.data_ptr(), dx.data_ptr(), _s()), 'segment_mean_bwd')
        return dx, (dout if ctx.has_add else None), None


class _Attention(Function):
    """TransformerConv message passing over the bit-matrix neighbourhoods (layers.py:354)."""

    @staticmethod
    def forward(ctx, q, k, v, skip, L, adj, heads):
        q, k, v, skip = q.contiguous(), k.contiguous(), v.contiguous(), skip.contiguous()
        Nn = L.num_nodes
        out, O, lse = _empty((Nn, 64), q), _empty((Nn, heads * 64), q), _empty((Nn, heads), q)
        capi.check(lib.edgehip_attn_fwd(byref(L), adj.data_ptr(), heads, q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                        skip.data_ptr(), out.data_ptr(), O.data_ptr(), lse.data_ptr(), _s()), 'attn_fwd')
        ctx.save_for_backward(q, k, v, O, lse)
        ctx.args = (L, adj, heads)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, O, lse = ctx.saved_tensors
        L, adj, heads = ctx.args
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        D = torch.empty_like(lse)
        capi.check(lib.edgehip_attn_bwd(byref(L), adj.data_ptr(), heads, q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                        O.data_ptr(), lse.data_ptr(), dout.data_ptr(), D.data_ptr(), dq.data_ptr(),
                                        dk.data_ptr(), dv.data_ptr(), _s()), 'attn_bwd')
        return dq, dk, dv, dout, None, None, None


class _GruGate(Function):
    @staticmethod
    def forward(ctx, gi, gh, h):
        gi, gh, h = gi.contiguous(), gh.contiguous(), h.contiguous()
        out = torch.empty_like(h)
        if h.shape[0]:
            capi.check(lib.edgehip_gru_gate_fwd(gi.data_ptr(), gh.data_ptr(), h.data_ptr(), h.shape[0], out.data_ptr(), _s()),
                       'gru_gate_fwd')
        ctx.save_for_backward(gi, gh, h)
        return out

    @staticmethod
    def backward(ctx, dout):
        gi, gh, h = ctx.saved_tensors
        dout = dout.contiguous()
        dgi, dgh, dh = torch.empty_like(gi), torch.empty_like(gh), torch.empty_like(h)
        if h.shape[0]:
            capi.check(lib.edgehip_gru_gate_bwd(gi.data_ptr(), gh.data_ptr(), h.data_ptr(), dout.data_ptr(), h.shape[0],
                                                dgi.data_ptr(), dgh.data_ptr(), dh.data_ptr(), _s()), 'gru_gate_bwd')
        return dgi, dgh, dh


class _PairGatherAdd(Function):
    @staticmethod
    def forward(ctx, z, row, col):
        z = z.contiguous()
        A, W = row.numel(), z.shape[1]
        e = _empty((A, W), z)
        if A:
            capi.check(lib.edgehip_pair_gather_add_fwd(z.data_ptr(), W, row.data_ptr(), col.data_ptr(), A, e.data_ptr(), _s()),
                       'pair_gather_add_fwd')
        ctx.args = (row, col, z.shape)
        return e

    @staticmethod
    def backward(ctx, de):
        row, col, shape = ctx.args
        de = de.contiguous()
        dz = _zeros(shape, de)
        if row.numel():
            capi.check(lib.edgehip_pair_gather_add_bwd(de.data_ptr(), shape[1], row.data_ptr(), col.data_ptr(), row.numel(),
                                                       dz.data_ptr(), _s()), 'pair_gather_add_bwd')
        return dz, None, None


class _EdgeCE(Function):
    """final_out[4] + log_softmax + CE + per-graph sum (layers.py:381, active.py:157-175)."""

    @staticmethod
    def forward(ctx, e2, W4, b4, tgt, pair_graph, B):
        e2 = e2.contiguous()
        A = e2.shape[0]
        logits, ce = _empty((A, 2), e2), _zeros((B,), e2)
        if A:
            capi.check(lib.edgehip_edge_ce_fwd(e2.data_ptr(), W4.data_ptr(), b4.data_ptr(), tgt.data_ptr(),
                                               pair_graph.data_ptr(), A, logits.data_ptr(), ce.data_ptr(), _s()), 'edge_ce_fwd')
        ctx.save_for_backward(e2, W4, b4)
        ctx.args = (tgt, pair_graph)
        ctx.mark_non_differentiable(logits)
        return ce, logits

    @staticmethod
    def backward(ctx, gce, _glogits):
        e2, W4, b4 = ctx.saved_tensors
        tgt, pair_graph = ctx.args
        gce = gce.contiguous()
        A = e2.shape[0]
        de2, dW4, db4 = torch.empty_like(e2), torch.zeros_like(W4), torch.zeros_like(b4)
        if A:
            capi.check(lib.edgehip_edge_ce_bwd(e2.data_ptr(), W4.data_ptr(), b4.data_ptr(), tgt.data_ptr(),
                                               pair_graph.data_ptr(), gce.data_ptr(), A, de2.data_ptr(), dW4.data_ptr(),
                                               db4.data_ptr(), _s()), 'edge_ce_bwd')
        return de2, dW4, db4, None, None, None


class _Dropout(Function):
    """torch.nn.Dropout in training mode (layers.py:357,361); the backward is the same masked scaling."""

    @staticmethod
    def forward(ctx, x, L, dp, slot):
        x = x.contiguous()
        y = torch.empty_like(x)
        capi.check(lib.edgehip_dropout_fwd(byref(L), x.data_ptr(), byref(dp), slot, y.data_ptr(), _s()), 'dropout_fwd')
        ctx.args = (L, dp, slot)
        return y

    @staticmethod
    def backward(ctx, dy):
        L, dp, slot = ctx.args
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        capi.check(lib.edgehip_dropout_fwd(byref(L), dy.data_ptr(), byref(dp), slot, dx.data_ptr(), _s()), 'dropout_bwd')
        return dx, None, None, None


def linear(x, m):
    return _Linear.apply(x, m.weight, m.bias)


def act(x, kind=SILU):
    return _Act.apply(x, kind)


def _mlp2(x, seq):
    return linear(act(linear(x, seq[0])), seq[2])


def pair_targets(batch, adj_tm1, row, col):
    A = row.numel()
    tgt = torch.empty(A, dtype=torch.uint8, device=row.device)
    pg = torch.empty(A, dtype=torch.int32, device=row.device)
    if A:
        capi.check(lib.edgehip_pair_targets(byref(batch.layout), adj_tm1.data_ptr(), row.data_ptr(), col.data_ptr(), A,
                                            tgt.data_ptr(), pg.data_ptr(), _s()), 'pair_targets')
    return tgt, pg


def tgnn_ce_forward(net, batch, adj_t, deg_t, flags, t_graph, row, col, tgt, pair_graph, dp=None):
    """Differentiable TGNN_degree_guided.forward on the active pairs (row, col) followed by the per-graph
    cross-entropy against `tgt`.  Returns (ce_graph [B] with grad, logits [A,2])."""
    L = batch.layout
    X0 = _Embed.apply(net.embedding_t.weight, net.embedding_t.bias, net.embedding_0.weight, net.embedding_0.bias,
                      net.embedding_sel.weight, L, deg_t, batch.degree_i32, flags, int(net.max_degree))
    nodes0 = act(linear(X0, net.node_in[0]))                                   # layers.py:335-336
    pe = net._pos_emb.index_select(0, t_graph.long())                          # table lookup, no grad
    mt = act(_mlp2(pe, net.mlp), MISH)                                         # layers.py:338-339, :31-34
    h = hid = nodes0                         # h: GRU output `nodes`; hid: the hidden state carried on (dropped, :361)
    drop = dp is not None and dp.p > 0
    cx = _mlp2(_SegmentMean.apply(nodes0, None, L), net.global_mlp)            # layers.py:342-343
    for i, heads in enumerate(net.num_heads):
        tl = linear(mt, net.layers[f'time{i}'].linear)
        x = _Assemble.apply(h, cx if i > 0 else None, tl, None, L)             # cat[nodes, t_emb], nodes = h + ctx
        cv = net.layers[f'conv{i}']
        conv = _Attention.apply(linear(x, cv.lin_query), linear(x, cv.lin_key), linear(x, cv.lin_value),
                                linear(x, cv.lin_skip), L, adj_t, heads)
        xs = act(conv)
        if drop:
            xs = _Dropout.apply(xs, L, dp, 2 * i)                              # :357
        gi = _Linear.apply(xs, net.gru.weight_ih_l0, net.gru.bias_ih_l0)
        gh = _Linear.apply(hid, net.gru.weight_hh_l0, net.gru.bias_hh_l0)
        h = _GruGate.apply(gi, gh, hid)
        hid = _Dropout.apply(h, L, dp, 2 * i + 1) if drop and i + 1 < len(net.num_heads) else h   # :361
        cin = _Assemble.apply(h, None, cx, None, L)                            # cat[nodes, contexts]  :366
        nctx = _mlp2(cin, net.context_mlp)
        cx = _mlp2(_SegmentMean.apply(nctx, cx, L), net.global_mlp)            # :367-368
    zin = _Assemble.apply(h, cx, None, X0, L)                                  # :371, :377
    no = net.node_out_mlp
    z = linear(act(linear(act(linear(zin, no[0])), no[2])), no[4])
    e = _PairGatherAdd.apply(z, row, col)                                      # :380
    fo = net.final_out
    e2 = act(linear(act(linear(e, fo[0])), fo[2]))
    return _EdgeCE.apply(e2, fo[4].weight, fo[4].bias, tgt, pair_graph, batch.num_graphs)


# ------------------------------------------------------------------------------------------------------
# The same network as ONE autograd node: edgehip_train_forward / edgehip_train_backward queue every op
# above from the library (csrc/train.hip section J).  `tgnn_ce_forward` (one node per op) stays as the
# checkable form; tests compare the two.
# ------------------------------------------------------------------------------------------------------
class GradArena:
    """One flat fp32 buffer holding the gradient of every parameter; p.grad are views into it.  The backward
    call ADDS into it directly (nothing is returned through autograd), the data-parallel exchange all-reduces
    it in place, and zeroing it is one memset."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        self._by_ptr = {p.data_ptr(): v for p, v in zip(self.params, self.views)}

    def matches(self, params):
        ps = [p for p in params if p.requires_grad]
        return len(ps) == len(self.params) and all(a.data_ptr() == b.data_ptr() for a, b in zip(ps, self.params))

    def zero_and_attach(self):
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v

    def grad_of(self, p):
        return self._by_ptr[p.data_ptr()]


_ACTIVE_ARENA = [None]


class use_arena:
    """with use_arena(arena): backward of the fused node accumulates straight into the arena."""

    def __init__(self, arena):
        self.arena = arena

    def __enter__(self):
        self.prev = _ACTIVE_ARENA[0]
        _ACTIVE_ARENA[0] = self.arena
        return self.arena

    def __exit__(self, *exc):
        _ACTIVE_ARENA[0] = self.prev


def _take_workspace(batch, L, P, cap, dev):
    """The batch keeps ONE workspace for reuse across steps; a forward that finds it busy (a second loss alive before
    the first one's backward) gets a workspace of its own."""
    nbytes = lib.edgehip_train_workspace_bytes(byref(L), byref(P), cap)
    if nbytes == 0:
        raise capi.EdgeHipError('edgehip_train_workspace_bytes: unsupported configuration')
    ws = getattr(batch, '_train_ws', None)
    if getattr(batch, '_train_ws_busy', False):
        return torch.empty(nbytes + 4096, dtype=torch.uint8, device=dev)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes + 4096, dtype=torch.uint8, device=dev)
        batch._train_ws = ws
    batch._train_ws_busy = True
    return ws


def _fwd_nodes(L, P, adj_t, deg_t, d0, flags, t_graph, cap, ws, dp):
    if dp is not None and dp.p > 0:
        capi.check(lib.edgehip_train_forward_nodes_dp(byref(L), byref(P), adj_t.data_ptr(), deg_t.data_ptr(), d0.data_ptr(),
                                                      flags.data_ptr(), t_graph.data_ptr(), cap, byref(dp), ws.data_ptr(),
                                                      ws.numel(), _s()), 'train_forward_nodes_dp')
    else:
        capi.check(lib.edgehip_train_forward_nodes(byref(L), byref(P), adj_t.data_ptr(), deg_t.data_ptr(), d0.data_ptr(),
                                                   flags.data_ptr(), t_graph.data_ptr(), cap, ws.data_ptr(), ws.numel(),
                                                   _s()), 'train_forward_nodes')


def prelaunch_nodes(net, batch, adj_t, deg_t, flags, t_graph, dp=None):
    """Queues the node part of the fused forward NOW, before the caller has read the number of active pairs back
    from the device (the step's only host sync): the device works through it while the host enumerates the pairs.
    Needs the pair capacity of an earlier step on this batch; returns False (nothing queued) without one."""
    cap = getattr(batch, '_train_cap', 0)
    if cap <= 0 or getattr(batch, '_train_ws_busy', False) or '_train_pre' in batch.__dict__:
        return False
    if not all(p.requires_grad for p in net.parameters()):
        return False
    L, P = batch.layout, net.raw_params()
    ws = _take_workspace(batch, L, P, cap, adj_t.device)
    _fwd_nodes(L, P, adj_t, deg_t, batch.degree_i32, flags, t_graph, cap, ws, dp)
    batch._train_pre = dict(ws=ws, cap=cap, adj=adj_t.data_ptr())
    return True


class _FusedNet(Function):
    @staticmethod
    def forward(ctx, net, batch, adj_t, deg_t, flags, t_graph, row, col, tgt, pair_graph, dp, arena_params, *inputs):
        # arena_params: the parameters when their gradients go straight into the active GradArena; `inputs` is then a
        # single anchor tensor (autograd only has to call backward: an apply() over 86 parameter inputs costs ~0.25 ms
        # of host time each way, right after the step's only host sync).  Otherwise `inputs` are the parameters.
        params = arena_params if arena_params is not None else inputs
        L = batch.layout
        P = net.raw_params()
        A, B = row.numel(), batch.num_graphs
        dev = adj_t.device
        d0 = batch.degree_i32
        ce = torch.empty(B, dtype=torch.float32, device=dev)
        logits = torch.empty((A, 2), dtype=torch.float32, device=dev)
        pre = batch.__dict__.pop('_train_pre', None)
        if pre is not None and (A > pre['cap'] or pre['adj'] != adj_t.data_ptr()):
            if pre['ws'] is getattr(batch, '_train_ws', None):
                batch._train_ws_busy = False              # the node part queued ahead does not fit this step: redo it
            pre = None
        if pre is not None:
            ws, cap = pre['ws'], pre['cap']
        else:
            # the workspace layout is fixed by a CAPACITY of active pairs that only grows, so that the node part of the
            # next step can be queued (prelaunch_nodes) before that step's pair count is known
            cap = getattr(batch, '_train_cap', 0)
            if A > cap:
                cap = batch._train_cap = int(A * 1.3) + 4096
            ws = _take_workspace(batch, L, P, cap, dev)
            _fwd_nodes(L, P, adj_t, deg_t, d0, flags, t_graph, cap, ws, dp)
        capi.check(lib.edgehip_train_forward_pairs(byref(L), byref(P), capi.ptr(row), capi.ptr(col), A, cap, capi.ptr(tgt),
                                                   capi.ptr(pair_graph), ce.data_ptr(), logits.data_ptr(), ws.data_ptr(),
                                                   ws.numel(), _s()), 'train_forward_pairs')
        ctx.cap = cap
        ctx.dp = dp
        ctx.args = (net, batch, P, adj_t, deg_t, d0, flags, row, col, tgt, pair_graph, ws, params, arena_params is not None)
        ctx.mark_non_differentiable(logits)
        return ce, logits

    @staticmethod
    def backward(ctx, gce, _glogits):
        net, batch, P, adj_t, deg_t, d0, flags, row, col, tgt, pair_graph, ws, params, anchored = ctx.args
        L = batch.layout
        arena = _ACTIVE_ARENA[0]
        direct = arena is not None and arena.matches(params)
        if anchored and not direct:
            raise capi.EdgeHipError('the gradient arena of this forward is no longer active (backward outside use_arena)')
        if direct:
            grads = None
            address = lambda p: arena.grad_of(p).data_ptr()
        else:
            flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=adj_t.device)
            grads, o, by_ptr = [], 0, {}
            for p in params:
                g = flat[o:o + p.numel()].view_as(p)
                grads.append(g)
                by_ptr[p.data_ptr()] = g
                o += p.numel()
            address = lambda p: by_ptr[p.data_ptr()].data_ptr()
        G = capi.GnnParams()
        net.fill_struct(G, address)
        gce = gce.contiguous()
        dp = ctx.dp
        capi.use_pool_for_training(adj_t.device)
        capi.check(lib.edgehip_train_backward_dp(byref(L), byref(P), byref(G), adj_t.data_ptr(), deg_t.data_ptr(),
                                                 d0.data_ptr(), flags.data_ptr(), capi.ptr(row), capi.ptr(col), row.numel(),
                                                 ctx.cap, capi.ptr(tgt), capi.ptr(pair_graph), gce.data_ptr(),
                                                 byref(dp) if dp is not None else None, ws.data_ptr(),
                                                 ws.numel(), _s()), 'train_backward')
        if ws is getattr(batch, '_train_ws', None):
            batch._train_ws_busy = False
        head = (None,) * 12
        if direct:
            return head + (None,) * (1 if anchored else len(params))
        used = {prm.data_ptr() for _, _, prm in net.param_slots()}
        return head + tuple(g if p.data_ptr() in used else None for p, g in zip(params, grads))


def tgnn_ce_fused(net, batch, adj_t, deg_t, flags, t_graph, row, col, tgt, pair_graph, dp=None):
    """Same contract as tgnn_ce_forward, one autograd node."""
    params = [p for p in net.parameters() if p.requires_grad]
    if len(params) != sum(1 for _ in net.parameters()):
        # frozen parameters: the op-by-op form lets autograd skip them
        return tgnn_ce_forward(net, batch, adj_t, deg_t, flags, t_graph, row, col, tgt, pair_graph, dp)
    arena = _ACTIVE_ARENA[0]
    if arena is not None and arena.matches(params):
        anchor = getattr(net, '_grad_anchor', None)
        if anchor is None or anchor.device != adj_t.device:
            anchor = net._grad_anchor = torch.zeros((), device=adj_t.device, requires_grad=True)
        return _FusedNet.apply(net, batch, adj_t, deg_t, flags, t_graph.contiguous(), row, col, tgt, pair_graph, dp,
                               tuple(params), anchor)
    return _FusedNet.apply(net, batch, adj_t, deg_t, flags, t_graph.contiguous(), row, col, tgt, pair_graph, dp, None, *params)
