"""Plugin surface of the hot path: `get_model(args, initial_graph_sampler)` with the reference's
flag strings (model.py:11-77 of the reference), returning a `BinomialDiffusionActive` whose
`log_prob(batch)` / `sample(n)` run on libedgehip's HIP kernels.

Module / parameter / buffer names follow the reference's state_dict (SURVEY.md App. C and
section 5 "checkpoint"), so `load_state_dict` of a reference checkpoint works:
  `_denoise_fn.embedding_t.weight`, `_denoise_fn.layers.conv0.lin_key.weight`, ...,
  `log_alpha`, `log_1_min_alpha`, `log_cumprod_alpha`, `log_1_min_cumprod_alpha`,
  `log_final_prob_node`, `log_final_prob_edge`, `Lt_history`, `Lt_count`.
The torch modules only HOLD parameters; no torch op touches the data path.
"""
import ctypes as C

import numpy as np
import torch
from torch import nn

from . import capi, tables
from .batch import GraphBatch

byref = C.byref


def _on_batch_device(fn):
    """Run an entry point with the batch's GPU current: the library launches on `torch.cuda.current_stream()` and keeps
    per-device state by hipGetDevice, so a batch on cuda:1 called while cuda:0 is current would put kernels on device
    0's stream against device 1's pointers.  The batch is the first positional argument after self."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, batch, *a, **k):
        dev = getattr(batch, 'device', None)
        if dev is not None and torch.device(dev).type == 'cuda':
            with torch.cuda.device(dev):
                return fn(self, batch, *a, **k)
        return fn(self, batch, *a, **k)
    return wrapped


def add_model_args(parser):
    """Same flags and defaults as the reference (model.py:11-24)."""
    parser.add_argument('--loss_type', type=str, default='vb_kl')
    parser.add_argument('--diffusion_steps', type=int, default=1000)
    parser.add_argument('--diffusion_dim', type=int, default=64)
    parser.add_argument('--dp_rate', type=float, default=0.)
    parser.add_argument('--num_heads', type=int, nargs="*", default=[8, 8, 8, 8, 1])
    parser.add_argument('--final_prob_node', type=float, nargs="*", default=None)
    parser.add_argument('--final_prob_edge', type=float, nargs="*", default=[1 - 1e-12, 1e-12])
    parser.add_argument('--parametrization', type=str, default='x0')
    parser.add_argument('--sample_time_method', type=str, default='importance')
    parser.add_argument('--arch', type=str, help='GAT | TGNN')
    parser.add_argument('--noise_schedule', type=str, default='cosine', help='cosine | linear')
    parser.add_argument('--norm', type=str, default='None', help='None | BN')


def get_model_id(args):
    return 'multinomial_diffusion'


def get_model(args, initial_graph_sampler):
    """Drop-in for the reference's factory (model.py:29-77); same asserts and error types."""
    if args.final_prob_node is not None:
        assert sum(args.final_prob_node) == 1
        assert len(args.final_prob_node) == args.num_node_classes
    assert sum(args.final_prob_edge) == 1
    assert len(args.final_prob_edge) == args.num_edge_classes
    if args.arch == 'TGNN':
        assert args.parametrization in ('x0', 'xt')
        raise NotImplementedError("arch 'TGNN' (non-degree-guided path) is outside the accelerated hot path; "
                                  "use --arch TGNN_degree_guided")
    elif args.arch == 'TGNN_degree_guided':
        assert args.parametrization in ('xt_prescribed_st')
    else:
        raise NotImplementedError()
    dynamics = TGNN_degree_guided(
        max_degree=args.max_degree,
        num_node_classes=2 if args.num_node_classes is None else args.num_node_classes,
        num_edge_classes=args.num_node_classes, dim=args.diffusion_dim, num_steps=args.diffusion_steps,
        num_heads=args.num_heads, dropout=args.dp_rate, norm=getattr(args, 'norm', 'None'), gru=True,
        degree=getattr(args, 'degree', True), augmented_features=getattr(args, 'augmented_feature_dict', {}),
        return_node_class=getattr(args, 'has_node_feature', False))
    if args.noise_schedule == 'cosine':
        noise_schedule = tables.cosine_beta_schedule
    elif args.noise_schedule == 'linear':
        noise_schedule = tables.linear_beta_schedule
    elif args.noise_schedule == 'Tt1':
        noise_schedule = tables.Tt1_beta_schedule
    else:
        raise NotImplementedError()
    return BinomialDiffusionActive(
        args.num_node_classes, args.num_edge_classes, initial_graph_sampler, dynamics,
        timesteps=args.diffusion_steps, loss_type=args.loss_type, final_prob_node=args.final_prob_node,
        final_prob_edge=args.final_prob_edge, parametrization=args.parametrization,
        sample_time_method=args.sample_time_method, noise_schedule=noise_schedule, device=args.device)


# ------------------------------------------------------------------------------------------
class _TimeEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.linear = nn.Linear(dim, dim)


class _Conv(nn.Module):
    """Parameter holder named like pyg.nn.TransformerConv(2*dim, dim, heads, concat=False)."""

    def __init__(self, dim, heads):
        super().__init__()
        self.lin_key = nn.Linear(2 * dim, heads * dim)
        self.lin_query = nn.Linear(2 * dim, heads * dim)
        self.lin_value = nn.Linear(2 * dim, heads * dim)
        self.lin_skip = nn.Linear(2 * dim, dim)


def _mlp(din, dhid, dout):
    return nn.Sequential(nn.Linear(din, dhid), nn.SiLU(), nn.Linear(dhid, dout))


class TGNN_degree_guided(nn.Module):
    """Parameters of the reference network (layers/layers.py:250-313); forward runs in HIP."""

    def __init__(self, max_degree, num_node_classes, num_edge_classes, dim, num_steps, num_heads=[4, 4, 4, 1],
                 dropout=0., norm='None', degree=False, augmented_features={}, **kwargs):
        super().__init__()
        if dim != 64:
            raise NotImplementedError('edgehip kernels are built for --diffusion_dim 64')
        if norm != 'None':
            raise NotImplementedError("only --norm None (the reference default) is implemented")
        if len(num_heads) > capi.MAX_LAYERS or any(h not in (1, 2, 4, 8) for h in num_heads):
            raise NotImplementedError('num_heads entries must be in {1,2,4,8}, at most 8 layers')
        if num_edge_classes != 2:
            raise NotImplementedError('the kernels implement the binomial (2 edge classes) diffusion only')
        if not kwargs.get('gru', False):
            raise NotImplementedError('the reference factory always builds the GRU variant (model.py:56)')
        self.max_degree, self.num_classes, self.num_heads = max_degree, num_edge_classes, list(num_heads)
        self.dim, self.num_steps, self.dropout_p = dim, num_steps, dropout
        self.embedding_t = nn.Linear(1, dim)
        self.embedding_0 = nn.Linear(1, dim)
        self.embedding_sel = nn.Embedding(2, dim)
        self.node_in = nn.Sequential(nn.Linear(dim * 3, dim), nn.SiLU())
        self.layers = nn.ModuleDict()
        self.global_mlp = _mlp(dim, dim * 4, dim)
        self.context_mlp = _mlp(dim * 2, dim * 4, dim)
        self.mlp = _mlp(dim, dim * 4, dim)
        self.gru = nn.GRU(dim, dim)
        for i, h in enumerate(num_heads):
            self.layers[f'time{i}'] = _TimeEmb(dim)
            self.layers[f'conv{i}'] = _Conv(dim, h)
            self.layers[f'norm{i}'] = nn.Identity()
            self.layers[f'act{i}'] = nn.SiLU()
        self.dummy_edge_feats = nn.Parameter(torch.randn(dim))
        self.node_out_mlp = nn.Sequential(nn.Linear(dim * 4, dim * 2), nn.SiLU(), nn.Linear(dim * 2, dim * 2),
                                          nn.SiLU(), nn.Linear(dim * 2, dim * 2))
        self.final_out = nn.Sequential(nn.Linear(dim * 2, dim * 2), nn.SiLU(), nn.Linear(dim * 2, dim), nn.SiLU(),
                                       nn.Linear(dim, self.num_classes))
        self._pos_emb = None
        self._folded = None
        self._cparams = None
        self._folded_key = None
        self._fold_epoch = 0

    # ---- C structs ----------------------------------------------------------------------
    def param_slots(self):
        """[(field of edgehip_gnn_params, layer index or None, parameter)]: which tensor each pointer of the
        C struct addresses (SURVEY App. C names)."""
        out = [('embedding_t_w', None, self.embedding_t.weight), ('embedding_t_b', None, self.embedding_t.bias),
               ('embedding_0_w', None, self.embedding_0.weight), ('embedding_0_b', None, self.embedding_0.bias),
               ('embedding_sel_w', None, self.embedding_sel.weight),
               ('node_in_w', None, self.node_in[0].weight), ('node_in_b', None, self.node_in[0].bias)]
        for nm, m in (('mlp', self.mlp), ('global', self.global_mlp), ('context', self.context_mlp)):
            out += [(f'{nm}_w0', None, m[0].weight), (f'{nm}_b0', None, m[0].bias),
                    (f'{nm}_w2', None, m[2].weight), (f'{nm}_b2', None, m[2].bias)]
        out += [('gru_w_ih', None, self.gru.weight_ih_l0), ('gru_w_hh', None, self.gru.weight_hh_l0),
                ('gru_b_ih', None, self.gru.bias_ih_l0), ('gru_b_hh', None, self.gru.bias_hh_l0)]
        for i in range(len(self.num_heads)):
            tm, cv = self.layers[f'time{i}'], self.layers[f'conv{i}']
            out += [('time_w', i, tm.linear.weight), ('time_b', i, tm.linear.bias),
                    ('query_w', i, cv.lin_query.weight), ('query_b', i, cv.lin_query.bias),
                    ('key_w', i, cv.lin_key.weight), ('key_b', i, cv.lin_key.bias),
                    ('value_w', i, cv.lin_value.weight), ('value_b', i, cv.lin_value.bias),
                    ('skip_w', i, cv.lin_skip.weight), ('skip_b', i, cv.lin_skip.bias)]
        for j, k in ((0, '0'), (2, '2'), (4, '4')):
            out += [(f'node_out_w{k}', None, self.node_out_mlp[j].weight), (f'node_out_b{k}', None, self.node_out_mlp[j].bias),
                    (f'final_w{k}', None, self.final_out[j].weight), (f'final_b{k}', None, self.final_out[j].bias)]
        return out

    def fill_struct(self, P, address_of):
        """Sets every parameter pointer of `P` to address_of(parameter) (the parameter itself, or its gradient)."""
        for field, layer, prm in self.param_slots():
            if layer is None:
                setattr(P, field, address_of(prm))
            else:
                getattr(P, field)[layer] = address_of(prm)

    def pos_emb(self, dev):
        """The sinusoid table on `dev`, built once ('cuda' and 'cuda:0' name the same device: compare resolved
        indices, or the table would be rebuilt per call and cached structs would keep the freed one's address)."""
        dev = torch.device(dev)
        if dev.type == 'cuda' and dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        if self._pos_emb is None or self._pos_emb.device != dev:
            self._pos_emb = tables.pos_emb_table(self.num_steps, self.dim).to(dev)
        return self._pos_emb

    def raw_params(self):
        """GnnParams over the raw parameters only (no folding): what the training calls read."""
        dev = self.embedding_t.weight.device
        if dev.type != 'cuda':
            raise capi.EdgeHipError('the model must live on a GPU (model.to("cuda")); there is no CPU path')
        ps = list(self.parameters())
        key = (dev, tuple(p.data_ptr() for p in ps), self.pos_emb(dev).data_ptr())
        if getattr(self, '_raw_key', None) == key:
            return self._raw
        for p in ps:
            assert p.dtype == torch.float32 and p.is_contiguous()
        P = capi.GnnParams()
        P.dim, P.num_layers, P.max_degree, P.num_steps = self.dim, len(self.num_heads), int(self.max_degree), self.num_steps
        for i, h in enumerate(self.num_heads):
            P.heads[i] = h
        self.fill_struct(P, lambda t: t.data_ptr())
        P.pos_emb = self._pos_emb.data_ptr()
        self._raw, self._raw_key = P, key
        return P

    def weights_changed(self):
        """Tell the network its parameters were written: the next inference call refolds them.  Tensor version
        counters catch copy_/load_state_dict/in-place ops, but torch's fused optimizers update the parameters WITHOUT
        bumping them, so every differentiable loss (a step may follow), `edgehip.train_step` and the optimizers of
        `edgehip.make_optimizer` (post-step hook) call this."""
        self._fold_epoch += 1

    def c_params(self):
        """(GnnParams, folded buffer); refolds when any parameter changed (version counters, weights_changed()), and
        on EVERY call once the model has been through a differentiable loss: an optimizer this package knows nothing
        about may have stepped since (one fold launch per sampling run / evaluation batch / op-by-op reverse step)."""
        dev = self.embedding_t.weight.device
        if dev.type != 'cuda':
            raise capi.EdgeHipError('the model must live on a GPU (model.to("cuda")); there is no CPU path')
        ps = list(self.parameters())
        key = (dev, tuple(p.data_ptr() for p in ps), tuple(p._version for p in ps), self.pos_emb(dev).data_ptr(),
               self._fold_epoch)
        if key == self._folded_key and self._fold_epoch == 0:
            return self._cparams, self._folded
        for p in ps:
            assert p.dtype == torch.float32 and p.is_contiguous()
        P = capi.GnnParams()
        P.dim, P.num_layers, P.max_degree, P.num_steps = self.dim, len(self.num_heads), int(self.max_degree), self.num_steps
        for i, h in enumerate(self.num_heads):
            P.heads[i] = h
        self.fill_struct(P, lambda t: t.data_ptr())
        P.pos_emb = self._pos_emb.data_ptr()
        nfl = capi.lib.edgehip_gnn_folded_floats(byref(P))
        if nfl == 0:
            raise capi.EdgeHipError('edgehip_gnn_folded_floats: unsupported configuration')
        if self._folded is None or self._folded.numel() != nfl or self._folded.device != dev:
            self._folded = torch.empty(nfl, dtype=torch.float32, device=dev)
        capi.check(capi.lib.edgehip_gnn_prepare(byref(P), self._folded.data_ptr(), capi.stream_ptr()), 'gnn_prepare')
        self._cparams, self._folded_key = P, key
        return P, self._folded

    @_on_batch_device
    def forward(self, pyg_data, t_node, t_edge=None):
        """(log_node_attr passthrough, edge_logits [A,2]) - layers/layers.py:315,383.

        `pyg_data` is a GraphBatch whose active set was computed by the diffusion object
        (`_p_sample_and_set_actives` / `_q_set_actives`)."""
        st = pyg_data._state
        P, folded = self.c_params()
        L = pyg_data.layout
        ws = _workspace(pyg_data, P)
        s = capi.stream_ptr()
        t_graph = t_node[pyg_data._t['node_ptr'][:-1].long()].to(torch.int32).contiguous() if pyg_data.num_nodes else \
            torch.zeros(0, dtype=torch.int32, device=pyg_data.device)
        node_repr = torch.empty((pyg_data.num_nodes + 32, 128), dtype=torch.float32, device=pyg_data.device)
        capi.check(capi.lib.edgehip_gnn_forward(byref(L), byref(P), folded.data_ptr(), st['adj_t'].data_ptr(),
                                                st['deg_t'].data_ptr(), pyg_data.degree_i32.data_ptr(),
                                                t_graph.data_ptr(), 0, st['flags'].data_ptr(),
                                                st['act_nodes'].data_ptr(), st['kptr'].data_ptr(),
                                                st['counts'].data_ptr(), node_repr.data_ptr(), None,
                                                ws.data_ptr(), ws.numel(), s), 'gnn_forward')
        st['node_repr'] = node_repr
        A = int(st['aptr'][-1].item())
        logits = torch.empty((A, 2), dtype=torch.float32, device=pyg_data.device)
        if A:
            # logits only (adj = NULL): the pair tail without a draw, the state is not touched
            capi.check(capi.lib.edgehip_edge_resample(byref(L), byref(P), folded.data_ptr(), node_repr.data_ptr(),
                                                      st['act_nodes'].data_ptr(), st['kptr'].data_ptr(),
                                                      st['aptr'].data_ptr(), st['tptr'].data_ptr(),
                                                      st['counts'].data_ptr(), None, 0, None, None, 0, None,
                                                      None, logits.data_ptr(), None, A, s),
                       'edge_resample(logits)')
        return getattr(pyg_data, 'log_node_attr', None), logits


def _workspace(batch, P):
    nbytes = capi.lib.edgehip_gnn_workspace_bytes(byref(batch.layout), byref(P))
    ws = getattr(batch, '_ws', None)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=batch.device)
        batch._ws = ws
    return ws


_LUT_DEV = {}
GUMBEL_MAGIC = 0x45484750
gumbel_patch_info = {}          # device -> dict(n, eps_fast): what the exhaustive comparison found (tests, bench)


def _gumbel_lut(device):
    """The `gumbel_patch` blob of include/edgehip.h for `device` (kept under the old name: every call site passes
    tabs['lut']).  Built once per process and device: the host's g over ALL 2^24 grid points (tables.gumbel_lut, the
    reference's own torch ops) is compared on the device with the device's `exact` transform; the blob lists the grid
    points where they differ with the host's values, and the largest |fast - host|.  64 MB cross the bus once, three
    64 MB temporaries live for the comparison; what stays resident is 16 bytes + 8 bytes per differing point."""
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    key = str(device)
    if key not in _LUT_DEV:
        with torch.cuda.device(device):
            host = tables.gumbel_lut().to(device)
            exact, fast = torch.empty_like(host), torch.empty_like(host)
            capi.check(capi.lib.edgehip_gumbel_grid(exact.data_ptr(), fast.data_ptr(), 0, host.numel(), 0,
                                                    capi.stream_ptr()), 'gumbel_grid')
            keys = (exact != host).nonzero().flatten()
            eps = float((fast - host).abs().max().item())
            if not (eps == eps and eps < 1e30):
                eps = 1e30                                  # a broken fast path only means every decision takes the exact one
            n = int(keys.numel())
            blob = torch.zeros(4 + 2 * n, dtype=torch.int32, device=device)
            blob[0], blob[1] = GUMBEL_MAGIC, n
            blob[2:3] = torch.tensor([eps], dtype=torch.float32, device=device).view(torch.int32)
            blob[4:4 + n] = keys.to(torch.int32)
            blob[4 + n:] = host[keys].view(torch.int32)
            gumbel_patch_info[key] = dict(n=n, eps_fast=eps, bytes=int(blob.numel()) * 4)
            _LUT_DEV[key] = blob.view(torch.float32)
    return _LUT_DEV[key]


# ------------------------------------------------------------------------------------------
class BinomialDiffusionActive(nn.Module):
    """Degree-guided binomial diffusion (diffusion/diffusion_binomial_active.py:15-264 on top of
    diffusion_binomial_vanilla.py:14-45 and diffusion_base.py:114-230) for
    parametrization 'xt_prescribed_st' / loss_type 'vb_ce_xt_prescribred_st'."""

    def __init__(self, num_node_classes, num_edge_classes, initial_graph_sampler, denoise_fn, timesteps=1000,
                 loss_type='vb_kl', parametrization='x0', final_prob_node=None, final_prob_edge=None,
                 sample_time_method='importance', noise_schedule=tables.cosine_beta_schedule, device='cuda',
                 exact_gumbel=True):
        super().__init__()
        if final_prob_node is None:
            final_prob_node = [1 - 1e-12, 1e-12]     # train.py:40-43 of the reference
        self.num_node_classes, self.num_edge_classes = num_node_classes, num_edge_classes
        self._denoise_fn = denoise_fn
        self.initial_graph_sampler = initial_graph_sampler
        self.num_timesteps = timesteps
        self.sample_time_method = sample_time_method
        self.device = device
        self.loss_type, self.parametrization = loss_type, parametrization
        self.exact_gumbel = exact_gumbel
        buf = tables.schedule_buffers(timesteps, noise_schedule, final_prob_node, final_prob_edge)
        for k, v in buf.items():
            self.register_buffer(k, v)
        self.register_buffer('Lt_history', torch.zeros(timesteps))
        self.register_buffer('Lt_count', torch.zeros(timesteps))
        self._host_buf = {k: v.clone() for k, v in buf.items()}     # NOT the registered tensors: a load copies into those in place
        self._q_cum_h, self._q_one_h = tables.q_tables(buf)       # also validates the sparse formulation
        self._tabs = {}
        self.rng_seed = None

    # ---- tables on the device -----------------------------------------------------------
    def _tables(self, dev, nmax):
        t = self._tabs.get('dev')
        if t is None or t['device'] != dev or t['nmax'] < nmax:
            t = dict(device=dev, nmax=nmax, post=tables.posterior_table(self._host_buf, nmax).to(dev),
                     q_cum=self._q_cum_h.to(dev), q_one=self._q_one_h.to(dev),
                     lut=_gumbel_lut(dev) if self.exact_gumbel else None)
            self._tabs['dev'] = t
        return t

    def _next_seed(self):
        if self.rng_seed is not None:
            s = self.rng_seed
            self.rng_seed = (s * 6364136223846793005 + 1442695040888963407) & ((1 << 64) - 1)
            return s
        return int(torch.randint(0, 2 ** 62, (1,)).item())

    # ---- a10 / a16 : reverse step ---------------------------------------------------------
    @_on_batch_device
    def _p_sample_and_set_actives(self, batch, t, u_node=None, rng=None):
        """diffusion_binomial_active.py:121-128. t: python int (shared) or int32 tensor [B]."""
        L, dev = batch.layout, batch.device
        nmax = int(batch.degree_i32.max().item()) if batch.num_nodes else 0
        tabs = self._tables(dev, nmax)
        st = _state(batch)
        s = capi.stream_ptr()
        t_graph = torch.full((batch.num_graphs,), int(t), dtype=torch.int32, device=dev) if not torch.is_tensor(t) \
            else t.to(torch.int32).contiguous()
        st['t_graph'] = t_graph
        st['adj_t'] = batch.adj
        capi.check(capi.lib.edgehip_degree_from_adj(byref(L), batch.adj.data_ptr(), st['deg_t'].data_ptr(), s), 'degree')
        capi.check(capi.lib.edgehip_active_select_sample(
            byref(L), batch.degree_i32.data_ptr(), st['deg_t'].data_ptr(), t_graph.data_ptr(), tabs['post'].data_ptr(),
            tabs['nmax'], capi.ptr(u_node), byref(rng) if rng is not None else None, capi.ptr(tabs['lut']),
            st['flags'].data_ptr(), st['act_nodes'].data_ptr(), st['act_count'].data_ptr(), s), 'active_select_sample')
        self._set_offsets(batch)

    def _set_offsets(self, batch):
        st, L = batch._state, batch.layout
        capi.check(capi.lib.edgehip_active_offsets(byref(L), st['act_count'].data_ptr(), st['kptr'].data_ptr(),
                                                   st['aptr'].data_ptr(), st['tptr'].data_ptr(),
                                                   st['counts'].data_ptr(), capi.stream_ptr()), 'active_offsets')

    def _read_counts(self, batch, after=None):
        """(K, A) of the step from the device-side edgehip_counts.  With `after` (an event recorded right behind the
        offsets kernel) the copy runs on a side stream behind that event into pinned memory, so the host does not
        wait for work queued on the current stream AFTER the event (the node part of the training forward)."""
        st = batch._state
        if after is None:
            c = st['counts'].cpu().numpy().view(np.int32)   # edgehip_counts: k_total, pair_tiles, a_total (int64), ..: one sync
        else:
            dev = batch.device
            rb = self.__dict__.setdefault('_rb', {}).get(str(dev))
            if rb is None:
                rb = self._rb[str(dev)] = (capi.side_streams(dev, 2)[1], torch.empty(8, dtype=torch.int32).pin_memory(),
                                           torch.cuda.Event())
            stream, pinned, done = rb
            stream.wait_event(after)
            with torch.cuda.stream(stream):
                pinned.copy_(st['counts'], non_blocking=True)
                done.record(stream)
            done.synchronize()
            c = pinned.numpy().copy().view(np.int32)
        return int(c[0]), int(c[2:4].view(np.int64)[0])

    @_on_batch_device
    def materialize_actives(self, batch, counts=None):
        """Sets batch.active_node_indices / active_edge_indices (int64, ascending) like the reference."""
        st, L = batch._state, batch.layout
        K, A = self._read_counts(batch) if counts is None else counts
        ani = torch.empty(K, dtype=torch.int64, device=batch.device)
        aei = torch.empty(A, dtype=torch.int64, device=batch.device)
        row = torch.empty(A, dtype=torch.int64, device=batch.device)
        col = torch.empty(A, dtype=torch.int64, device=batch.device)
        capi.check(capi.lib.edgehip_active_pairs_enumerate(
            byref(L), st['act_nodes'].data_ptr(), st['kptr'].data_ptr(), st['aptr'].data_ptr(), st['tptr'].data_ptr(),
            st['counts'].data_ptr(), capi.ptr(ani) if K else None, K, capi.ptr(aei) if A else None,
            capi.ptr(row) if A else None, capi.ptr(col) if A else None, A, capi.stream_ptr()), 'enumerate')
        batch.active_node_indices, batch.active_edge_indices = ani, aei
        batch.active_row, batch.active_col = row, col
        return ani, aei

    @_on_batch_device
    @torch.no_grad()
    def p_sample(self, batch, t, u_node=None, u_pair=None, rng=None, logits_in=None, want=None):
        """One reverse step, op by op (diffusion_binomial_active.py:72-90).  With supplied uniforms
        `u_pair` may be a callable A -> tensor[A,2] (A is only known after the selection).
        `want`: optional dict that receives logits / bits."""
        self._p_sample_and_set_actives(batch, t, u_node, rng)
        st, L = batch._state, batch.layout
        A = int(st['aptr'][-1].item())
        if A == 0:
            return batch
        net = self._denoise_fn
        P, folded = net.c_params()
        ws = _workspace(batch, P)
        s = capi.stream_ptr()
        node_repr = torch.empty((batch.num_nodes + 32, 128), dtype=torch.float32, device=batch.device)
        if logits_in is None:
            capi.check(capi.lib.edgehip_gnn_forward(
                byref(L), byref(P), folded.data_ptr(), batch.adj.data_ptr(), st['deg_t'].data_ptr(),
                batch.degree_i32.data_ptr(), st['t_graph'].data_ptr(), 0, st['flags'].data_ptr(),
                st['act_nodes'].data_ptr(), st['kptr'].data_ptr(), st['counts'].data_ptr(), node_repr.data_ptr(), None,
                ws.data_ptr(), ws.numel(), s), 'gnn_forward')
        if callable(u_pair):
            u_pair = u_pair(A)
        logits_out = torch.empty((A, 2), dtype=torch.float32, device=batch.device) if want is not None else None
        bits_out = torch.empty(A, dtype=torch.uint8, device=batch.device) if want is not None else None
        lut = self._tabs['dev']['lut']
        step = int(t) if not torch.is_tensor(t) else 0
        capi.check(capi.lib.edgehip_edge_resample(
            byref(L), byref(P), folded.data_ptr(), node_repr.data_ptr(), st['act_nodes'].data_ptr(),
            st['kptr'].data_ptr(), st['aptr'].data_ptr(), st['tptr'].data_ptr(), st['counts'].data_ptr(),
            capi.ptr(logits_in), 0, capi.ptr(u_pair), byref(rng) if rng is not None else None, step, capi.ptr(lut),
            batch.adj.data_ptr(), capi.ptr(logits_out), capi.ptr(bits_out), A, s), 'edge_resample')
        if want is not None:
            want['logits'], want['bits'] = logits_out, bits_out
        return batch

    @torch.no_grad()
    def sample(self, num_samples, seed=None):
        """diffusion_base.py:190-230: T fused reverse steps (no host sync), then the edge-list epilogue."""
        batch = self.initial_graph_sampler.sample(num_samples)
        batch = batch.to(self.device)
        return self.sample_batch(batch, seed)

    @staticmethod
    def default_parts(num_graphs):
        """Sub-batches (= HIP streams, the caller's included) the sampler cuts a batch into: two from 16 graphs on (the
        side stream comes from the package's one pool, capi.side_streams).  polblogs on one MI355X, one box, graphs/s
        with 1 / 2 / 3 / 4 streams: 64 graphs 135.4 / 143.2 / 141.0 / 139.0, 32 graphs - / 106.6 / 104.8 / 107.1,
        16 graphs 65.8 / 70.2: while one block sits in its latency-bound launches the other's fused layer kernel
        fills the chip; a third and fourth block add launches without more overlap."""
        return 2 if num_graphs >= 16 else 1

    @_on_batch_device
    @torch.no_grad()
    def sample_batch(self, batch, seed=None, parts=None, step_stats=None):
        """T fused reverse steps of a prepared batch (no host sync), then the edge-list epilogue.
        parts > 1: contiguous blocks of graphs run on separate HIP streams, their steps issued round-robin.
        Graphs do not interact and the uniforms are keyed by the global graph id, so the result is the same
        bit for bit; but while one block sits in its latency-bound launches (per-graph update, prepare, tail)
        another block's fused layer kernel fills the chip."""
        L, dev = batch.layout, batch.device
        net = self._denoise_fn
        P, folded = net.c_params()
        nmax = int(batch.degree_i32.max().item()) if batch.num_nodes else 0
        tabs = self._tables(dev, nmax)
        rng = capi.Rng(self._next_seed() if seed is None else seed)
        parts = self.default_parts(batch.num_graphs) if parts is None else max(1, min(int(parts), max(batch.num_graphs, 1)))
        if step_stats is not None:
            # measurement: int64 [T, 8] device tensor, row t <- edgehip_prof_step_stats after step t (no host sync)
            assert step_stats.dtype == torch.int64 and tuple(step_stats.shape) == (self.num_timesteps, 8) and \
                step_stats.is_contiguous() and step_stats.device == dev
            parts = 1
        step = capi.lib.edgehip_sample_step
        post, lut, tn = tabs['post'].data_ptr(), capi.ptr(tabs['lut']), tabs['nmax']
        fptr, rref, pref = folded.data_ptr(), byref(rng), byref(P)
        if parts == 1:
            ws = _workspace(batch, P)
            batch.adj.zero_()               # x_T: the prior [1-1e-12, 1e-12] always yields the empty graph
            work = [(byref(L), batch.adj.data_ptr(), batch.degree_i32.data_ptr(), _state(batch)['counts'].data_ptr(),
                     ws.data_ptr(), ws.numel(), capi.stream_ptr())]
            streams = []
        else:
            subs = batch.split(parts)
            cur = torch.cuda.current_stream(dev)
            # the first block stays on the caller's stream: a process has 4 hardware queues (GPU_MAX_HW_QUEUES), and a
            # stream beyond them shares a queue with another one (measured: 82.6 graphs/s with 4 side streams)
            pool = [cur] + self._streams(dev, len(subs) - 1)      # at most 1 + capi.MAX_SIDE_STREAMS different streams
            streams = [pool[i % len(pool)] for i in range(len(subs))]
            work = []
            # every buffer of the blocks is allocated (and zero-filled) on the current stream BEFORE the side streams
            # are ordered behind it: a fill queued later on `cur` could land on top of a block's first step
            prepared = [(_state(b)['counts'], _workspace(b, P)) for b in subs]
            for st in pool[1:]:
                st.wait_stream(cur)          # also: tables / folded weights were produced on the current stream
            for b, st, (counts, ws) in zip(subs, streams, prepared):
                with torch.cuda.stream(st):
                    b.adj.zero_()
                work.append((byref(b.layout), b.adj.data_ptr(), b.degree_i32.data_ptr(), counts.data_ptr(),
                             ws.data_ptr(), ws.numel(), st.cuda_stream))
        for t in reversed(range(self.num_timesteps)):
            for (Lr, adj, d0, counts, ws, wsn, s) in work:
                rc = step(Lr, pref, fptr, adj, d0, t, post, tn, None, None, rref, lut, counts, ws, wsn, s)
                if rc:
                    capi.check(rc, f'sample_step t={t}')
                if step_stats is not None:
                    capi.check(capi.lib.edgehip_prof_step_stats(Lr, pref, ws, wsn, counts, step_stats.data_ptr() + 64 * t, s),
                               'prof_step_stats')
        for st in set(streams[1:]) - {streams[0]} if streams else ():
            streams[0].wait_stream(st)
        return self._finish_sample(batch)

    def _streams(self, dev, n):
        return capi.side_streams(dev, n)

    def _finish_sample(self, batch):
        edge_index, edge_slice = batch.extract_edges()
        batch.edge_index = edge_index
        batch.edge_attr = torch.ones(edge_index.shape[1], dtype=torch.long, device=batch.device)
        batch.node_attr = torch.zeros(batch.num_nodes, dtype=torch.long, device=batch.device)
        batch._slice_dict['edge_index'] = edge_slice
        batch._inc_dict['edge_index'] = batch._inc_dict.get('full_edge_index')
        return batch

    # ---- a20 : time sampling ---------------------------------------------------------------
    def _importance_ready(self):
        """`(Lt_count > 10).all()` of diffusion_binomial_vanilla.py:250 without a device sync on every step:
        the counts only grow, one per sampled t, so the test cannot pass before 11*T samples were added (a host
        counter, seeded from the buffer once so that restored checkpoints count) and never fails again once it
        has passed."""
        if getattr(self, '_lt_ready', False):
            return True
        if getattr(self, '_lt_samples', None) is None:
            self._lt_samples = int(self.Lt_count.sum().item())
        if self._lt_samples < 11 * self.num_timesteps:
            return False
        ready = bool((self.Lt_count > 10).all())
        self._lt_ready = ready
        return ready

    _SCHEDULE_KEYS = ('log_alpha', 'log_1_min_alpha', 'log_cumprod_alpha', 'log_1_min_cumprod_alpha',
                      'log_final_prob_node', 'log_final_prob_edge')

    def load_state_dict(self, *args, **kwargs):
        self._lt_ready, self._lt_samples = False, None      # Lt_count may change under us
        out = super().load_state_dict(*args, **kwargs)
        self._refresh_schedule()
        return out

    def _refresh_schedule(self):
        """The kernels read host-built tables (posterior, q tables, prior-KL constants), not the registered buffers.
        The reference always computes from the buffers a checkpoint restored (diffusion/base.py:171), so after a load
        the tables are rebuilt from them - whatever --noise_schedule the constructor was given, and whether the
        module was moved to the GPU before or after the load."""
        buf = {k: getattr(self, k).detach().cpu().clone() for k in self._SCHEDULE_KEYS}
        if all(torch.equal(buf[k], self._host_buf[k]) for k in self._SCHEDULE_KEYS):
            return
        self._host_buf = buf
        self._q_cum_h, self._q_one_h = tables.q_tables(buf)
        self._tabs = {}
        self._kl_consts = None

    def _sample_time(self, b, device, method='uniform'):
        """diffusion_binomial_vanilla.py:248-269."""
        if method == 'importance':
            if not self._importance_ready():
                return self._sample_time(b, device, method='uniform')
            Lt_sqrt = torch.sqrt(self.Lt_history + 1e-10) + 0.0001
            Lt_sqrt[0] = Lt_sqrt[1]
            pt_all = Lt_sqrt / Lt_sqrt.sum()
            t = torch.multinomial(pt_all, num_samples=b, replacement=True)
            return t, pt_all.gather(dim=0, index=t)
        elif method == 'uniform':
            length = self.Lt_count.shape[0]
            t = torch.randint(0, length, (b,), device=device).long()
            return t, torch.ones_like(t).float() / length
        raise ValueError

    # ---- a19 : prior KL, closed form per (#nodes, #pairs, #edges) ----------------------------
    def _kl_prior(self, batch):
        """diffusion_binomial_vanilla.py:116-136.  Every element of a class contributes the same
        value, so the per-graph sum is count x value (three values, computed with the reference's ops)."""
        cached = getattr(batch, '_kl_prior_cache', None)
        if cached is not None and cached[0] is self._host_buf:
            return cached[1]
        consts = getattr(self, '_kl_consts', None)
        if consts is None:
            T = self.num_timesteps
            b = self._host_buf
            la, l1a = b['log_cumprod_alpha'][T - 1], b['log_1_min_cumprod_alpha'][T - 1]
            zero = torch.log(torch.tensor([[1.0, 1e-30]]))
            one = torch.log(torch.tensor([[1e-30, 1.0]]))

            def kl(x, lf):
                q = tables._log_add_exp(x + la, l1a + lf)
                return float((q.exp() * (q - lf)).sum(dim=1).double())
            consts = self._kl_consts = (kl(zero, b['log_final_prob_node']), kl(zero, b['log_final_prob_edge']),
                                        kl(one, b['log_final_prob_edge']))
        kn, k0, k1 = consts
        n = batch.nodes_per_graph.double()
        P = batch.edges_per_graph.double()
        E = _x0_edge_counts(batch).double()
        out = (n * kn + (P - E) * k0 + E * k1).float()
        batch._kl_prior_cache = (self._host_buf, out)        # x0, the sizes and the schedule are fixed for a batch object
        return out

    # ---- a18, a21 : loss -------------------------------------------------------------------------
    def log_prob(self, batch, t=None, pt=None, u_tm1=None, u_t=None, want=None, dropout_masks=None):
        """dropout_masks (parity runs): uint8 [2*layers, Nn, 64] keep-masks used instead of generated ones when the
        network is in training mode with --dp_rate > 0."""
        self._dropout_masks = dropout_masks
        if self.loss_type != 'vb_ce_xt_prescribred_st':
            if self.loss_type in ('vb_kl', 'vb_ce_xt', 'vb_ce_xt_kl_st', 'vb_ce_xt_ce_st'):
                raise NotImplementedError(f"loss_type '{self.loss_type}' is outside the accelerated hot path")
            raise ValueError()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._denoise_fn.parameters()):
            return self._loss(batch, self.training, t, pt, u_tm1, u_t, want, differentiable=True)
        with torch.no_grad():
            return self._loss(batch, self.training, t, pt, u_tm1, u_t, want)

    @_on_batch_device
    def _loss(self, batch, training, t, pt, u_tm1, u_t, want, differentiable=False):
        """_train_loss / _eval_loss, diffusion_binomial_active.py:228-264 / 195-226. Returns -vb [B].
        differentiable=False: the fused inference kernels (no grad).  True: the op-by-op training form
        (edgehip/train.py), whose result back-propagates to the network's parameters."""
        L, dev, B = batch.layout, batch.device, batch.num_graphs
        if t is None:
            t, pt = self._sample_time(B, dev, self.sample_time_method)
        t, pt = t.to(dev), pt.to(dev)
        nmax = 0
        tabs = self._tables(dev, nmax)
        st = _state(batch)
        if differentiable:
            # the autograd node keeps deg_t / flags for its backward: a second differentiable loss on the same batch
            # before the first backward must not overwrite them (adj_t, adj_tm1, deg_tm1 are fresh per call already)
            st['deg_t'], st['flags'] = torch.zeros_like(st['deg_t']), torch.zeros_like(st['flags'])
        s = capi.stream_ptr()
        t_graph = t.to(torch.int32).contiguous()
        adj_tm1, adj_t = batch.new_adj(), batch.new_adj()
        rng = None if u_tm1 is not None else capi.Rng(self._next_seed())
        capi.check(capi.lib.edgehip_q_sample_edges(
            byref(L), batch.adj.data_ptr(), t_graph.data_ptr(), tabs['q_cum'].data_ptr(), tabs['q_one'].data_ptr(),
            capi.ptr(u_tm1), capi.ptr(u_t), byref(rng) if rng is not None else None, capi.ptr(tabs['lut']),
            adj_tm1.data_ptr(), adj_t.data_ptr(), s), 'q_sample_edges')
        deg_tm1 = torch.empty_like(st['deg_t'])
        capi.check(capi.lib.edgehip_degree_from_adj(byref(L), adj_tm1.data_ptr(), deg_tm1.data_ptr(), s), 'degree')
        capi.check(capi.lib.edgehip_degree_from_adj(byref(L), adj_t.data_ptr(), st['deg_t'].data_ptr(), s), 'degree')
        capi.check(capi.lib.edgehip_active_select_train(byref(L), deg_tm1.data_ptr(), st['deg_t'].data_ptr(),
                                                        st['flags'].data_ptr(), st['act_nodes'].data_ptr(),
                                                        st['act_count'].data_ptr(), s), 'active_select_train')
        self._set_offsets(batch)
        st['adj_t'], st['adj_tm1'], st['deg_tm1'], st['t_graph'] = adj_t, adj_tm1, deg_tm1, t_graph
        net = self._denoise_fn
        # torch.nn.Dropout follows the MODULE's mode (layers.py:286,357,361): active in train(), also under no_grad
        dp = None
        if net.training and net.dropout_p > 0:
            dp = capi.Dropout()
            dp.p, dp.seed = float(net.dropout_p), self._next_seed()
            masks = getattr(self, '_dropout_masks', None)
            if masks is not None:
                assert masks.dtype == torch.uint8 and masks.is_contiguous() and masks.device == dev and \
                    tuple(masks.shape) == (2 * len(net.num_heads), batch.num_nodes, 64)
                dp.masks = masks.data_ptr()
                dp._keep = masks
        if differentiable or dp is not None:      # the op-by-op / fused TRAINING form (the inference kernels have no dropout)
            from . import train
            net.pos_emb(dev)
            if differentiable:
                net.weights_changed()       # an optimizer step may follow this loss (see weights_changed)
            counts = None
            if getattr(self, 'train_form', 'fused') != 'ops':
                # the node part of the network needs no active pairs: it is queued before the step's only host sync
                # (the pair counts), read back on a side stream behind the offsets kernel
                after = torch.cuda.Event()
                after.record(torch.cuda.current_stream(dev))
                train.prelaunch_nodes(net, batch, adj_t, st['deg_t'], st['flags'], t_graph, dp)
                counts = self._read_counts(batch, after)
            self.materialize_actives(batch, counts)
            row, col = batch.active_row, batch.active_col
            tgt, pair_graph = train.pair_targets(batch, adj_tm1, row, col)
            fwd = train.tgnn_ce_forward if getattr(self, 'train_form', 'fused') == 'ops' else train.tgnn_ce_fused
            ce, logits = fwd(net, batch, adj_t, st['deg_t'], st['flags'], t_graph, row, col, tgt, pair_graph, dp)
            if not ce.requires_grad:
                batch._train_ws_busy = False      # no backward will come to release the workspace
            npairs = batch.edges_per_graph.to(dev).float()
            n_act = (st['aptr'][1:] - st['aptr'][:-1]).float()
            kl = ce + 6.9078e-29 * (npairs - n_act) + 6.9078e-29 * npairs          # active.py:176-181
        else:
            P, folded = net.c_params()
            ws = _workspace(batch, P)
            node_repr = torch.empty((batch.num_nodes + 32, 128), dtype=torch.float32, device=dev)
            capi.check(capi.lib.edgehip_gnn_forward(
                byref(L), byref(P), folded.data_ptr(), adj_t.data_ptr(), st['deg_t'].data_ptr(),
                batch.degree_i32.data_ptr(), t_graph.data_ptr(), 0, st['flags'].data_ptr(), st['act_nodes'].data_ptr(),
                st['kptr'].data_ptr(), st['counts'].data_ptr(), node_repr.data_ptr(), None, ws.data_ptr(), ws.numel(), s),
                'gnn_forward')
            A = int(st['aptr'][-1].item())
            ntile = int(st['tptr'][-1].item())
            partial = torch.zeros(max(ntile, 1), dtype=torch.float32, device=dev)
            kl = torch.empty(B, dtype=torch.float32, device=dev)
            logits = torch.empty((A, 2), dtype=torch.float32, device=dev) if want is not None else None
            capi.check(capi.lib.edgehip_edge_ce_loss(
                byref(L), byref(P), folded.data_ptr(), node_repr.data_ptr(), st['act_nodes'].data_ptr(),
                st['kptr'].data_ptr(), st['aptr'].data_ptr(), st['tptr'].data_ptr(), st['counts'].data_ptr(),
                adj_tm1.data_ptr(), partial.data_ptr(), ntile, kl.data_ptr(),
                capi.ptr(logits) if A and logits is not None else None, A, s), 'edge_ce_loss')
        if training:     # active.py:250-254; under data parallelism every replica applies every rank's pairs
            from . import shard
            # one all-gather carries the local pair count and every rank's (t, kl)
            batch._global_entries, t_all, kl_all = shard.gather_step_scalars(batch.num_pairs, t, kl.detach())
            Lt2 = kl_all.pow(2)
            prev = self.Lt_history.gather(dim=0, index=t_all)
            self.Lt_history.scatter_(dim=0, index=t_all, src=(0.1 * Lt2 + 0.9 * prev))
            self.Lt_count.scatter_add_(dim=0, index=t_all, src=torch.ones_like(Lt2))
            if getattr(self, '_lt_samples', None) is not None:
                self._lt_samples += int(t_all.numel())
        prior = self._kl_prior(batch).to(dev)
        vb = kl / pt + prior
        batch.num_entries = batch.num_pairs            # active.py:192-193
        if want is not None:
            want.update(kl=kl.detach(), kl_prior=prior, logits=logits, t=t, pt=pt)
        return -vb


def _state(batch):
    st = getattr(batch, '_state', None)
    if st is None:
        dev, Nn, B = batch.device, batch.num_nodes, batch.num_graphs
        st = dict(deg_t=torch.zeros(Nn + 32, dtype=torch.int32, device=dev),
                  flags=torch.zeros(Nn + 32, dtype=torch.uint8, device=dev),
                  act_nodes=torch.zeros(Nn + 32, dtype=torch.int32, device=dev),
                  act_count=torch.zeros(B + 1, dtype=torch.int32, device=dev),
                  kptr=torch.zeros(B + 1, dtype=torch.int32, device=dev),
                  aptr=torch.zeros(B + 1, dtype=torch.int64, device=dev),
                  tptr=torch.zeros(B + 1, dtype=torch.int32, device=dev),
                  counts=torch.zeros(8, dtype=torch.int32, device=dev))
        batch._state = st
    return st


def _x0_edge_counts(batch):
    c = getattr(batch, '_x0_counts', None)
    if c is None:
        _, sl = batch.extract_edges(batch.adj)
        c = (sl[1:] - sl[:-1])
        batch._x0_counts = c
    return c


def elbo_bpd(model, x):
    """diffusion/loss.py:10-12,24-29."""
    import math
    return -model.log_prob(x).sum() / (math.log(2) * x.num_entries)
