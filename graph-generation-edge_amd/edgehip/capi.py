"""ctypes binding of libedgehip.so (include/edgehip.h).

The library is the product: if it is missing this module raises at import - there is no
CPU or eager fallback.  Tensors are passed as raw device pointers (`tensor.data_ptr()`), the
stream as `torch.cuda.current_stream().cuda_stream`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('EDGEHIP_LIB') or os.path.join(os.path.dirname(_HERE), 'lib', 'libedgehip.so')

MAX_LAYERS = 8
NSTAT = 10
TILE_ROWS = 32
PAIR_TILE = 32

OK, ERR_ARG, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_LAUNCH, ERR_TOO_LARGE = 0, -1, -2, -3, -4, -5
_ERR = {ERR_ARG: 'bad argument', ERR_UNSUPPORTED: 'unsupported configuration', ERR_WORKSPACE: 'workspace too small',
        ERR_LAUNCH: 'kernel launch failed', ERR_TOO_LARGE: 'input exceeds a fixed capacity (static scratch of a no-allocation entry point, or a layout whose '
                       '32-bit pair-tile counters could wrap: edgehip_layout_pair_tile_bound > EDGEHIP_MAX_PAIR_TILES)'}
STATUS_CAPACITY = 1

vp = C.c_void_p


class Layout(C.Structure):
    _fields_ = [('num_graphs', C.c_int32), ('num_nodes', C.c_int32), ('num_tiles', C.c_int32),
                ('max_nodes', C.c_int32), ('gid_base', C.c_int32), ('reserved', C.c_int32),
                ('node_ptr', vp), ('node_graph', vp), ('adj_ptr', vp), ('adj_stride', vp), ('pair_ptr', vp),
                ('tile_ptr', vp), ('tile_graph', vp), ('tile_row0', vp), ('tile_rows', vp)]


class Counts(C.Structure):
    _fields_ = [('k_total', C.c_int32), ('pair_tiles', C.c_int32), ('a_total', C.c_int64),
                ('status', C.c_int32), ('reserved', C.c_int32)]


class Rng(C.Structure):
    _fields_ = [('seed', C.c_uint64)]


class Dropout(C.Structure):
    _fields_ = [('p', C.c_float), ('reserved', C.c_int32), ('seed', C.c_uint64), ('masks', vp)]


_PL = vp * MAX_LAYERS


class GnnParams(C.Structure):
    _fields_ = [('dim', C.c_int32), ('num_layers', C.c_int32), ('heads', C.c_int32 * MAX_LAYERS),
                ('max_degree', C.c_int32), ('num_steps', C.c_int32)] + \
        [(n, vp) for n in ('embedding_t_w', 'embedding_t_b', 'embedding_0_w', 'embedding_0_b', 'embedding_sel_w',
                           'node_in_w', 'node_in_b', 'mlp_w0', 'mlp_b0', 'mlp_w2', 'mlp_b2',
                           'global_w0', 'global_b0', 'global_w2', 'global_b2',
                           'context_w0', 'context_b0', 'context_w2', 'context_b2',
                           'gru_w_ih', 'gru_w_hh', 'gru_b_ih', 'gru_b_hh')] + \
        [(n, _PL) for n in ('time_w', 'time_b', 'query_w', 'query_b', 'key_w', 'key_b', 'value_w', 'value_b',
                            'skip_w', 'skip_b')] + \
        [(n, vp) for n in ('node_out_w0', 'node_out_b0', 'node_out_w2', 'node_out_b2', 'node_out_w4', 'node_out_b4',
                           'final_w0', 'final_b0', 'final_w2', 'final_b2', 'final_w4', 'final_b4', 'pos_emb')]


EXPORTS = {
    'edgehip_degree_from_adj': [C.POINTER(Layout), vp, vp, vp],
    'edgehip_active_select_sample': [C.POINTER(Layout), vp, vp, vp, vp, C.c_int32, vp, C.POINTER(Rng), vp, vp, vp, vp, vp],
    'edgehip_active_select_train': [C.POINTER(Layout), vp, vp, vp, vp, vp, vp],
    'edgehip_active_offsets': [C.POINTER(Layout), vp, vp, vp, vp, vp, vp],
    'edgehip_active_pairs_enumerate': [C.POINTER(Layout), vp, vp, vp, vp, vp, vp, C.c_int64, vp, vp, vp, C.c_int64, vp],
    'edgehip_q_sample_edges': [C.POINTER(Layout), vp, vp, vp, vp, vp, vp, C.POINTER(Rng), vp, vp, vp, vp],
    'edgehip_gumbel_grid': [vp, vp, C.c_int64, C.c_int64, C.c_int32, vp],
    'edgehip_count_edges': [C.POINTER(Layout), vp, vp, vp, vp],
    'edgehip_extract_edges': [C.POINTER(Layout), vp, vp, vp, vp, C.c_int64, vp],
    'edgehip_adj_from_edges': [C.POINTER(Layout), vp, C.c_int64, vp, vp],
    'edgehip_gnn_prepare': [C.POINTER(GnnParams), vp, vp],
    'edgehip_gnn_forward': [C.POINTER(Layout), C.POINTER(GnnParams), vp, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp,
                            vp, vp, vp, C.c_size_t, vp],
    'edgehip_edge_resample': [C.POINTER(Layout), C.POINTER(GnnParams), vp, vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp,
                              C.POINTER(Rng), C.c_int32, vp, vp, vp, vp, C.c_int64, vp],
    'edgehip_edge_ce_loss': [C.POINTER(Layout), C.POINTER(GnnParams), vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int64,
                             vp, vp, C.c_int64, vp],
    'edgehip_edge_ce_loss_static': [C.POINTER(Layout), C.POINTER(GnnParams), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                    C.c_int64, vp],
    'edgehip_prof_step_stats': [C.POINTER(Layout), C.POINTER(GnnParams), vp, C.c_size_t, vp, vp, vp],
    'edgehip_sample_step': [C.POINTER(Layout), C.POINTER(GnnParams), vp, vp, vp, C.c_int32, vp, C.c_int32, vp, vp,
                            C.POINTER(Rng), vp, vp, vp, C.c_size_t, vp],
    'edgehip_linear_fwd': [vp, C.c_int64, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp, vp],
    'edgehip_graph_stats': [C.POINTER(Layout), vp, vp, vp, vp],
    'edgehip_graph_stats_ext': [C.POINTER(Layout), vp, vp, vp, vp, vp],
    'edgehip_edge_overlap': [C.POINTER(Layout), vp, vp, vp, vp],
    # training ops (csrc/train.hip)
    'edgehip_train_embed_fwd': [C.POINTER(Layout), vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp],
    'edgehip_train_embed_bwd': [C.POINTER(Layout), vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp],
    'edgehip_linear_bwd_input': [vp, C.c_int64, C.c_int32, vp, C.c_int32, vp, vp, vp],
    'edgehip_linear_bwd_weight': [vp, vp, C.c_int64, C.c_int32, C.c_int32, vp, vp, vp],
    'edgehip_act_fwd': [C.c_int32, vp, C.c_int64, vp, vp],
    'edgehip_act_bwd': [C.c_int32, vp, vp, C.c_int64, vp, vp],
    'edgehip_assemble_fwd': [C.POINTER(Layout), vp, vp, vp, vp, C.c_int32, vp, vp],
    'edgehip_assemble_bwd': [C.POINTER(Layout), vp, C.c_int32, vp, vp, vp, vp, vp],
    'edgehip_segment_mean_fwd': [C.POINTER(Layout), vp, vp, vp, vp],
    'edgehip_segment_mean_bwd': [C.POINTER(Layout), vp, vp, vp],
    'edgehip_attn_fwd': [C.POINTER(Layout), vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp],
    'edgehip_attn_bwd': [C.POINTER(Layout), vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    'edgehip_gru_gate_fwd': [vp, vp, vp, C.c_int64, vp, vp],
    'edgehip_gru_gate_bwd': [vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp],
    'edgehip_pair_gather_add_fwd': [vp, C.c_int32, vp, vp, C.c_int64, vp, vp],
    'edgehip_pair_gather_add_bwd': [vp, C.c_int32, vp, vp, C.c_int64, vp, vp],
    'edgehip_pair_targets': [C.POINTER(Layout), vp, vp, vp, C.c_int64, vp, vp, vp],
    'edgehip_edge_ce_fwd': [vp, vp, vp, vp, vp, C.c_int64, vp, vp, vp],
    'edgehip_edge_ce_bwd': [vp, vp, vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp],
    # the whole differentiable network in two calls
    'edgehip_train_forward': [C.POINTER(Layout), C.POINTER(GnnParams), vp, vp, vp, vp, vp, vp, vp, C.c_int64, vp, vp, vp,
                              vp, vp, C.c_size_t, vp],
    'edgehip_train_backward': [C.POINTER(Layout), C.POINTER(GnnParams), C.POINTER(GnnParams), vp, vp, vp, vp, vp, vp,
                               C.c_int64, vp, vp, vp, vp, C.c_size_t, vp],
    'edgehip_train_forward_nodes': [C.POINTER(Layout), C.POINTER(GnnParams), vp, vp, vp, vp, vp, C.c_int64, vp, C.c_size_t, vp],
    'edgehip_train_forward_pairs': [C.POINTER(Layout), C.POINTER(GnnParams), vp, vp, C.c_int64, C.c_int64, vp, vp, vp, vp, vp,
                                    C.c_size_t, vp],
    'edgehip_dropout_fwd': [C.POINTER(Layout), vp, C.POINTER(Dropout), C.c_int32, vp, vp],
    'edgehip_train_forward_nodes_dp': [C.POINTER(Layout), C.POINTER(GnnParams), vp, vp, vp, vp, vp, C.c_int64,
                                       C.POINTER(Dropout), vp, C.c_size_t, vp],
    'edgehip_train_backward_dp': [C.POINTER(Layout), C.POINTER(GnnParams), C.POINTER(GnnParams), vp, vp, vp, vp, vp, vp,
                                  C.c_int64, C.c_int64, vp, vp, vp, C.POINTER(Dropout), vp, C.c_size_t, vp],
    'edgehip_train_backward_cap': [C.POINTER(Layout), C.POINTER(GnnParams), C.POINTER(GnnParams), vp, vp, vp, vp, vp, vp,
                                   C.c_int64, C.c_int64, vp, vp, vp, vp, C.c_size_t, vp],
}
SIZE_EXPORTS = {
    'edgehip_gnn_folded_floats': [C.POINTER(GnnParams)],
    'edgehip_gnn_workspace_bytes': [C.POINTER(Layout), C.POINTER(GnnParams)],
    'edgehip_train_workspace_bytes': [C.POINTER(Layout), C.POINTER(GnnParams), C.c_int64],
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f'{LIB_PATH} not found: build it with `python __graft_entry__.py` '
                          '(make -C graph-generation-edge_amd/csrc); there is no fallback path')
    lib = C.CDLL(LIB_PATH)
    for name, args in EXPORTS.items():
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = C.c_int
    for name, args in SIZE_EXPORTS.items():
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = C.c_size_t
    lib.edgehip_version.restype = C.c_char_p
    lib.edgehip_edge_ce_loss_static_max_tiles.restype = C.c_int64
    lib.edgehip_layout_pair_tile_bound.argtypes = [C.POINTER(Layout)]
    lib.edgehip_layout_pair_tile_bound.restype = C.c_int64
    lib.edgehip_train_set_side_stream.argtypes = [C.c_void_p]
    lib.edgehip_train_set_side_stream.restype = C.c_int
    lib.edgehip_prof_enable.argtypes = [C.c_int]
    lib.edgehip_prof_enable.restype = C.c_int
    lib.edgehip_prof_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]
    lib.edgehip_prof_collect.restype = C.c_int
    lib.edgehip_prof_kernel_name.argtypes = [C.c_int32]
    lib.edgehip_prof_kernel_name.restype = C.c_char_p
    lib.edgehip_prof_num_kernels.restype = C.c_int32
    lib.edgehip_prof_layer_rows.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    lib.edgehip_prof_layer_rows.restype = C.c_int
    lib.edgehip_prof_layer_visits.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    lib.edgehip_prof_layer_visits.restype = C.c_int
    return lib


lib = _load()


class EdgeHipError(RuntimeError):
    pass


def check(rc, what=''):
    if rc != OK:
        raise EdgeHipError(f'{what}: {_ERR.get(rc, rc)} ({rc})')


MISC_EXPORTS = ('edgehip_edge_ce_loss_static_max_tiles', 'edgehip_train_set_side_stream', 'edgehip_layout_pair_tile_bound')
MAX_PAIR_TILES = 1 << 29
PROF_EXPORTS = ('edgehip_prof_enable', 'edgehip_prof_collect', 'edgehip_prof_kernel_name', 'edgehip_prof_num_kernels',
                'edgehip_prof_layer_rows', 'edgehip_prof_layer_visits')


def edgehip_static_tiles():
    """Capacity (pair tiles) of the static scratch of edgehip_edge_ce_loss_static."""
    return int(lib.edgehip_edge_ce_loss_static_max_tiles())


def prof_collect():
    """{kernel class: (total ms, launches)} since the last collect (needs edgehip_prof_enable(1))."""
    n = lib.edgehip_prof_num_kernels()
    ms, cnt = (C.c_double * n)(), (C.c_int64 * n)()
    check(lib.edgehip_prof_collect(ms, cnt, n), 'prof_collect')
    return {lib.edgehip_prof_kernel_name(i).decode(): (ms[i], cnt[i]) for i in range(n) if cnt[i]}


def ptr(t):
    """Device (or host) pointer of a contiguous tensor, or NULL for None."""
    if t is None:
        return None
    assert t.is_contiguous(), 'edgehip needs contiguous tensors'
    return t.data_ptr()


def stream_ptr():
    import torch
    if torch.cuda.is_available():
        return torch.cuda.current_stream().cuda_stream
    return None


# One pool of side streams per device for everything in this package (the sampler's sub-batch streams, the training
# backward's weight-gradient stream, the count read-back): a process has few hardware queues (GPU_MAX_HW_QUEUES, 4 by
# default) and HIP deals streams to them in creation order, so streams created independently by the sampler and the
# trainer ended up sharing a queue - and two streams on one queue run one after the other (three-stream sampling after
# training in the same process: 162 -> 114 graphs/s).
_SIDE_STREAMS = {}
_TRAIN_SIDE_SET = set()
MAX_SIDE_STREAMS = 3


def side_streams(dev, n):
    """The first n (<= MAX_SIDE_STREAMS) side streams of the device's pool."""
    import torch
    dev = torch.device(dev)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    have = _SIDE_STREAMS.setdefault(key, [])
    n = min(n, MAX_SIDE_STREAMS)
    while len(have) < n:
        have.append(torch.cuda.Stream(device=dev))
    return have[:n]


def use_pool_for_training(dev):
    """Hands the pool's first stream to the library as the training backward's side stream (once per device; the
    device must be current)."""
    import torch
    dev = torch.device(dev)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _TRAIN_SIDE_SET:
        check(lib.edgehip_train_set_side_stream(side_streams(dev, 1)[0].cuda_stream), 'train_set_side_stream')
        _TRAIN_SIDE_SET.add(key)
