"""CPU: host-side product code (tables, Philox, C-ABI surface) - no GPU compute calls."""
import os
import re

import numpy as np
import pytest
import torch

import cases
from helpers import golden, philox4x32, u24


def test_product_tables_match_reference_golden():
    from edgehip import tables
    g = golden('schedule.npz')
    for kind, fn in (('linear', tables.linear_beta_schedule), ('cosine', tables.cosine_beta_schedule),
                     ('Tt1', tables.Tt1_beta_schedule)):
        for T in (32, 64, 128, 512):
            buf = tables.schedule_buffers(T, fn, [1 - 1e-12, 1e-12], [1 - 1e-12, 1e-12])
            for nm in ('log_alpha', 'log_1_min_alpha', 'log_cumprod_alpha', 'log_1_min_cumprod_alpha'):
                assert np.array_equal(buf[nm].numpy(), g[f'{kind}_{T}_{nm}']), (kind, T, nm)
            assert np.array_equal(buf['log_final_prob_edge'].numpy(), g['log_final_prob_edge'])
    for T in (32, 48, 512):
        assert np.array_equal(tables.pos_emb_table(T, 64).numpy(), g[f'posemb_{T}'])


def test_posterior_table_matches_reference_golden():
    from edgehip import tables
    g = golden('posterior.npz')
    buf = tables.schedule_buffers(512, tables.linear_beta_schedule, [1 - 1e-12, 1e-12], [1 - 1e-12, 1e-12])
    nmax = g['table'].shape[1] - 1
    tab = tables.posterior_table(buf, nmax).numpy()
    for k, t in enumerate(g['t'].tolist()):
        assert np.array_equal(tab[t], g['table'][k]), t


def test_q_tables_and_sparse_guard():
    from edgehip import tables
    from oracle import edge_oracle as orc
    buf = tables.schedule_buffers(64, tables.linear_beta_schedule, [1 - 1e-12, 1e-12], [1 - 1e-12, 1e-12])
    q_cum, q_one = tables.q_tables(buf)
    s = orc.make_schedule(64)
    one = orc.index_to_log_onehot(torch.ones(64, dtype=torch.long))
    t = torch.arange(64)
    assert torch.equal(q_cum, orc.q_pred(s, one, t, s.log_final_prob_edge))
    assert torch.equal(q_one, orc.q_pred_one_timestep(s, one, t, s.log_final_prob_edge))
    dense = tables.schedule_buffers(64, tables.linear_beta_schedule, [1 - 1e-12, 1e-12], [0.9, 0.1])
    with pytest.raises(NotImplementedError):
        tables.q_tables(dense)


def test_gumbel_lut_is_the_reference_transform():
    from edgehip import tables
    lut = tables.gumbel_lut()
    k = torch.tensor([0, 1, 12345, (1 << 24) - 1])
    u = k.float() * 2.0 ** -24
    assert torch.equal(lut[k], -torch.log(-torch.log(u + 1e-30) + 1e-30))
    assert float(lut.max() - lut.min()) < tables.GUMBEL_MAX_GAP + 1e-3
    assert float(lut.max() - lut.min()) > tables.GUMBEL_MAX_GAP - 1e-3


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32 10 rounds."""
    x = philox4x32(0, 0, 0, 0, 0)
    assert [int(v) for v in x] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    x = philox4x32(0xffffffffffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(v) for v in x] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    x = philox4x32(0x299f31d0a4093822, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344)
    assert [int(v) for v in x] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    u = u24(np.array([0, 0xffffffff], dtype=np.uint32))
    assert u[0] == 0.0 and u[1] == np.float32(1 - 2.0 ** -24)


def test_library_exports_every_declared_symbol():
    import ctypes
    from edgehip import capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, 'include', 'edgehip.h')).read()
    names = set(re.findall(r'\b(edgehip_[a-z0-9_]+)\s*\(', hdr))
    assert len(names) >= 18
    lib = ctypes.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/edgehip.h but not exported'
    assert set(capi.EXPORTS) | set(capi.SIZE_EXPORTS) | set(capi.PROF_EXPORTS) | set(capi.MISC_EXPORTS) | {'edgehip_version'} == names
    assert b'gfx950' in capi.lib.edgehip_version()


def test_struct_sizes_match_header():
    import ctypes
    from edgehip import capi
    assert ctypes.sizeof(capi.Layout) == 6 * 4 + 9 * 8
    assert ctypes.sizeof(capi.Counts) == 24
    assert ctypes.sizeof(capi.GnnParams) == 12 * 4 + (23 + 10 * 8 + 13) * 8


def test_argument_errors_without_gpu():
    """NULL pointers are rejected before any launch (EDGEHIP_ERR_ARG)."""
    import ctypes
    from edgehip import capi
    L = capi.Layout()
    assert capi.lib.edgehip_degree_from_adj(ctypes.byref(L), None, None, None) == capi.ERR_ARG
    assert capi.lib.edgehip_linear_fwd(None, 4, 64, None, None, 64, 0, None, None) == capi.ERR_ARG
    P = capi.GnnParams()
    P.dim = 32
    assert capi.lib.edgehip_gnn_folded_floats(ctypes.byref(P)) == 0
    assert capi.lib.edgehip_gnn_prepare(ctypes.byref(P), None, None) == capi.ERR_UNSUPPORTED


def test_get_model_flag_surface():
    """Same dispatch and error types as the reference factory (model.py:36-45, active.py:225-226)."""
    import edgehip
    from helpers import make_args
    a = make_args(64, 10, device='cpu')
    m = edgehip.get_model(a, None)
    names = set(m.state_dict().keys())
    from oracle import edge_oracle as orc
    want = {'_denoise_fn.' + k for k in orc.tgnn_param_shapes()} | {
        'log_alpha', 'log_1_min_alpha', 'log_cumprod_alpha', 'log_1_min_cumprod_alpha', 'log_final_prob_node',
        'log_final_prob_edge', 'Lt_history', 'Lt_count'}
    assert names == want
    assert sum(p.numel() for p in m._denoise_fn.parameters()) == 1123650
    for k, shp in orc.tgnn_param_shapes().items():
        assert tuple(m.state_dict()['_denoise_fn.' + k].shape) == tuple(shp), k
    a.arch = 'nope'
    with pytest.raises(NotImplementedError):
        edgehip.get_model(a, None)
    a.arch, a.parametrization = 'TGNN_degree_guided', 'x0'
    with pytest.raises(AssertionError):
        edgehip.get_model(a, None)
    a.parametrization, a.loss_type = 'xt_prescribed_st', 'bogus'
    with pytest.raises(ValueError):
        edgehip.get_model(a, None).log_prob(None)
    a.loss_type, a.noise_schedule = 'vb_ce_xt_prescribred_st', 'bogus'
    with pytest.raises(NotImplementedError):
        edgehip.get_model(a, None)


def test_synthetic_polblogs_shape():
    from edgehip import synth
    n, e = synth.polblogs_like(0)
    d = synth.degrees(n, e)
    assert n == 1222 and len(e) == 16714 and abs(int(d.max()) - 351) <= 12
    assert (e[:, 0] < e[:, 1]).all()
    n2, e2 = synth.polblogs_like(0)
    assert np.array_equal(e, e2)


def test_degree_only_statistics_against_the_reference_fixture():
    """power law / gini / edge-distribution entropy are host formulas over the device's degree vector
    (eval_utils/graph_statistics.py:146-199); here on degrees computed with numpy from the same graphs."""
    import cases
    from edgehip import stats, synth
    g = np.load(os.path.join(cases.GOLDEN_DIR, 'stats.npz'), allow_pickle=False)
    keys = [str(k) for k in g['keys']]
    fns = {'power_law_exp': stats.power_law_alpha, 'gini': stats.gini, 'rel_edge_distr_entropy': stats.edge_distribution_entropy}
    n = 0
    for gi, x in enumerate(cases.stats_graphs()):
        d = synth.degrees(x['n'], x['edges']) if len(x['edges']) else np.zeros(x['n'])
        for k, f in fns.items():
            ref = g['values'][gi, keys.index(k)]
            got = f(d)
            if np.isfinite(ref):
                assert abs(got - ref) <= 1e-12 * max(1.0, abs(ref)), (gi, k, got, ref)
                n += 1
            else:
                assert not np.isfinite(got) or (k == 'gini' and x['n'] == 1), (gi, k, got, ref)
    assert n >= 18


def test_empty_graph_sampler_draws_like_the_reference_under_a_seed():
    """tests/golden/sampler.npz: the reference's EmpiricalEmptyGraphGenerator (datasets/data_utils.py:78-147) seeded
    with torch.manual_seed + random.seed.  Same sizes, same chosen degree sequences, same batch fields - over two
    consecutive sample() calls, so the number of RNG draws per call matches too."""
    import random
    import cases
    import edgehip
    g = np.load(os.path.join(cases.GOLDEN_DIR, 'sampler.npz'), allow_pickle=False)
    train = [(x['n'], x['edges']) for x in cases.sampler_train_graphs()]
    gen = edgehip.EmpiricalEmptyGraphGenerator(train, degree=True, device='cpu')
    assert gen.min_node == int(g['min_node']) and gen.max_node == int(g['max_node'])
    assert np.array_equal(gen.empirical_graph_size_dist.numpy(), g['size_dist'])
    torch.manual_seed(cases.SAMPLER_SEED)
    random.seed(cases.SAMPLER_SEED)
    for tag, k in (('1', cases.SAMPLER_DRAWS), ('2', 7)):
        b = gen.sample(k)
        assert np.array_equal(b.nodes_per_graph.numpy(), g['nodes_per_graph_' + tag])
        assert np.array_equal(b.edges_per_graph.numpy(), g['edges_per_graph_' + tag])
        assert str(b.degree.dtype) == str(g['degree_dtype']) and np.array_equal(b.degree.numpy(), g['degree_' + tag])
        assert b.num_graphs == k and int(b.adj.abs().sum()) == 0                   # x_T input: empty graphs


def test_pair_tile_bound_refuses_layouts_that_could_wrap_the_32_bit_counters():
    """include/edgehip.h: tptr / counts->pair_tiles are int32; the offsets entry point refuses a layout whose worst-case
    tile count exceeds EDGEHIP_MAX_PAIR_TILES on the HOST, before any launch (so this runs without a GPU)."""
    import ctypes as C
    from edgehip import capi
    n = 20000
    per = (n * (n - 1) // 2 + capi.PAIR_TILE - 1) // capi.PAIR_TILE + 1
    L = capi.Layout()
    L.num_graphs, L.max_nodes = 86, n
    assert capi.lib.edgehip_layout_pair_tile_bound(C.byref(L)) == 86 * per > capi.MAX_PAIR_TILES
    junk = (C.c_int64 * 4)()
    p = C.cast(junk, C.c_void_p)
    assert capi.lib.edgehip_active_offsets(C.byref(L), p, p, p, p, p, None) == capi.ERR_TOO_LARGE
    L.num_graphs = 64
    L.max_nodes = 1222                                   # the bench's batch: 64 polblogs graphs
    assert 0 < capi.lib.edgehip_layout_pair_tile_bound(C.byref(L)) < capi.edgehip_static_tiles()
    assert capi.lib.edgehip_layout_pair_tile_bound(None) == -1
