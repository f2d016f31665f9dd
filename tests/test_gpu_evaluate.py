"""GPU: the evaluate.py counterpart: checkpoint -> get_model -> sample -> statistics."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

import cases
import helpers as H
from oracle import edge_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_script():
    spec = importlib.util.spec_from_file_location('edgehip_evaluate', os.path.join(ROOT, 'graph-generation-edge_amd', 'evaluate.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_evaluate_script_loads_checkpoint_and_samples(tmp_path):
    ev = _load_script()
    graphs = [g for g in cases.ragged_graphs(6) if len(g['edges']) > 3]      # an edge list cannot hold an empty graph
    files = []
    for k, g in enumerate(graphs):
        p = tmp_path / f'g{k}.txt'
        np.savetxt(p, np.asarray(g['edges']) + 7, fmt='%d')          # arbitrary labels: the script relabels
        files.append(str(p))
    T = 32
    dmax = cases.max_degree_of(graphs)
    # a "run": the reference-named state_dict inside {'model': ...}, as diffusion/base.py:158 saves it
    ref_model = H.build_model(orc.random_tgnn_params(21), T, dmax)
    ck = tmp_path / 'checkpoint_0.pt'
    torch.save({'model': ref_model.state_dict(), 'current_epoch': 1}, ck)
    argv = ['--checkpoint_path', str(ck), '--edge_list', *files, '--diffusion_steps', str(T), '--num_samples', '5',
            '--seed', '3', '--networkx', '--out', str(tmp_path / 'stats.json')]
    res, out = ev.main(argv)
    assert out.num_graphs == 5 and len(res['triangle_count']) == 5
    assert json.load(open(tmp_path / 'stats.json'))['num_edges'] == res['num_edges']
    assert sum(res['num_edges']) == out.edge_index.shape[1]
    assert all(1 <= l <= n for l, n in zip(res['LCC'], res['nodes'])) and res['LCC'] == res['LCC_networkx']
    # same weights, same seeds -> the same graphs as sampling from the in-memory model
    res2, out2 = ev.main(argv)
    assert torch.equal(out.edge_index, out2.edge_index)
    # a checkpoint with a wrong tensor shape is refused by load_state_dict, not silently ignored
    bad = {k: v for k, v in ref_model.state_dict().items()}
    bad['_denoise_fn.node_in.0.weight'] = torch.zeros(3, 3)
    torch.save({'model': bad}, tmp_path / 'bad.pt')
    with pytest.raises(RuntimeError):
        ev.main(['--checkpoint_path', str(tmp_path / 'bad.pt'), '--edge_list', *files, '--diffusion_steps', str(T)])


@pytest.mark.parametrize('order', ['load_then_cuda', 'cuda_then_load'])
def test_loaded_schedule_buffers_drive_the_kernels(order):
    """A checkpoint carries its schedule buffers (log_alpha ... ; diffusion/base.py:158,171).  The reference always
    computes from them, whatever --noise_schedule says at load time; the kernel tables here are host-built, so they
    must be rebuilt on load - in either order of .to(device) and load_state_dict."""
    import edgehip
    graphs = [g for g in cases.ragged_graphs(6) if len(g['edges']) > 3]
    T, dmax = 32, cases.max_degree_of(graphs)
    sampler = lambda: edgehip.EmpiricalEmptyGraphGenerator([(g['n'], g['edges']) for g in graphs], degree=True, device='cuda')

    def build(schedule, device):
        a = H.make_args(T, dmax, device)
        a.noise_schedule = schedule
        m = edgehip.get_model(a, sampler())
        m._denoise_fn.load_state_dict(orc.random_tgnn_params(21), strict=True)
        return m
    src = build('cosine', 'cuda').to('cuda').eval()
    sd = {k: v.cpu() for k, v in src.state_dict().items()}
    dst = build('linear', 'cuda')
    assert not torch.equal(dst.log_alpha.cpu(), sd['log_alpha'])
    if order == 'load_then_cuda':
        dst.load_state_dict(sd)
        dst = dst.to('cuda').eval()
    else:
        dst = dst.to('cuda').eval()
        dst.load_state_dict(sd)
    import random
    torch.manual_seed(5); random.seed(5)
    a = src.sample(6, seed=11)
    torch.manual_seed(5); random.seed(5)
    b = dst.sample(6, seed=11)
    assert a.edge_index.shape[1] > 0 and torch.equal(a.edge_index, b.edge_index)
    # and the loss path (q tables, prior-KL constants) follows too
    gb = lambda: H.graph_batch(graphs, degree_dtype=np.int64)
    t = torch.tensor([3, 9, 20, 30][:len(graphs)] + [1] * max(0, len(graphs) - 4))
    pt = torch.full((len(graphs),), 1.0 / T)
    src.rng_seed = dst.rng_seed = 77
    with torch.no_grad():
        la, lb = src.log_prob(gb(), t, pt), dst.log_prob(gb(), t, pt)
    assert torch.equal(la, lb)
