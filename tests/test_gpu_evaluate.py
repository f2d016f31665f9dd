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
    assert all(1 <= l <= n for l, n in zip(res['LCC'], res['nodes']))
    # same weights, same seeds -> the same graphs as sampling from the in-memory model
    res2, out2 = ev.main(argv)
    assert torch.equal(out.edge_index, out2.edge_index)
    # a checkpoint with a wrong tensor shape is refused by load_state_dict, not silently ignored
    bad = {k: v for k, v in ref_model.state_dict().items()}
    bad['_denoise_fn.node_in.0.weight'] = torch.zeros(3, 3)
    torch.save({'model': bad}, tmp_path / 'bad.pt')
    with pytest.raises(RuntimeError):
        ev.main(['--checkpoint_path', str(tmp_path / 'bad.pt'), '--edge_list', *files, '--diffusion_steps', str(T)])
