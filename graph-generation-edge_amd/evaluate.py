"""evaluate.py - counterpart of the reference's evaluate.py (evaluate.py:20-61): load a checkpoint of a
TGNN_degree_guided run, sample graphs on the MI355X path, report their statistics.

    python graph-generation-edge_amd/evaluate.py --checkpoint_path run/check/checkpoint_5499.pt \
        --edge_list graphs/polblogs.txt --diffusion_steps 512 --noise_schedule linear --num_samples 8

Differences from the reference script, on purpose:
  * the checkpoint is read with `torch.load(..., weights_only=True)` (tensors only; `checkpoint['model']`
    is the reference's state_dict, whose names this package keeps);
  * the run's `args.pickle` and the `graphs/*.pkl` networkx pickles are NOT unpickled: model flags come
    from the command line (same names and defaults as the reference, `add_model_args`) or `--args_json`,
    training graphs from `--edge_list` text files ("i j" per line, one file per graph) or `--synthetic`;
  * statistics come from `edgehip.graph_statistics` (bit-matrix kernels) on the device; `--networkx`
    additionally builds the largest connected components like the reference (evaluate.py:54-58).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import edgehip  # noqa: E402
from edgehip import synth  # noqa: E402


def read_edge_list(path):
    e = np.loadtxt(path, dtype=np.int64, ndmin=2)
    if e.size == 0:
        raise ValueError(f'{path}: no edges')
    ids, inv = np.unique(e.reshape(-1), return_inverse=True)          # relabel to 0..n-1 (nx.convert_node_labels_to_integers)
    return len(ids), synth._finish(len(ids), inv.reshape(-1, 2))


def build_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument('--checkpoint_path', type=str, default=None, help='checkpoint_*.pt of a run (dict with key "model")')
    p.add_argument('--edge_list', type=str, nargs='*', default=[], help='training graphs as text edge lists')
    p.add_argument('--synthetic', type=str, default=None, choices=['polblogs_like', 'cora_like', 'homo_sapiens_like'],
                   help='use a fixed-seed synthetic graph of a published shape instead of --edge_list')
    p.add_argument('--args_json', type=str, default=None, help='JSON file with model flags (overrides the defaults)')
    p.add_argument('--num_samples', type=int, default=8)
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--device', type=str, default='cuda:0')
    p.add_argument('--networkx', action='store_true', help='also report LCC sizes via networkx')
    p.add_argument('--out', type=str, default=None, help='write the statistics as JSON')
    edgehip.add_model_args(p)
    p.set_defaults(arch='TGNN_degree_guided', parametrization='xt_prescribed_st', loss_type='vb_ce_xt_prescribred_st',
                   noise_schedule='linear')
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    if a.args_json:
        with open(a.args_json) as fh:
            for k, v in json.load(fh).items():
                setattr(a, k, v)
    torch.manual_seed(a.seed)
    if torch.device(a.device).type == 'cuda':
        torch.cuda.set_device(torch.device(a.device))     # the library launches on the CURRENT device's stream
    if a.synthetic:
        graphs = [getattr(synth, a.synthetic)(0)]
    elif a.edge_list:
        graphs = [read_edge_list(p) for p in a.edge_list]
    else:
        raise SystemExit('give --edge_list FILE [FILE ...] or --synthetic SHAPE')
    a.max_degree = int(max(synth.degrees(n, e).max() for n, e in graphs))        # datasets/data.py: max train degree
    a.num_node_classes, a.num_edge_classes = 2, 2
    a.degree, a.augmented_feature_dict, a.has_node_feature = True, {}, False
    sampler = edgehip.EmpiricalEmptyGraphGenerator(graphs, degree=True, device=a.device)
    model = edgehip.get_model(a, sampler)
    if a.checkpoint_path:
        ck = torch.load(a.checkpoint_path, map_location='cpu', weights_only=True)
        model.load_state_dict(ck['model'] if 'model' in ck else ck)
    model = model.to(a.device).eval()
    out = model.sample(a.num_samples, seed=a.seed)
    st = edgehip.graph_statistics(out)
    res = {k: np.asarray(v).tolist() for k, v in st.items()}
    res['nodes'] = out.nodes_per_graph.cpu().tolist()
    if a.networkx:
        import networkx as nx
        lcc = []
        for d in out.to_data_list():
            G = nx.Graph()
            G.add_nodes_from(range(d.num_nodes))
            G.add_edges_from(d.edge_index.t().tolist())
            lcc.append(len(max(nx.connected_components(G), key=len)))
        res['LCC_networkx'] = lcc
    print(json.dumps(res))
    if a.out:
        with open(a.out, 'w') as fh:
            json.dump(res, fh)
    return res, out


if __name__ == '__main__':
    main()
