"""Where does the 2e-4 bound on the gradients come from?  For the polblogs golden case: d elbo_bpd / d theta from the
product (fused training form), from torch autograd through the fp32 oracle (what the tests compare with), and from
autograd through the SAME oracle run in fp64 on the same noised states and active sets.  Prints, per tensor class,
max |a - b| / max |b| for product-vs-fp64, oracle32-vs-fp64 and product-vs-oracle32.
    python tools/grad_error_probe.py [out.json]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'graph-generation-edge_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import cases  # noqa: E402
import helpers as H  # noqa: E402
from oracle import edge_oracle as orc  # noqa: E402


def main():
    g = H.golden('loss_polblogs1.npz')
    builder, T, wseed, useed = cases.CASES['polblogs1']
    graphs = builder()
    rng = np.random.default_rng(useed + 1)
    Nn = sum(x['n'] for x in graphs)
    P = sum(x['n'] * (x['n'] - 1) // 2 for x in graphs)
    us = [cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2), cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)]
    t, pt = torch.from_numpy(g['t']), torch.from_numpy(g['pt'])
    params = orc.random_tgnn_params(wseed)
    dmax = int(g['max_degree'])
    # fp32 oracle
    p32 = {k: v.clone().requires_grad_() for k, v in params.items()}
    o = orc.EdgeOracle(p32, T, dmax)
    ob = orc.make_batch(graphs, degree_dtype=torch.int64)
    ob.log_node_attr = orc.index_to_log_onehot(ob.node_attr)
    ob.log_full_edge_attr = orc.index_to_log_onehot(ob.full_edge_attr)
    orc.elbo_bpd(o.loss(ob, t, pt, orc.UniformTape(us), training=True), ob.num_entries).backward()
    # fp64: the same noised states / active sets (ob keeps them), network + CE in double
    torch.set_default_dtype(torch.float64)
    p64 = {k: v.double().clone().requires_grad_() for k, v in params.items()}
    ob.degree = ob.degree.double()
    t_node = t.repeat_interleave(ob.nodes_per_graph)
    logits = orc.tgnn_forward(p64, ob, t_node, dmax, T)
    lp = F.log_softmax(logits, dim=1)
    tgt = ob.log_full_edge_attr_tmin1.index_select(0, ob.active_edge_indices).double()
    ce = -(tgt.exp() * lp).sum(dim=1)
    gidx = ob.batch[ob.full_edge_index[0].index_select(0, ob.active_edge_indices)]
    kl = torch.zeros(ob.num_graphs).index_add_(0, gidx, ce)
    (((kl / pt.double()).sum()) / (np.log(2) * ob.num_entries)).backward()
    torch.set_default_dtype(torch.float32)
    # product
    model = H.build_model(params, T, dmax)
    model.train()
    b = H.graph_batch(graphs, degree_dtype=np.int64)
    nv = model.log_prob(b, t, pt, us[1].cuda(), us[3].cuda())
    (-nv.sum() / (np.log(2) * b.num_entries)).backward()
    named = dict(model._denoise_fn.named_parameters())
    rows, worst = [], dict(product_vs_fp64=0.0, oracle32_vs_fp64=0.0, product_vs_oracle32=0.0)
    gmax = max(float(r.grad.abs().max()) for r in p64.values() if r.grad is not None)
    for k, r64 in p64.items():
        if r64.grad is None or float(r64.grad.abs().max()) == 0.0:
            continue
        scale = max(float(r64.grad.abs().max()), 1e-6 * gmax)
        a = named[k].grad.detach().cpu().double()
        e = dict(name=k, product_vs_fp64=float((a - r64.grad).abs().max()) / scale,
                 oracle32_vs_fp64=float((p32[k].grad.double() - r64.grad).abs().max()) / scale,
                 product_vs_oracle32=float((a - p32[k].grad.double()).abs().max()) / scale)
        rows.append(e)
        for kk in worst:
            worst[kk] = max(worst[kk], e[kk])
    rows.sort(key=lambda e: -e['product_vs_fp64'])
    res = dict(case='loss_polblogs1 (n 1222, T 512, 86 parameter tensors)', worst=worst, worst_tensors=rows[:6],
               median_product_vs_fp64=float(np.median([e['product_vs_fp64'] for e in rows])),
               median_oracle32_vs_fp64=float(np.median([e['oracle32_vs_fp64'] for e in rows])))
    s = json.dumps(res, indent=1)
    print(s)
    if len(sys.argv) > 1:
        open(sys.argv[1], 'w').write(s)


if __name__ == '__main__':
    main()
