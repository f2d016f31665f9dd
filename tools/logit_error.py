"""Logit error of the inference kernels against the reference's golden logits (fp32 CPU) on the golden loss
configurations: max |diff| / max |ref| and rms.  Used to compare MFMA formulations (EH_BF16X3 = 1 / 0 builds)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'graph-generation-edge_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cases  # noqa: E402
import helpers as H  # noqa: E402
from oracle import edge_oracle as orc  # noqa: E402


def main():
    for name, case in (('loss_toy_mid', 'toy'), ('loss_ragged', 'ragged'), ('loss_polblogs1', 'polblogs1')):
        g = H.golden(name + '.npz')
        builder, T, wseed, useed = cases.CASES[case]
        graphs = builder()
        rng = np.random.default_rng(useed + 1)
        Nn = sum(x['n'] for x in graphs)
        P = sum(x['n'] * (x['n'] - 1) // 2 for x in graphs)
        us = [cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2), cases.uniforms(rng, Nn, 2), cases.uniforms(rng, P, 2)]
        t, pt = torch.from_numpy(g['t']), torch.from_numpy(g['pt'])
        model = H.build_model(orc.random_tgnn_params(wseed), T, int(g['max_degree']))
        want = {}
        with torch.no_grad():
            model.log_prob(H.graph_batch(graphs, degree_dtype=np.int64), t, pt, us[1].cuda(), us[3].cuda(), want)
        ref = torch.from_numpy(g['logits']).double()
        got = want['logits'].double().cpu()
        if ref.numel() == 0:
            continue
        d = (got - ref).abs()
        print(f'{name:16s} A={ref.shape[0]:6d}  max|diff|/max|ref| = {float(d.max() / ref.abs().max()):.3e}   rms rel = '
              f'{float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e}')


if __name__ == '__main__':
    main()
