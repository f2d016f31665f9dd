"""Helper for a kernel trace of trained-network sampling: trains briefly, then ONE sample_batch pass with `parts` streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'graph-generation-edge_amd'))
import numpy as np, torch, bench, edgehip
from edgehip import synth
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, edges = synth.polblogs_like(0)
model, deg = bench.make_model(n, edges, 'cuda')
tb = edgehip.GraphBatch([n] * 4, degree=[deg] * 4, edges=[edges] * 4, device='cuda', degree_dtype=np.int64)
model.rng_seed = 1
torch.manual_seed(0)
opt = edgehip.make_optimizer(model)
state = {}
for i in range(3000):
    edgehip.train_step(model, tb, opt, clip_value=1.0, state=state)
torch.cuda.synchronize()
model.eval()
batch = edgehip.GraphBatch([n] * 64, degree=[deg] * 64, device='cuda')
model.sample_batch(batch, seed=1, parts=parts)
torch.cuda.synchronize()
t0 = time.perf_counter()
model.sample_batch(batch, seed=2, parts=parts)
torch.cuda.synchronize()
print('pass s', time.perf_counter() - t0)
