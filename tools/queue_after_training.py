import os, sys, time
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/graph-generation-edge_amd')
import numpy as np, torch, bench, edgehip
from edgehip import synth
n, edges = synth.polblogs_like(0)
model, deg = bench.make_model(n, edges, 'cuda')
B=64
batch = edgehip.GraphBatch([n]*B, degree=[deg]*B, device='cuda')
def rate(parts):
    model.sample_batch(batch, seed=1, parts=parts); torch.cuda.synchronize()
    t0=time.perf_counter()
    for i in range(2): model.sample_batch(batch, seed=2+i, parts=parts)
    torch.cuda.synchronize()
    return B*2/(time.perf_counter()-t0)
print('before training: parts 3', rate(3), 'parts 2', rate(2))
tb = edgehip.GraphBatch([n]*4, degree=[deg]*4, edges=[edges]*4, device='cuda', degree_dtype=np.int64)
tm, _ = bench.make_model(n, edges, 'cuda')
opt = edgehip.make_optimizer(tm); st={}
for _ in range(3): edgehip.train_step(tm, tb, opt, clip_value=1.0, state=st)
torch.cuda.synchronize()
print('after 3 training steps of ANOTHER model in this process: parts 3', rate(3), 'parts 2', rate(2))
