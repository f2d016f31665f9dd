"""Which logarithm does the host's torch use?  Compares torch.log (fp32) over the 2^24 uniform grid with the correctly
rounded value (fp64 log rounded to fp32) and with numpy's fp32 log; prints the CPU and torch's CPU capability."""
import subprocess
import numpy as np
import torch
print(subprocess.run('lscpu | head -20', shell=True, capture_output=True, text=True).stdout)
print('capability', torch.backends.cpu.get_cpu_capability(), 'mkl', torch.backends.mkl.is_available(), 'threads', torch.get_num_threads())
u = torch.arange(1 << 24, dtype=torch.float32) * (2.0 ** -24)
a = u + 1e-30
l1 = torch.log(a)
cr = torch.log(a.double()).float()
d = (l1.view(torch.int32) - cr.view(torch.int32)).abs()
print('inner log: torch fp32 vs correctly rounded: differ', int((d != 0).sum()), 'max ulps', int(d.max()))
npl = torch.from_numpy(np.log(a.numpy()))
print('inner log: torch vs numpy fp32:', int((npl != l1).sum()), ' numpy vs CR:', int((npl != cr).sum()))
for nt in (1,):
    torch.set_num_threads(nt)
    l1b = torch.log(a)
    print('threads', nt, 'same as default threads:', bool(torch.equal(l1b, l1)))
# small tensors take another code path? (vectorised tail / non-MKL)
small = torch.log(a[:1000].clone())
print('first 1000 alone equal:', bool(torch.equal(small, l1[:1000])))
st = torch.log(a[::7][:100000].contiguous())
print('strided subset equal:', bool(torch.equal(st, l1[::7][:100000])))
b = -l1 + 1e-30
g = -torch.log(b)
gcr = -torch.log((-cr + 1e-30).double()).float()
print('composite differ from CR composite:', int((g != gcr).sum()))
