"""One-rank RCCL self-test on the GPU box: the collectives bench.py and the data-parallel trainer issue under N > 1
(barrier, MAX all-reduce of the elapsed time, fp64 all-gather of the step scalars, in-place all-reduce of the flat
gradient arena), through the same edgehip.shard functions, with backend nccl and world size 1."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'graph-generation-edge_amd'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0')
os.environ.setdefault('WORLD_SIZE', '1')
import torch  # noqa: E402
import torch.distributed as td  # noqa: E402


def main():
    torch.cuda.set_device(0)
    td.init_process_group('nccl', device_id=torch.device('cuda', 0))
    td.barrier()
    t = torch.tensor([1.25], dtype=torch.float64, device='cuda')
    td.all_reduce(t, op=td.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    arena = torch.ones(1123650, device='cuda')
    td.all_reduce(arena)
    assert float(arena.sum().item()) == 1123650.0
    parts = [torch.zeros(9, dtype=torch.float64, device='cuda')]
    td.all_gather(parts, torch.arange(9, dtype=torch.float64, device='cuda'))
    assert parts[0].tolist() == list(range(9))
    from edgehip import shard
    assert shard.max_over_ranks(2.5, 'cuda') == 2.5
    td.barrier()
    torch.cuda.synchronize()
    td.destroy_process_group()
    print('rccl self-test ok: backend', 'nccl', 'torch', torch.__version__)


if __name__ == '__main__':
    main()
