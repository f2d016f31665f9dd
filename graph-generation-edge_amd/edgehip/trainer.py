"""trainer.py - one optimisation step of the reference's training loop (experiment.py:23-46 of the
reference: zero_grad -> elbo_bpd -> backward -> clip -> optimizer.step), with the data-parallel
exchange of SURVEY.md 8e when a process group is up: the loss is normalised by the GLOBAL number of
pairs, gradients are summed in one flat bucket, and `log_prob` applies every rank's (t, kl) to the
importance-sampling history so replicas stay identical.  Two collectives per step: one all-gather of
[pair count, t, kl] between forward and backward, one all-reduce of the gradient arena after it."""
import math

import torch

from . import shard


def make_optimizer(model, lr=1e-3, momentum=0.9, momentum_sqr=0.999):
    """The reference default: Adam(lr 1e-3, betas (0.9, 0.999)) (diffusion/optim/multistep.py:27-46).  On the GPU
    torch's fused implementation is used (same update rule, one multi-tensor launch instead of ~13)."""
    params = list(model.parameters())
    opt = None
    if params and params[0].is_cuda:
        try:
            opt = torch.optim.Adam(params, lr=lr, betas=(momentum, momentum_sqr), fused=True)
        except (RuntimeError, TypeError):
            pass
    if opt is None:
        opt = torch.optim.Adam(params, lr=lr, betas=(momentum, momentum_sqr))
    net = getattr(model, '_denoise_fn', None)
    if net is not None and hasattr(net, 'weights_changed'):
        # the fused update does not bump the parameters' version counters: tell the inference side to refold
        opt.register_step_post_hook(lambda *_: net.weights_changed())
    return opt


def train_step(model, batch, optimizer, clip_value=None, clip_norm=None, state=None, **log_prob_kwargs):
    """Returns the global bits/dim of the step (python float is NOT forced: a 0-d tensor, no host sync).
    With `state` (a dict kept by the caller across steps) the gradients live in one flat arena: the fused
    backward adds into it, the data-parallel all-reduce runs on it in place and zeroing it is one memset."""
    from . import train
    model.train()
    batch._train_ws_busy = False          # graphs of earlier steps are dead: their workspace is free again
    params = [p for p in model.parameters()]
    arena = None
    if state is not None and getattr(model, 'train_form', 'fused') == 'fused':
        arena = state.get('arena')
        if arena is None or not arena.matches(params):
            arena = state['arena'] = train.GradArena(params)
        arena.zero_and_attach()
    else:
        optimizer.zero_grad(set_to_none=True)
    with train.use_arena(arena):
        neg_vb = model.log_prob(batch, **log_prob_kwargs)                   # [B], differentiable
        # global pair count: gathered together with (t, kl) inside log_prob (one collective), no host sync
        P = batch.__dict__.pop('_global_entries', None)
        if P is None:
            P = shard.sum_over_ranks_device(batch.num_entries, batch.device)
        loss = -neg_vb.sum() / (math.log(2) * P)                            # diffusion/loss.py:10-12
        loss.backward()
    if arena is not None:
        shard.allreduce_flat(arena.flat)
    else:
        bucket = shard.allreduce_gradients(params, None if state is None else state.get('bucket'))
        if state is not None:
            state['bucket'] = bucket
    if clip_value and arena is not None:
        arena.flat.clamp_(-clip_value, clip_value)          # every p.grad is a view of the arena: one launch
    elif clip_value:
        torch.nn.utils.clip_grad_value_(params, clip_value)
    if clip_norm:
        torch.nn.utils.clip_grad_norm_(params, clip_norm)
    optimizer.step()
    net = getattr(model, '_denoise_fn', None)
    if net is not None and hasattr(net, 'weights_changed'):
        net.weights_changed()               # whatever optimizer this was: inference refolds the weights
    return loss.detach()
