"""Host-built constant tables of the diffusion path.

Everything here depends only on (T, schedule, max degree) and is built ONCE with the same
torch CPU op sequence the reference uses, so the numbers the device kernels consume are
bit-identical to what the reference would compute per element:

  schedule buffers   diffusion/diffusion_binomial_vanilla.py:21-45, diffusion_base.py:84-111
  posterior table    diffusion/diffusion_binomial_active.py:97-119  (function of (t, d0 - d_t))
  q tables           diffusion_binomial_vanilla.py:272-288 / 226-243 for a source bit 1
  sinusoid table     layers/layers.py:45-60
  Gumbel LUT         diffusion_base.py:177 for the 2^24 uniforms torch.rand can emit
"""
import math

import numpy as np
import torch

GUMBEL_MAX_GAP = 20.8708   # max g(u1) - g(u0) over the 2^-24 grid


def linear_beta_schedule(timesteps):
    scale = 1000 / timesteps
    return 1 - torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64).numpy()


def cosine_beta_schedule(timesteps, s=0.008):
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    return np.sqrt(np.clip(ac[1:] / ac[:-1], a_min=0.001, a_max=1.))


def Tt1_beta_schedule(timesteps):
    return 1 / torch.linspace(1 + 1e-8, timesteps + 1e-8, timesteps, dtype=torch.float64).flip(0).numpy()


def _log_1_min_a(a):
    return torch.log(1 - a.exp() + 1e-40)


def _log_add_exp(a, b):
    m = torch.max(a, b)
    return m + torch.log(torch.exp(a - m) + torch.exp(b - m))


def schedule_buffers(timesteps, noise_schedule, final_prob_node, final_prob_edge):
    alphas = torch.tensor(noise_schedule(timesteps).astype('float64'))
    log_alpha = torch.from_numpy(np.log(alphas.numpy()))
    log_cumprod_alpha = torch.from_numpy(np.cumsum(log_alpha.numpy()))
    log_1_min_alpha = _log_1_min_a(log_alpha)
    log_1_min_cumprod_alpha = _log_1_min_a(log_cumprod_alpha)
    assert _log_add_exp(log_alpha, log_1_min_alpha).abs().sum().item() < 1.e-5
    assert _log_add_exp(log_cumprod_alpha, log_1_min_cumprod_alpha).abs().sum().item() < 1e-5
    return dict(
        log_alpha=log_alpha.float(), log_1_min_alpha=log_1_min_alpha.float(),
        log_cumprod_alpha=log_cumprod_alpha.float(), log_1_min_cumprod_alpha=log_1_min_cumprod_alpha.float(),
        log_final_prob_node=torch.tensor(list(final_prob_node))[None, :].log().float(),
        log_final_prob_edge=torch.tensor(list(final_prob_edge))[None, :].log().float())


def posterior_table(buf, nmax):
    """[T, nmax+1, 2] fp32: log P(node unchanged / changes | t, n = max(d0 - d_t, 0))."""
    T = buf['log_alpha'].shape[0]
    t = torch.arange(T).repeat_interleave(nmax + 1)
    n = torch.arange(nmax + 1, dtype=torch.float32).repeat(T)
    tm1 = torch.where(t - 1 < 0, torch.zeros_like(t), t - 1)
    logprob = buf['log_1_min_alpha'].gather(-1, t) + buf['log_cumprod_alpha'].gather(-1, tm1) \
        - buf['log_1_min_cumprod_alpha'].gather(-1, t)
    logit = logprob - _log_1_min_a(logprob)
    n_trials = torch.max(n - torch.zeros_like(n), torch.zeros_like(n))
    nochange = torch.distributions.Binomial(total_count=n_trials, logits=logit).log_prob(torch.zeros_like(n))
    change = _log_1_min_a(nochange)
    un = torch.stack([nochange, change], dim=1)
    out = un - un.logsumexp(1, keepdim=True)
    return out.view(T, nmax + 1, 2).contiguous()


def q_tables(buf):
    """Log-probs of a SOURCE BIT 1 under q(x_t|x_0) and q(x_t|x_{t-1}); each [T,2].

    Also checks that a source bit 0 can never flip to 1, which is what lets the
    kernels visit only the E0 true edges."""
    T = buf['log_alpha'].shape[0]
    lf = buf['log_final_prob_edge']
    one = torch.log(torch.tensor([[1e-30, 1.0]])).expand(T, 2)    # log(clamp(one_hot(1), 1e-30))
    zero = torch.log(torch.tensor([[1.0, 1e-30]])).expand(T, 2)
    out = []
    for a, b in ((buf['log_cumprod_alpha'], buf['log_1_min_cumprod_alpha']),
                 (buf['log_alpha'], buf['log_1_min_alpha'])):
        q1 = _log_add_exp(one + a[:, None], b[:, None] + lf)
        q0 = _log_add_exp(zero + a[:, None], b[:, None] + lf)
        if not bool(((q0[:, 1] - q0[:, 0]) + GUMBEL_MAX_GAP < -1.0).all()):
            raise NotImplementedError('final_prob_edge lets absent pairs appear under forward noising; '
                                      'the sparse edge-set formulation needs p(edge) <= ~1e-10')
        out.append(q1.contiguous())
    return out[0], out[1]


def pos_emb_table(T, dim, rescale_steps=4000.0):
    x = torch.arange(T) / float(T) * float(rescale_steps)
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half) * -emb)
    emb = x[:, None] * emb[None, :]
    return torch.cat((emb.sin(), emb.cos()), dim=-1).contiguous()


_LUT = None


def gumbel_lut():
    """fp32[2^24]: -log(-log(u + 1e-30) + 1e-30) for u = k * 2^-24, with the host's torch ops."""
    global _LUT
    if _LUT is None:
        u = torch.arange(1 << 24, dtype=torch.float32) * (2.0 ** -24)
        _LUT = (-torch.log(-torch.log(u + 1e-30) + 1e-30)).contiguous()
    return _LUT
