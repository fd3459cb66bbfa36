"""Oracle: forward noising, rounding ("clamp") and the reverse-diffusion steps/loops.

TEST INFRASTRUCTURE (see oracle/__init__.py).  torch-CPU fp32, tables float64
cast to float32 at lookup exactly like the reference's _extract_into_tensor
(MuseDiffusion/models/diffusion.py:904-917).  Random draws are made with the
same torch calls, in the same order, as the reference, so a seeded CPU run of
the oracle consumes the generator identically.

`d` is the namespace built by oracle.schedule.make_diffusion; `model_fn(x, ts)`
is the raw denoiser (it receives the *rescaled* float timesteps).
"""
import math

import numpy as np
import torch


def extract(arr, t, shape):
    """diffusion.py:904-917."""
    res = torch.tensor(np.asarray(arr), dtype=torch.float)[t]
    while res.dim() < len(shape):
        res = res[..., None]
    return res.expand(shape)


def model_timesteps(d, t):
    """_WrappedModel.__call__ (diffusion.py:1027-1032): map through timestep_map, rescale to float."""
    ts = torch.tensor(d.timestep_map, dtype=t.dtype)[t]
    if d.rescale_timesteps:
        ts = ts.float() * (1000.0 / d.original_num_steps)
    return ts


def q_sample(d, x_start, t, noise=None, mask=None):
    """diffusion.py:229-255."""
    if noise is None:
        noise = torch.randn_like(x_start)
    x_t = (extract(d.sqrt_alphas_cumprod, t, x_start.shape) * x_start
           + extract(d.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)
    if mask is None:
        return x_t
    mask = torch.broadcast_to(mask.unsqueeze(-1), x_start.shape)
    return torch.where(mask == 0, x_start, x_t)


def nearest_token(emb_w, x):
    """rounding.py:21-28: argmax_v -(|W_v|^2 + |x_n|^2 - 2 W x^T), distance clamped at 0."""
    flat = x.reshape(-1, x.size(-1))
    wn = (emb_w ** 2).sum(-1).view(-1, 1)
    xn = (flat ** 2).sum(-1).view(-1, 1)
    dist = wn + xn.transpose(0, 1) - 2.0 * torch.mm(emb_w, flat.transpose(0, 1))
    dist = torch.clamp(dist, 0.0, math.inf)
    return torch.max(-dist, dim=0).indices


def round_to_embedding(emb_w, x):
    """rounding.py:31-47 with dist=None: snap every position to its nearest embedding row."""
    return emb_w[nearest_token(emb_w, x)].view(x.shape)


def p_mean_variance(d, model_fn, x, t, clip_denoised=True, emb_w=None):
    """diffusion.py:280-347.  `emb_w` not None <=> denoised_fn = denoised_fn_round active."""
    out = model_fn(x, model_timesteps(d, t))
    variance = extract(d.model_variance, t, x.shape)
    log_variance = extract(d.model_log_variance, t, x.shape)
    if d.predict_xstart:
        x0 = out
    else:
        x0 = (extract(d.sqrt_recip_alphas_cumprod, t, x.shape) * x
              - extract(d.sqrt_recipm1_alphas_cumprod, t, x.shape) * out)
    if emb_w is not None:
        x0 = round_to_embedding(emb_w, x0)
    if clip_denoised:
        x0 = x0.clamp(-1, 1)
    mean = (extract(d.posterior_mean_coef1, t, x.shape) * x0
            + extract(d.posterior_mean_coef2, t, x.shape) * x)
    return dict(mean=mean, variance=variance, log_variance=log_variance, pred_xstart=x0,
                model_output=out)


def truncated_noise(x, top_p):
    """diffusion.py:378-388: redraw every |z| > top_p until none is left."""
    noise = torch.randn_like(x)
    if top_p is not None and top_p > 0:
        bad = noise.abs() > top_p
        while bad.any():
            noise[bad] = torch.randn_like(noise[bad])
            bad = noise.abs() > top_p
    return noise


def p_sample(d, model_fn, x, t, clip_denoised=True, emb_w=None, top_p=None, mask=None,
             x_start=None, noise=None):
    """diffusion.py:349-404.  `noise` injects the (already truncated) draw."""
    out = p_mean_variance(d, model_fn, x, t, clip_denoised, emb_w)
    if noise is None:
        noise = truncated_noise(x, top_p)
    nz = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
    sample = out["mean"] + nz * torch.exp(0.5 * out["log_variance"]) * noise
    if mask is not None:
        sample = torch.where(mask == 0, x_start, sample)
    return dict(sample=sample, pred_xstart=out["pred_xstart"], greedy_mean=out["mean"], out=out)


def ddim_sample(d, model_fn, x, t, clip_denoised=True, emb_w=None, eta=0.0, mask=None,
                x_start=None, noise=None):
    """diffusion.py:701-757."""
    out = p_mean_variance(d, model_fn, x, t, clip_denoised, emb_w)
    x0 = out["pred_xstart"]
    eps = ((extract(d.sqrt_recip_alphas_cumprod, t, x.shape) * x - x0)
           / extract(d.sqrt_recipm1_alphas_cumprod, t, x.shape))
    ab = extract(d.alphas_cumprod, t, x.shape)
    ab_prev = extract(d.alphas_cumprod_prev, t, x.shape)
    sigma = eta * torch.sqrt((1 - ab_prev) / (1 - ab)) * torch.sqrt(1 - ab / ab_prev)
    if noise is None:
        noise = torch.randn_like(x)
    mean_pred = x0 * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev - sigma ** 2) * eps
    nz = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
    sample = mean_pred + nz * sigma * noise
    if mask is not None:
        sample = torch.where(mask == 0, x_start, sample)
    return dict(sample=sample, pred_xstart=x0, out=out)


def p_loop_indices(T, t_enc=None):
    """diffusion.py:508."""
    return list(range(T))[::-1][slice(t_enc)]


def ddim_loop_indices(T, gap=1, t_enc=None):
    """diffusion.py:878."""
    return list(range(T))[::-1][::gap][slice(t_enc)]


def p_sample_loop(d, model_fn, shape, noise, clip_denoised=True, emb_w=None, top_p=None,
                  clamp_step=None, clamp_first=None, mask=None, x_start=None, t_enc=None,
                  step_noise=None, trace=None):
    """diffusion.py:475-540 (progressive loop) reduced to the final sample.

    step_noise: optional callable (k, i, x) -> injected noise for iteration k / timestep i.
    """
    x = noise
    for k, i in enumerate(p_loop_indices(d.num_timesteps, t_enc)):
        t = torch.tensor([i] * shape[0])
        if not clamp_first:
            use = None if i > clamp_step else emb_w
        else:
            use = emb_w if i >= clamp_step else None
        nz = step_noise(k, i, x) if step_noise is not None else None
        out = p_sample(d, model_fn, x, t, clip_denoised, use, top_p, mask, x_start, noise=nz)
        if trace is not None:
            trace.append(out)
        x = out["sample"]
    return x


def ddim_sample_loop(d, model_fn, shape, noise, clip_denoised=True, emb_w=None, mask=None,
                     x_start=None, gap=1, eta=0.0, t_enc=None, step_noise=None, trace=None):
    """diffusion.py:848-901.  top_p / clamp_step / clamp_first are not forwarded by the
    reference's ddim_sample_loop (diffusion.py:825-839), so rounding runs on every step."""
    x = noise
    for k, i in enumerate(ddim_loop_indices(d.num_timesteps, gap, t_enc)):
        t = torch.tensor([i] * shape[0])
        nz = step_noise(k, i, x) if step_noise is not None else None
        out = ddim_sample(d, model_fn, x, t, clip_denoised, emb_w, eta, mask, x_start, noise=nz)
        if trace is not None:
            trace.append(out)
        x = out["sample"]
    return x


def start_latent_generation(x_start, mask3, noise=None):
    """run/sample.py:190-193: random noise everywhere except the anchored (mask==0) prefix."""
    if noise is None:
        noise = torch.randn_like(x_start)
    return torch.where(torch.eq(mask3, 0), x_start, noise)


def start_latent_modification(d, x_start, mask3, noising_t, noise=None):
    """run/sample.py:195-197: q_sample on [B,L,E,1] with t [B,1] = noising_t-1 and a [B,L,E] mask."""
    B = x_start.shape[0]
    t = torch.full((B, 1), noising_t - 1)
    xs = x_start.unsqueeze(-1)
    if noise is not None:
        noise = noise.unsqueeze(-1)
    return q_sample(d, xs, t, noise=noise, mask=mask3).squeeze(-1)
