"""The reference's sampling hot block (run/sample.py:185-220) as two functions: everything between
"a batch of token ids arrives" and "a batch of token ids leaves", sharded over ranks when a process
group is initialised.  MIDI decoding / metrics (run/sample.py:222-294) are out of scope."""
from functools import partial

import torch

from . import sharding
from .models.rounding import denoised_fn_round


def _prepare(model, cond, device):
    ids = cond["input_ids"].to(device)
    mask = cond["input_mask"].to(device)
    x_start = model.get_embeds(ids)                                              # sample.py:185
    mask3 = torch.broadcast_to(mask.unsqueeze(dim=-1), x_start.shape)            # sample.py:186
    return x_start, mask3


def _run(model, diffusion, x_start, mask3, x_noised, step, noising_t, clip_denoised, top_p, clamp_step):
    T = diffusion.num_timesteps
    if step == T:                                                                 # sample.py:109-114
        gap, sample_fn = 1, diffusion.p_sample_loop
    else:
        gap, sample_fn = T // step, diffusion.ddim_sample_loop
    emb = torch.nn.Embedding(model.word_embedding.num_embeddings, model.word_embedding.embedding_dim,
                             _weight=model.word_embedding.weight.detach().clone()).eval().requires_grad_(False)
    samples = sample_fn(model=model, shape=tuple(x_start.shape), noise=x_noised, clip_denoised=clip_denoised,
                        denoised_fn=partial(denoised_fn_round, emb, dist=None), model_kwargs={}, top_p=top_p,
                        clamp_step=clamp_step, clamp_first=True, mask=mask3, x_start=x_start, gap=gap,
                        t_enc=noising_t, only_last=True)                          # sample.py:200-215
    return model.argmax_tokens(samples[-1])                                       # sample.py:218-220


@torch.no_grad()
def generate(model, diffusion, cond, step=None, clip_denoised=True, top_p=1, clamp_step=0, sharded=True, noise=None,
             t_enc=None):
    """Generation mode: noise everywhere except the anchored meta prefix (run/sample.py:190-193).
    `cond` = {'input_ids', 'input_mask'} for the GLOBAL batch; returns int64 tokens [B, L] on every rank.
    Additions to the reference's block: `noise` = the start latent's draw for the GLOBAL batch (each rank takes its rows;
    default: this rank's `torch.randn_like`, as the reference), `t_enc` = stop after that many iterations
    (the loops' own argument, diffusion.py:425)."""
    device = model.word_embedding.weight.device
    B = cond["input_ids"].shape[0]
    local = sharding.shard_batch(cond) if sharded else cond
    x_start, mask3 = _prepare(model, local, device)
    if noise is None:
        noise = torch.randn_like(x_start)
    else:
        lo, hi = sharding.shard_bounds(B) if sharded else (0, B)
        noise = noise[lo:hi].to(device)
    x_noised = torch.where(torch.eq(mask3, 0), x_start, noise)
    tokens = _run(model, diffusion, x_start, mask3, x_noised, step or diffusion.num_timesteps, t_enc, clip_denoised,
                  top_p, clamp_step)
    return sharding.gather_rows(tokens, B) if sharded else tokens


@torch.no_grad()
def modify(model, diffusion, cond, step, strength=0.75, clip_denoised=True, top_p=1, clamp_step=0, sharded=True, noise=None):
    """Modification mode: q_sample the corrupted sequence to noising_t = int(step * strength), then
    run the reverse loop for noising_t iterations (run/sample.py:195-197, :214).  `noise`: q_sample's draw for the
    GLOBAL batch, [B, L, E] (default: this rank's device generator, as the reference)."""
    device = model.word_embedding.weight.device
    B = cond["input_ids"].shape[0]
    local = sharding.shard_batch(cond) if sharded else cond
    x_start, mask3 = _prepare(model, local, device)
    noising_t = int(step * strength)
    timestep = torch.full((x_start.shape[0], 1), noising_t - 1, device=device)
    if noise is not None:
        lo, hi = sharding.shard_bounds(B) if sharded else (0, B)
        noise = noise[lo:hi].to(device).unsqueeze(-1)
    x_noised = diffusion.q_sample(x_start.unsqueeze(-1), timestep, noise=noise, mask=mask3).squeeze(-1)
    tokens = _run(model, diffusion, x_start, mask3, x_noised, step, noising_t, clip_denoised, top_p, clamp_step)
    return sharding.gather_rows(tokens, B) if sharded else tokens
