"""Nearest-embedding rounding ("clamp"), API of MuseDiffusion/models/rounding.py.

`denoised_fn_round(model_emb, text_emb, t, dist=None)` snaps every position of `text_emb` to its
nearest word-embedding row (squared-L2, rounding.py:21-28) with one fused kernel
(`mh_round_to_embedding`, then an embedding gather).  The sampling loops recognise
`functools.partial(denoised_fn_round, model_emb, dist=None)` — the exact object run/sample.py:205
builds — and fuse the rounding into the captured reverse step instead of calling it.
"""
import functools

import torch

from .. import _lib, ops


def get_knn(model_emb, text_emb, dist="cos"):
    """Top-6 neighbours by cosine / L2 (rounding.py:8-18).  Not used by the sampling callers
    (they pass dist=None); kept for API completeness on top of the GEMM kernel."""
    if dist not in ("cos", "l2"):
        raise ValueError("get_knn function got unknown `dist`: expected 'cos' or 'l2', got {!r}".format(dist))
    _lib.require_device(model_emb, text_emb)
    V, E = model_emb.shape
    Ep = ops.pad64(E)
    w = ops.cast_pad(model_emb.detach().float(), Ep, _lib.MH_F32)
    x = ops.cast_pad(text_emb.detach().float().reshape(-1, E), Ep, _lib.MH_F32)
    adjacency = ops.gemm_bias_act(w, x, None, None, None, _lib.MH_F32, out_f32=True, N=x.shape[0], K=Ep)  # [V, n]
    if dist == "l2":
        wn = ops.row_sqnorm(model_emb.detach().float()).view(-1, 1)
        xn = ops.row_sqnorm(text_emb.detach().float().reshape(-1, E)).view(1, -1)
        adjacency = -torch.sqrt(torch.clamp(wn + xn - 2.0 * adjacency, min=0.0))
    topk_out = torch.topk(adjacency, k=6, dim=0)
    return topk_out.values, topk_out.indices


def get_efficient_knn(model_emb, text_emb):
    """(values, indices) of the nearest embedding row per position, shapes [1, n] (rounding.py:21-28)."""
    _lib.require_device(model_emb, text_emb)
    E = model_emb.shape[-1]
    flat = text_emb.reshape(-1, E).float()
    idx = ops.round_to_embedding(flat, model_emb.detach()).long()
    near = model_emb.detach()[idx]
    dist = torch.clamp(((near - flat) ** 2).sum(-1), min=0.0)
    return (-dist).unsqueeze(0), idx.unsqueeze(0)


def denoised_fn_round(model, text_emb, t, dist=None):  # NOQA
    """Replace every latent vector by its nearest embedding row (rounding.py:31-47).
    `model` is an nn.Embedding (uses `.weight`)."""
    model_emb = model.weight
    old_shape, old_device = text_emb.shape, text_emb.device
    flat = text_emb.reshape(-1, text_emb.size(-1)) if text_emb.dim() > 2 else text_emb
    flat = flat.to(model_emb.device)
    if dist is not None:
        _, indices = get_knn(model_emb, flat, dist=dist)
        rounded = indices[0]
    else:
        rounded = ops.round_to_embedding(flat.float(), model_emb.detach())
    return ops.embed_gather(model_emb.detach(), rounded).view(old_shape).to(old_device)


def embedding_of_denoised_fn(fn):
    """The embedding table if `fn` is the standard rounding closure
    `partial(denoised_fn_round, model_emb, dist=None)` (run/sample.py:205), else None."""
    if isinstance(fn, functools.partial) and fn.func is denoised_fn_round and len(fn.args) == 1 \
            and fn.keywords.get("dist", None) is None and set(fn.keywords) <= {"dist"}:
        emb = fn.args[0]
        w = getattr(emb, "weight", None)
        if torch.is_tensor(w) and w.dim() == 2:
            return w
    return None
