"""Oracle: the TransformerNetModel denoiser as plain torch-CPU fp32 ops over a state_dict.

TEST INFRASTRUCTURE (see oracle/__init__.py).  No `transformers` import: the
BertEncoder (third-party, pinned 4.22.2, reference call sites
MuseDiffusion/models/network.py:74 and :151) is restated from its published
algorithm: per layer
    a   = softmax(Q K^T / sqrt(dh)) V            (no mask is ever passed, network.py:151)
    x1  = LN(a Wo^T + bo + x)                    (BertSelfOutput, eps 1e-12)
    x2  = LN(gelu_erf(x1 Wi^T + bi) Wd^T + bd + x1)   (BertIntermediate + BertOutput)
Dropouts are inference no-ops; train mode is modelled with EXPLICIT keep masks (`masks`: site name -> bool tensor, `p`: the
rate whose 1 / (1 - p) rescales the kept units) at the reference's sites: "emb" after the embedding LayerNorm (network.py:149),
per layer "l{i}.attn" on the attention probabilities, "l{i}.ao" / "l{i}.ffn" after the attention-output / FFN-output dense
(HF BertSelfAttention / BertSelfOutput / BertOutput).
"""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(timesteps, dim, max_period=10000):
    """network.py:108-129: [cos(t f) | sin(t f)], f = exp(-ln(max_period) * arange(half)/half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def count_layers(sd):
    n = 0
    while "input_transformers.layer.%d.attention.self.query.weight" % n in sd:
        n += 1
    return n


def _lin(x, sd, prefix):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def _drop(x, masks, name, p):
    if not masks or name not in masks:
        return x
    return x * masks[name].to(x.dtype) / (1.0 - p)


def encoder_layer(x, sd, i, num_heads, eps=1e-12, collect=None, masks=None, p=0.0):
    """One post-LN BERT layer (HF BertLayer; reached from network.py:151)."""
    pre = "input_transformers.layer.%d." % i
    B, L, H = x.shape
    dh = H // num_heads

    def heads(t):
        return t.view(B, L, num_heads, dh).permute(0, 2, 1, 3)

    q = heads(_lin(x, sd, pre + "attention.self.query"))
    k = heads(_lin(x, sd, pre + "attention.self.key"))
    v = heads(_lin(x, sd, pre + "attention.self.value"))
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
    probs = _drop(torch.softmax(scores, dim=-1), masks, "l%d.attn" % i, p)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(B, L, H)
    if collect is not None:
        collect.setdefault("ctx", []).append(ctx)
    x1 = F.layer_norm(_drop(_lin(ctx, sd, pre + "attention.output.dense"), masks, "l%d.ao" % i, p) + x, (H,),
                      sd[pre + "attention.output.LayerNorm.weight"],
                      sd[pre + "attention.output.LayerNorm.bias"], eps)
    inter = F.gelu(_lin(x1, sd, pre + "intermediate.dense"))  # exact erf form (hidden_act="gelu")
    x2 = F.layer_norm(_drop(_lin(inter, sd, pre + "output.dense"), masks, "l%d.ffn" % i, p) + x1, (H,),
                      sd[pre + "output.LayerNorm.weight"], sd[pre + "output.LayerNorm.bias"], eps)
    return x2


def embed_inputs(sd, x, timesteps, hidden_t_dim):
    """network.py:139-149: time MLP, optional up-projection, pos + x + t, LayerNorm."""
    emb_t = timestep_embedding(timesteps, hidden_t_dim)
    emb_t = _lin(F.silu(_lin(emb_t, sd, "time_embed.0")), sd, "time_embed.2")
    if "input_up_proj.0.weight" in sd:
        emb_x = _lin(torch.tanh(_lin(x, sd, "input_up_proj.0")), sd, "input_up_proj.2")
    else:
        emb_x = x
    L = x.shape[1]
    pos = sd["position_embeddings.weight"][:L]
    h = pos[None] + emb_x + emb_t[:, None, :]
    H = h.shape[-1]
    return F.layer_norm(h, (H,), sd["LayerNorm.weight"], sd["LayerNorm.bias"], 1e-12), emb_t


def forward(sd, x, timesteps, num_heads, hidden_t_dim=None, collect=None, masks=None, p=0.0):
    """TransformerNetModel.forward (network.py:131-158).  x [B,L,E] fp32, timesteps [B]."""
    if hidden_t_dim is None:
        hidden_t_dim = sd["time_embed.0.weight"].shape[1]
    h, emb_t = embed_inputs(sd, x, timesteps, hidden_t_dim)
    h = _drop(h, masks, "emb", p)
    if collect is not None:
        collect["emb_t"] = emb_t
        collect["emb_inputs"] = h
        collect["hidden"] = []
    for i in range(count_layers(sd)):
        h = encoder_layer(h, sd, i, num_heads, collect=collect, masks=masks, p=p)
        if collect is not None:
            collect["hidden"].append(h)
    if "output_down_proj.0.weight" in sd:
        h = _lin(torch.tanh(_lin(h, sd, "output_down_proj.0")), sd, "output_down_proj.2")
    return h.type(x.dtype)


def get_embeds(sd, ids):
    """network.py:88-89."""
    return sd["word_embedding.weight"][ids.long()]


def get_logits(sd, hidden, logits_mode=1):
    """network.py:91-106.  logits_mode 1: lm_head(hidden) (lm_head.weight is tied to word_embedding.weight); logits_mode 2 (:94-104): the
    negative Euclidean distance to every lm_head row, -sqrt(clamp(|W_v|^2 + |x_n|^2 - 2 W_v.x_n, 0, inf)), the three terms associated
    as the reference adds them."""
    if logits_mode == 1:
        return F.linear(hidden, sd["lm_head.weight"], sd["lm_head.bias"])
    if logits_mode != 2:
        raise NotImplementedError
    W = sd["lm_head.weight"]
    flat = hidden.reshape(-1, hidden.shape[-1])
    emb_norm = (W ** 2).sum(-1).view(-1, 1)                          # [V, 1]
    arr_norm = (flat ** 2).sum(-1).view(-1, 1)                       # [N, 1]
    dist = emb_norm + arr_norm.transpose(0, 1) - 2.0 * torch.mm(W, flat.transpose(0, 1))      # [V, N]
    scores = torch.sqrt(torch.clamp(dist, 0.0, float("inf")))
    return -scores.transpose(0, 1).reshape(*hidden.shape[:-1], W.shape[0])


def random_state_dict(E, H, F_, num_layers, V, L, Tt, seed=0, emb_std=1.0):
    """A state_dict with the reference's key names/shapes and torch-default-like init.

    Used by tests/bench to build synthetic weights without the reference; values do not
    have to match the reference's init bit for bit, only its layout and scale.
    """
    g = torch.Generator().manual_seed(seed)

    def lin(out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)
        w = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound
        b = (torch.rand(out_f, generator=g) * 2 - 1) * bound
        return w, b

    sd = {}
    sd["position_ids"] = torch.arange(L)[None]
    sd["word_embedding.weight"] = torch.randn(V, E, generator=g) * emb_std
    sd["lm_head.weight"] = sd["word_embedding.weight"]
    sd["lm_head.bias"] = (torch.rand(V, generator=g) * 2 - 1) / math.sqrt(E)
    for name, (o, i) in {"time_embed.0": (4 * Tt, Tt), "time_embed.2": (H, 4 * Tt)}.items():
        sd[name + ".weight"], sd[name + ".bias"] = lin(o, i)
    if E != H:
        sd["input_up_proj.0.weight"], sd["input_up_proj.0.bias"] = lin(H, E)
        sd["input_up_proj.2.weight"], sd["input_up_proj.2.bias"] = lin(H, H)
    for l in range(num_layers):
        p = "input_transformers.layer.%d." % l
        for nm in ("attention.self.query", "attention.self.key", "attention.self.value",
                   "attention.output.dense"):
            sd[p + nm + ".weight"], sd[p + nm + ".bias"] = lin(H, H)
        sd[p + "attention.output.LayerNorm.weight"] = 1 + 0.1 * torch.randn(H, generator=g)
        sd[p + "attention.output.LayerNorm.bias"] = 0.1 * torch.randn(H, generator=g)
        sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"] = lin(F_, H)
        sd[p + "output.dense.weight"], sd[p + "output.dense.bias"] = lin(H, F_)
        sd[p + "output.LayerNorm.weight"] = 1 + 0.1 * torch.randn(H, generator=g)
        sd[p + "output.LayerNorm.bias"] = 0.1 * torch.randn(H, generator=g)
    sd["position_embeddings.weight"] = torch.randn(L, H, generator=g)
    sd["LayerNorm.weight"] = 1 + 0.1 * torch.randn(H, generator=g)
    sd["LayerNorm.bias"] = 0.1 * torch.randn(H, generator=g)
    if E != H:
        sd["output_down_proj.0.weight"], sd["output_down_proj.0.bias"] = lin(H, H)
        sd["output_down_proj.2.weight"], sd["output_down_proj.2.bias"] = lin(E, H)
    return sd
