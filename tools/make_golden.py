#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python (imported from
/root/reference, read-only) on small seeded inputs.  Runs only in the build container:
the reference never travels to the GPU box; the fixtures (data only) do.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py

The only patch applied is to `transformers.AutoConfig.from_pretrained` (needs the HF hub for
'bert-base-uncased', MuseDiffusion/models/network.py:44): it returns a local BertConfig whose
defaults equal bert-base-uncased, narrowed (hidden_size / layers / heads / ffn) to reach
BASELINE's small shapes.  Nothing in /root/reference is modified.  Weights are the seeded
synthetic state_dict of oracle.denoiser.random_state_dict loaded through load_state_dict
(the reference's key names), inputs come from oracle.fixtures.
"""
import hashlib
import os
import sys
from functools import partial
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MUSE_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import transformers  # noqa: E402
from transformers import BertConfig  # noqa: E402

_BERT = {}
transformers.AutoConfig.from_pretrained = staticmethod(lambda name, **kw: BertConfig(**_BERT))

from MuseDiffusion.models import diffusion as rdiff  # noqa: E402
from MuseDiffusion.models import rounding as rround  # noqa: E402
from MuseDiffusion.models import step_sample as rstep  # noqa: E402
from MuseDiffusion.models.network import TransformerNetModel  # noqa: E402
from MuseDiffusion.utils.initialization import create_model_and_diffusion  # noqa: E402

from oracle import fixtures as fx  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def npy(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def build(cfg, dropout=0.0, predict_xstart=True):
    """create_model_and_diffusion through the reference factory (utils/initialization.py:108-136)."""
    _BERT.clear()
    _BERT.update(hidden_size=cfg["H"], num_hidden_layers=cfg["nL"], num_attention_heads=cfg["nh"],
                 intermediate_size=cfg["F"], hidden_dropout_prob=dropout,
                 attention_probs_dropout_prob=dropout)
    args = SimpleNamespace(hidden_dim=cfg["E"], hidden_t_dim=cfg["Tt"], vocab_size=cfg["V"],
                           seq_len=cfg["L"], dropout=dropout, noise_schedule="sqrt",
                           diffusion_steps=2000, timestep_respacing="",
                           rescale_timesteps=True, predict_xstart=predict_xstart)
    return create_model_and_diffusion(args)


def sd_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(npy(sd[k])).tobytes())
    return h.hexdigest()


def gen_schedules():
    out = {}
    idx = np.array([0, 1, 2, 10, 500, 1000, 1998, 1999])
    d = rdiff.GaussianDiffusion(betas=rdiff.get_named_beta_schedule("sqrt", 2000), predict_xstart=True)
    out["sqrt2000_idx"] = idx
    for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next",
                 "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
                 "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                 "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
        out["sqrt2000_" + name] = getattr(d, name)[idx]
    for sched in ("linear", "cosine", "sqrt", "trunc_cos", "trunc_lin", "pw_lin"):
        b = rdiff.get_named_beta_schedule(sched, 50)
        out["betas50_" + sched] = b
        g = rdiff.GaussianDiffusion(betas=b, predict_xstart=True)
        out["pmc1_50_" + sched] = g.posterior_mean_coef1
        out["plvc_50_" + sched] = g.posterior_log_variance_clipped
    for key, (T, spec) in {"ddim50": (2000, "ddim50"), "sec": (300, "10,15,20"), "full": (2000, [2000]),
                           "odd": (100, [7, 3])}.items():
        use = rdiff.space_timesteps(T, spec)
        out["space_" + key] = np.array(sorted(use))
        sp = rdiff.SpacedDiffusion(use_timesteps=use, betas=rdiff.get_named_beta_schedule("sqrt", T),
                                   rescale_timesteps=True, predict_xstart=True)
        out["spaced_betas_" + key] = sp.betas
        out["spaced_map_" + key] = np.array(sp.timestep_map)
    t = torch.tensor([0.0, 0.5, 3.0, 499.5, 999.5])
    out["temb_t"] = npy(t)
    out["temb_128"] = npy(TransformerNetModel.timestep_embedding(t, 128))
    out["temb_33"] = npy(TransformerNetModel.timestep_embedding(t, 33))
    fake = SimpleNamespace(num_timesteps=6)
    rs = rstep.LossSecondMomentResampler(fake, history_per_term=3)
    rng = np.random.RandomState(3)
    ts = rng.randint(0, 6, size=64).tolist() + list(range(6)) * 3
    ls = rng.rand(len(ts)).tolist()
    rs.update_with_all_losses(ts[:10], ls[:10])
    out["lsr_w_cold"] = rs.weights().copy()
    rs.update_with_all_losses(ts[10:], ls[10:])
    out["lsr_ts"], out["lsr_ls"] = np.array(ts), np.array(ls)
    out["lsr_w_warm"] = rs.weights().copy()
    np.random.seed(11)
    idx_t, w_t = rs.sample(16, "cpu")
    out["lsr_sample_idx"], out["lsr_sample_w"] = npy(idx_t), npy(w_t)
    np.random.seed(11)
    idx_u, w_u = rstep.UniformSampler(fake).sample(16, "cpu")
    out["uni_sample_idx"], out["uni_sample_w"] = npy(idx_u), npy(w_u)
    out["fix_w"] = rstep.FixSampler(SimpleNamespace(num_timesteps=10)).weights()
    np.savez_compressed(os.path.join(OUT, "schedules.npz"), **out)
    print("schedules.npz", len(out), "arrays")


def gen_model_case(tag, compact):
    cfg = fx.CONFIGS[tag]
    model, diffusion = build(cfg)
    sd = fx.state_dict(tag)
    model.load_state_dict(sd)
    model.eval().requires_grad_(False)
    out = {"sd_sha256": np.array(sd_digest(sd))}

    def keep(name, t, is_input=False):
        t = npy(t)
        if compact:
            if is_input:
                return  # rebuilt from oracle.fixtures seeds by the tests
            if t.dtype.kind == "f" and t.ndim == 3:
                t = t[:, ::8]
        out[name] = t

    if not compact:
        for k, v in model.state_dict().items():
            out["sd." + k] = npy(v)
    B, L, E, V = cfg["B"], cfg["L"], cfg["E"], cfg["V"]
    inp = fx.case_inputs(tag, model.word_embedding.weight)
    batch, x_start, mask3 = inp["batch"], inp["x_start"], inp["mask3"]
    for k in ("input_ids", "input_mask", "correct_ids"):
        keep(k, batch[k], is_input=True)
    # ---- forward with per-layer hiddens (network.py:131-158)
    hiddens = []
    hooks = [lyr.register_forward_hook(lambda m, i, o: hiddens.append(o[0] if isinstance(o, tuple) else o))
             for lyr in model.input_transformers.layer]
    y = model(inp["fwd_x"], inp["fwd_t"])
    for h in hooks:
        h.remove()
    keep("fwd_x", inp["fwd_x"], True)
    keep("fwd_t", inp["fwd_t"], True)
    keep("fwd_y", y)
    keep("fwd_emb_t", model.time_embed(model.timestep_embedding(inp["fwd_t"], model.hidden_t_dim)))
    for i, h in enumerate(hiddens):
        if i in fx.HIDDEN_KEEP.get(tag, range(len(hiddens))):
            keep("fwd_hidden%d" % i, h)
    # ---- logits / rounding (network.py:91-93, rounding.py:21-47)
    keep("logits", model.get_logits(y))
    keep("logits_argmax", torch.argmax(model.get_logits(y), dim=-1))
    model_emb = torch.nn.Embedding(V, E, _weight=model.word_embedding.weight.clone()).eval().requires_grad_(False)
    keep("round_in", inp["round_in"], True)
    keep("round_out", rround.denoised_fn_round(model_emb, inp["round_in"], None))
    keep("round_idx", rround.get_efficient_knn(model_emb.weight, inp["round_in"].reshape(-1, E))[1][0])
    # ---- start latents (run/sample.py:185-197)
    x_gen = torch.where(torch.eq(mask3, 0), x_start, inp["gen_noise0"])
    keep("gen_start", x_gen)
    tt = torch.full((B, 1), fx.NOISING_T - 1)
    x_mod = diffusion.q_sample(x_start.unsqueeze(-1), tt, noise=inp["mod_noise"].unsqueeze(-1),
                               mask=mask3).squeeze(-1)
    keep("mod_start", x_mod)
    keep("q_out", diffusion.q_sample(x_start, inp["q_t"], noise=inp["q_noise"], mask=batch["input_mask"]))
    # ---- single reverse steps; draws come from the seeded global generator (diffusion.py:349-404, :701-757)
    fn = partial(rround.denoised_fn_round, model_emb, dist=None)
    for name, tval in (("hi", 1999), ("mid", 700), ("zero", 0)):
        tvec = torch.tensor([tval] * B)
        torch.manual_seed(fx.step_seed(tag))
        r = diffusion.p_sample(model, x_gen, tvec, clip_denoised=True, denoised_fn=fn, model_kwargs={},
                               top_p=1, mask=mask3, x_start=x_start)
        keep("ps_%s_sample" % name, r["sample"])
        keep("ps_%s_x0" % name, r["pred_xstart"])
        keep("ps_%s_mean" % name, r["greedy_mean"])
        torch.manual_seed(fx.step_seed(tag))
        r = diffusion.ddim_sample(model, x_gen, tvec, clip_denoised=True, denoised_fn=fn, model_kwargs={},
                                  mask=mask3, x_start=x_start)
        keep("dd_%s_sample" % name, r["sample"])
    torch.manual_seed(fx.free_seed(tag))
    r = diffusion.p_sample(model, x_gen, inp["free_t"], clip_denoised=False, denoised_fn=None, model_kwargs={},
                           top_p=None)
    keep("ps_free_sample", r["sample"])
    torch.manual_seed(fx.free_seed(tag))
    r = diffusion.ddim_sample(model, x_gen, inp["free_t"], clip_denoised=False, denoised_fn=None,
                              model_kwargs={}, eta=0.5)
    keep("dd_free_sample", r["sample"])
    # ---- loops (diffusion.py:406-540, :797-901) + final tokens (run/sample.py:218-220)
    common = dict(model=model, shape=(B, L, E), clip_denoised=True, denoised_fn=fn, model_kwargs={},
                  top_p=1, clamp_step=0, clamp_first=True, mask=mask3, x_start=x_start, only_last=True)
    torch.manual_seed(fx.loop_seed(tag, "ddim50"))
    s = diffusion.ddim_sample_loop(noise=x_gen, gap=40, t_enc=None, **common)[-1]
    keep("loop_ddim50", s)
    keep("loop_ddim50_tokens", torch.argmax(model.get_logits(s), dim=-1))
    torch.manual_seed(fx.loop_seed(tag, "p12"))
    s = diffusion.p_sample_loop(noise=x_gen, gap=1, t_enc=12, **common)[-1]
    keep("loop_p12", s)
    keep("loop_p12_tokens", torch.argmax(model.get_logits(s), dim=-1))
    torch.manual_seed(fx.loop_seed(tag, "mod"))
    s = diffusion.ddim_sample_loop(noise=x_mod, gap=10, t_enc=fx.NOISING_T, **common)[-1]
    keep("loop_mod", s)
    keep("loop_mod_tokens", torch.argmax(model.get_logits(s), dim=-1))
    np.savez_compressed(os.path.join(OUT, "model_%s.npz" % tag), **out)
    print("model_%s.npz" % tag, len(out), "arrays")


def gen_losses(tag, slim=False, eps=False):
    """training_losses both variants (diffusion.py:594-699), dropout 0, grads of four parameters (slim: + two more, large
    matrices kept as every 4th row and column - oracle.fixtures.slim).  eps: predict_xstart=False (the `_x0_helper` branch of
    diffusion.py:586-590: the model output is read as the noise) -> losses_<tag>_eps.npz."""
    keepg = (lambda t: npy(fx.slim(t))) if slim else npy
    cfg = fx.CONFIGS[tag]
    model, diffusion = build(cfg, dropout=0.0, predict_xstart=not eps)
    sd = fx.state_dict(tag)
    model.load_state_dict(sd)
    model.train().requires_grad_(True)
    li = fx.loss_inputs(tag)
    batch, t, w = li["batch"], li["t"], li["w"]
    out = {"sd_sha256": np.array(sd_digest(sd)), "t": npy(t), "loss_w": npy(w)}
    for variant in ("plain", "corrupt"):
        kw = {k: v for k, v in batch.items() if variant == "corrupt" or k != "correct_ids"}
        model.zero_grad(set_to_none=True)
        torch.manual_seed(fx.loss_seed(tag))
        terms = diffusion.training_losses(model, t, model_kwargs=kw)
        (terms["loss"] * w).mean().backward()
        for k in ("mse", "nll", "loss"):
            out["%s_%s" % (variant, k)] = npy(terms[k])
        out["%s_g_word" % variant] = keepg(model.word_embedding.weight.grad)
        out["%s_g_q0" % variant] = keepg(model.input_transformers.layer[0].attention.self.query.weight.grad)
        out["%s_g_te0" % variant] = keepg(model.time_embed[0].weight.grad)
        out["%s_g_lmb" % variant] = keepg(model.lm_head.bias.grad)
        if slim:
            out["%s_g_v1" % variant] = keepg(model.input_transformers.layer[1].attention.self.value.weight.grad)
            out["%s_g_ff2" % variant] = keepg(model.input_transformers.layer[0].output.dense.weight.grad)
    name = "losses_%s%s.npz" % (tag, "_eps" if eps else "")
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, len(out), "arrays")


class InjectedDropout:
    """Train-mode dropout of the reference with KNOWN masks: `torch.nn.functional.dropout` (what nn.Dropout.forward and HF's
    eager attention both call) is replaced, for the duration of one forward, by `x * mask_k / (1 - p)` with the k-th mask of
    oracle.fixtures.dropout_masks in call order (network.py:149, then per BertLayer: probabilities, attention output, FFN
    output).  Nothing in the reference or in transformers is modified on disk; the patch is removed on exit."""

    def __init__(self, tag, p):
        self.order = [name for name, _ in fx.dropout_sites(tag)]
        self.masks, self.p, self.calls = fx.dropout_masks(tag, p), p, 0

    def __enter__(self):
        self.orig = torch.nn.functional.dropout

        def fake(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            assert abs(p - self.p) < 1e-12, p
            m = self.masks[self.order[self.calls]]
            self.calls += 1
            assert m.shape == x.shape, (self.order[self.calls - 1], m.shape, x.shape)
            return x * m.to(x.dtype) / (1.0 - p)
        torch.nn.functional.dropout = fake
        return self

    def __exit__(self, *a):
        torch.nn.functional.dropout = self.orig


def gen_losses_dropout(tag, slim=False):
    """training_losses (both variants) in TRAIN mode with dropout 0.1 at the reference's three kinds of site, masks injected."""
    keepg = (lambda t: npy(fx.slim(t))) if slim else npy
    cfg = fx.CONFIGS[tag]
    p = fx.DROPOUT_P
    model, diffusion = build(cfg, dropout=p)
    sd = fx.state_dict(tag)
    model.load_state_dict(sd)
    model.train().requires_grad_(True)
    li = fx.loss_inputs(tag)
    batch, t, w = li["batch"], li["t"], li["w"]
    out = {"sd_sha256": np.array(sd_digest(sd)), "t": npy(t), "loss_w": npy(w), "p": np.array(p)}
    for variant in ("plain", "corrupt"):
        kw = {k: v for k, v in batch.items() if variant == "corrupt" or k != "correct_ids"}
        model.zero_grad(set_to_none=True)
        torch.manual_seed(fx.loss_seed(tag))
        with InjectedDropout(tag, p) as inj:
            terms = diffusion.training_losses(model, t, model_kwargs=kw)
        assert inj.calls == len(inj.order), (inj.calls, len(inj.order))
        (terms["loss"] * w).mean().backward()
        for k in ("mse", "nll", "loss"):
            out["%s_%s" % (variant, k)] = npy(terms[k])
        out["%s_g_word" % variant] = keepg(model.word_embedding.weight.grad)
        out["%s_g_q0" % variant] = keepg(model.input_transformers.layer[0].attention.self.query.weight.grad)
        out["%s_g_v1" % variant] = keepg(model.input_transformers.layer[1].attention.self.value.weight.grad)
        out["%s_g_ff2" % variant] = keepg(model.input_transformers.layer[0].output.dense.weight.grad)
        out["%s_g_te0" % variant] = keepg(model.time_embed[0].weight.grad)
        out["%s_g_lmb" % variant] = keepg(model.lm_head.bias.grad)
    # the forward alone, train mode, same masks
    inp = fx.case_inputs(tag, model.word_embedding.weight.detach())
    with torch.no_grad(), InjectedDropout(tag, p):
        y = model(inp["fwd_x"], inp["fwd_t"])
        out["fwd_y_train"] = npy(y[:, ::8] if slim else y)
    np.savez_compressed(os.path.join(OUT, "losses_%s_dropout.npz" % tag), **out)
    print("losses_%s_dropout.npz" % tag, len(out), "arrays")


def gen_overload_freeze(tag="tiny"):
    """The reference's pretrained-embedding start (run/train.py:93-95 -> utils/initialization.py:54-68, config/train.py:67-68
    `freeze_embedding`) followed by one optimizer step of its TrainLoop (train_util.py:246-254: torch.optim.AdamW.step, then
    update_ema over every master parameter), on the reference's own model / diffusion classes.
    `overload_embedding` itself cannot be imported here (its first statement imports utils/dist_util.py, which needs `blobfile`),
    so its three effective statements are applied to the reference's model object verbatim in effect:
    `model.word_embedding.weight = Parameter(emb_weight)`; `model.word_embedding.requires_grad_(False)`.
    Recorded: losses, gradients (lm_head.weight is a separate trainable Parameter from here on; the new embedding has none),
    parameters and EMA copies after the step, AdamW's per-parameter state keys."""
    cfg = fx.CONFIGS[tag]
    model, diffusion = build(cfg, dropout=0.0)
    sd = fx.state_dict(tag)
    model.load_state_dict(sd)
    emb = fx.seeded_randn(4242, cfg["V"], cfg["E"]) * fx.EMB_STD
    with torch.no_grad():
        model.word_embedding.weight = torch.nn.Parameter(emb.clone())
    model.word_embedding.requires_grad_(False)
    assert model.lm_head.weight is not model.word_embedding.weight and model.lm_head.weight.requires_grad
    model.train()
    names = [n for n, _ in model.named_parameters()]
    params = list(model.parameters())
    lr, wd, rate = 1e-3, 0.01, 0.9
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=wd)
    ema = [p.detach().clone() for p in params]
    li = fx.loss_inputs(tag)
    batch, t, w = li["batch"], li["t"], li["w"]
    out = {"sd_sha256": np.array(sd_digest(sd)), "emb": npy(emb), "t": npy(t), "loss_w": npy(w), "lr": np.array(lr), "wd": np.array(wd),
           "ema_rate": np.array(rate), "param_names": np.array(names)}
    torch.manual_seed(fx.loss_seed(tag))
    terms = diffusion.training_losses(model, t, model_kwargs=dict(batch))
    (terms["loss"] * w).mean().backward()
    for k in ("mse", "nll", "loss"):
        out[k] = npy(terms[k])
    assert model.word_embedding.weight.grad is None
    watch = {"lmw": "lm_head.weight", "lmb": "lm_head.bias", "q0": "input_transformers.layer.0.attention.self.query.weight",
             "te0": "time_embed.0.weight", "ff2": "input_transformers.layer.0.output.dense.weight", "pos": "position_embeddings.weight"}
    byname = dict(model.named_parameters())
    for k, n in watch.items():
        out["g_" + k] = npy(byname[n].grad)
    opt.step()
    for trg, src in zip(ema, params):                       # update_ema, train_util.py:21-31
        trg.detach().mul_(rate).add_(src, alpha=1 - rate)
    for k, n in watch.items():
        out["p_" + k] = npy(byname[n])
        out["ema_" + k] = npy(ema[names.index(n)])
    out["p_word"] = npy(model.word_embedding.weight)
    out["ema_word"] = npy(ema[names.index("word_embedding.weight")])
    out["opt_state_keys"] = np.array(sorted(opt.state_dict()["state"].keys()))
    out["word_index"] = np.array(names.index("word_embedding.weight"))
    np.savez_compressed(os.path.join(OUT, "overload_freeze_%s.npz" % tag), **out)
    print("overload_freeze_%s.npz" % tag, len(out), "arrays; params without optimizer state:",
          sorted(set(range(len(params))) - set(opt.state_dict()["state"].keys())))


def gen_reference_checkpoint():
    """A checkpoint directory as the REFERENCE writes it (utils/train_util.py:294-319: `th.save(model.state_dict())`,
    `th.save(opt.state_dict())`, one `ema_{rate}_{step:06d}.pt` per rate) from the reference's own model class and
    torch.optim.AdamW, at a micro shape so that the files stay a few tens of KB; plus the reference's forward on the saved weights."""
    cfg = dict(H=64, nL=1, nh=2, F=64, E=16, Tt=16, L=8, B=2, V=60)
    torch.manual_seed(4321)
    model, diffusion = build(cfg)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.0)
    g = torch.Generator().manual_seed(77)
    for p_ in model.parameters():
        p_.grad = torch.randn(p_.shape, generator=g) * 0.01
    opt.step()
    ema = {k: v.detach().clone() * 0.999 for k, v in model.state_dict().items()}     # stands for an EMA copy: any tensors under the model's keys
    d = os.path.join(OUT, "ref_ckpt")
    os.makedirs(d, exist_ok=True)
    step = 7
    torch.save(model.state_dict(), os.path.join(d, "model_%06d.pt" % step))
    torch.save(ema, os.path.join(d, "ema_0.9999_%06d.pt" % step))
    torch.save(opt.state_dict(), os.path.join(d, "opt_%06d.pt" % step))
    model.eval()
    x = fx.seeded_randn(31, cfg["B"], cfg["L"], cfg["E"])
    t = torch.tensor([12.5, 700.0])
    with torch.no_grad():
        y = model(x, t)
    np.savez_compressed(os.path.join(d, "forward.npz"), x=npy(x), t=npy(t), y=npy(y), **{"cfg_" + k: np.array(v) for k, v in cfg.items()})
    print("ref_ckpt:", sorted(os.listdir(d)))


def gen_logits_mode2(tag="tiny"):
    """get_logits with logits_mode 2 (network.py:94-104; the reference implements it and never constructs it): the reference's own method on
    a model whose logits_mode attribute is set to 2, over seeded hidden rows incl. exact embedding rows (distance 0: the clamp / sqrt edge)."""
    cfg = fx.CONFIGS[tag]
    model, _ = build(cfg)
    sd = fx.state_dict(tag)
    model.load_state_dict(sd)
    model.eval()
    model.logits_mode = 2
    g = torch.Generator().manual_seed(41)
    hidden = torch.randn(2, 9, cfg["E"], generator=g)
    hidden[0, :3] = sd["lm_head.weight"][[5, 0, cfg["V"] - 1]]           # rows of the table itself
    with torch.no_grad():
        out = model.get_logits(hidden)
    np.savez_compressed(os.path.join(OUT, "logits_mode2.npz"), tag=np.array(tag), hidden=npy(hidden), scores=npy(out), sd_digest=np.array(sd_digest(sd)))
    print("logits_mode2.npz", out.shape)


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:]
    if not only or "logits2" in only:
        gen_logits_mode2()
    if not only or "ckpt" in only:
        gen_reference_checkpoint()
    if not only or "base" in only:
        gen_schedules()
        gen_model_case("tiny", compact=False)
        gen_model_case("same", compact=False)
        gen_model_case("c1", compact=True)
        gen_losses("tiny")
    if not only or "overload" in only:
        gen_overload_freeze("tiny")
    if not only or "dropout" in only:
        gen_losses_dropout("tiny")
    if not only or "bench" in only:
        gen_model_case("c2s", compact=True)      # BASELINE config 2's width and seq_len (2 layers, 2 sequences)
        gen_losses("c5s", slim=True)             # config 5's seq_len 1024, same width
        gen_losses_dropout("c5s", slim=True)
    if not only or "eps" in only:
        gen_losses("tiny", eps=True)             # training_losses with predict_xstart=False (diffusion.py:577-592)
    if not only or "deep" in only:
        gen_model_case("c2d", compact=True)      # config 2's TRUE depth: 12 layers, seq_len 512, 2 sequences
        gen_model_case("bbd", compact=True)      # bert-base as the reference instantiates it: 12 layers of H 768
    if not only or "deeptrain" in only:
        gen_losses("c5d", slim=True)             # config 5's TRUE depth: training_losses at 12 layers, seq_len 1024
        gen_losses_dropout("c5d", slim=True)
    if not only or "bertbase" in only:
        gen_model_case("bb", compact=True)       # the reference-true width (H 768, 12 heads of 64, ffn 3072)
        gen_model_case("bb500", compact=True)    # ... with the released weights' E = 500
