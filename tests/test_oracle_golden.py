"""Pin the CPU oracle (oracle/) against fixtures produced by the reference's own Python
(tools/make_golden.py).  CPU only.  Tolerances: the oracle restates the same fp32 torch ops in
the same order, so everything except the BertEncoder boundary is bit-exact; the encoder is
restated from the pinned transformers 4.22.2 algorithm and checked at 2e-5 abs (fixture made
with transformers 5.15.0 eager attention; scale by multiply vs divide)."""
import numpy as np
import pytest
import torch

from oracle import denoiser as odn
from oracle import fixtures as fx
from oracle import losses as olo
from oracle import sampling as osa
from oracle import schedule as osc
from conftest import load_golden

TABLES = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next",
          "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
          "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
          "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")


def T(a):
    return torch.from_numpy(np.asarray(a))


def sub(t, compact):
    t = t.detach()
    return t[:, ::8] if (compact and t.dtype.is_floating_point and t.dim() == 3) else t


def test_schedule_tables_bit_exact():
    g = load_golden("schedules.npz")
    d = osc.tables(osc.named_betas("sqrt", 2000))
    idx = g["sqrt2000_idx"]
    for name in TABLES:
        np.testing.assert_array_equal(getattr(d, name)[idx], g["sqrt2000_" + name], err_msg=name)
    for sched in ("linear", "cosine", "sqrt", "trunc_cos", "trunc_lin", "pw_lin"):
        b = osc.named_betas(sched, 50)
        np.testing.assert_array_equal(b, g["betas50_" + sched])
        t = osc.tables(b)
        np.testing.assert_array_equal(t.posterior_mean_coef1, g["pmc1_50_" + sched])
        np.testing.assert_array_equal(t.posterior_log_variance_clipped, g["plvc_50_" + sched])
    with pytest.raises(NotImplementedError):
        osc.named_betas("nope", 10)


def test_respacing():
    g = load_golden("schedules.npz")
    for key, (Tn, spec) in {"ddim50": (2000, "ddim50"), "sec": (300, "10,15,20"), "full": (2000, [2000]),
                            "odd": (100, [7, 3])}.items():
        use = osc.space_timesteps(Tn, spec)
        np.testing.assert_array_equal(np.array(sorted(use)), g["space_" + key])
        sp = osc.spaced(osc.named_betas("sqrt", Tn), use)
        np.testing.assert_array_equal(sp.betas, g["spaced_betas_" + key])
        np.testing.assert_array_equal(np.array(sp.timestep_map), g["spaced_map_" + key])
    with pytest.raises(ValueError):
        osc.space_timesteps(10, [11])
    with pytest.raises(ValueError):
        osc.space_timesteps(2000, "ddim1999")


def test_timestep_embedding():
    g = load_golden("schedules.npz")
    t = T(g["temb_t"])
    np.testing.assert_array_equal(odn.timestep_embedding(t, 128).numpy(), g["temb_128"])
    np.testing.assert_array_equal(odn.timestep_embedding(t, 33).numpy(), g["temb_33"])


def test_samplers():
    g = load_golden("schedules.npz")
    rs = olo.SecondMomentResampler(6, history_per_term=3)
    ts, ls = g["lsr_ts"].tolist(), g["lsr_ls"].tolist()
    rs.update(ts[:10], ls[:10])
    np.testing.assert_array_equal(rs.weights(), g["lsr_w_cold"])
    rs.update(ts[10:], ls[10:])
    np.testing.assert_array_equal(rs.weights(), g["lsr_w_warm"])
    np.random.seed(11)
    idx, w = olo.sample_timesteps(rs.weights(), 16)
    np.testing.assert_array_equal(idx, g["lsr_sample_idx"])
    np.testing.assert_allclose(w.astype(np.float32), g["lsr_sample_w"], rtol=0, atol=0)
    np.random.seed(11)
    idx, w = olo.sample_timesteps(np.ones(6), 16)
    np.testing.assert_array_equal(idx, g["uni_sample_idx"])


@pytest.mark.parametrize("tag", ["tiny", "same", "c1", "bb", "bb500", "c2s", "c2d", "bbd"])
def test_model_case(tag):
    """bb / bb500: the reference-true encoder shape (H 768, 12 heads of 64, ffn 3072), E = 128 and the released E = 500.
    c2s: BASELINE config 2's width and seq_len 512 (2 layers, 2 sequences) - the benchmarked shape.
    c2d / bbd (round 5): config 2's denoiser and bert-base at their TRUE depth of 12 layers."""
    g = load_golden("model_%s.npz" % tag)
    compact = tag in fx.COMPACT
    cfg = fx.CONFIGS[tag]
    sd = fx.state_dict(tag)
    if not compact:  # drift detector: regenerated weights/inputs equal the stored ones
        for k in sd:
            np.testing.assert_array_equal(sd[k].numpy(), g["sd." + k], err_msg=k)
    inp = fx.case_inputs(tag, sd["word_embedding.weight"])
    if not compact:
        for k in ("fwd_x", "fwd_t", "round_in"):
            np.testing.assert_array_equal(inp[k].numpy(), g[k])
    B, L, E = cfg["B"], cfg["L"], cfg["E"]
    nh = cfg["nh"]
    model_fn = lambda x, ts: odn.forward(sd, x, ts, nh)
    emb_w = sd["word_embedding.weight"]
    d = osc.make_diffusion()
    x_start, mask3, batch = inp["x_start"], inp["mask3"], inp["batch"]

    def close(name, val, atol):
        np.testing.assert_allclose(sub(val, compact).numpy(), g[name], rtol=0, atol=atol, err_msg=name)

    # forward, per-layer
    col = {}
    y = odn.forward(sd, inp["fwd_x"], inp["fwd_t"], nh, collect=col)
    wide = cfg["H"] >= 768            # 768 / 3072-term fp32 dot products in another order than the fixture's MKL build
    close("fwd_emb_t", col["emb_t"], 1e-6)
    deep = cfg["nL"] >= 12            # twelve layers of re-ordered fp32 sums
    tol = (1e-4 if deep else 5e-5) if wide else (5e-5 if deep else 2e-5)
    for i, h in enumerate(col["hidden"]):
        if i in fx.HIDDEN_KEEP.get(tag, range(cfg["nL"])):
            close("fwd_hidden%d" % i, h, tol)
    close("fwd_y", y, tol)
    # logits on the golden y (isolates get_logits)
    close("logits", odn.get_logits(sd, y), 1e-4)
    # rounding: indices exact
    idx = osa.nearest_token(emb_w, inp["round_in"])
    np.testing.assert_array_equal(idx.numpy(), g["round_idx"])
    close("round_out", osa.round_to_embedding(emb_w, inp["round_in"]), 0)
    # start latents, q_sample: bit exact
    x_gen = osa.start_latent_generation(x_start, mask3, inp["gen_noise0"])
    close("gen_start", x_gen, 0)
    x_mod = osa.start_latent_modification(d, x_start, mask3, fx.NOISING_T, noise=inp["mod_noise"])
    close("mod_start", x_mod, 0)
    close("q_out", osa.q_sample(d, x_start, inp["q_t"], noise=inp["q_noise"], mask=batch["input_mask"]), 0)
    # single steps
    for name, tval in (("hi", 1999), ("mid", 700), ("zero", 0)):
        tvec = torch.tensor([tval] * B)
        torch.manual_seed(fx.step_seed(tag))
        r = osa.p_sample(d, model_fn, x_gen, tvec, True, emb_w, top_p=1, mask=mask3, x_start=x_start)
        close("ps_%s_x0" % name, r["pred_xstart"], 0)      # rounded -> rows of W, exact
        close("ps_%s_mean" % name, r["greedy_mean"], 0)
        close("ps_%s_sample" % name, r["sample"], 0)
        torch.manual_seed(fx.step_seed(tag))
        r = osa.ddim_sample(d, model_fn, x_gen, tvec, True, emb_w, mask=mask3, x_start=x_start)
        close("dd_%s_sample" % name, r["sample"], 0)
    torch.manual_seed(fx.free_seed(tag))
    r = osa.p_sample(d, model_fn, x_gen, inp["free_t"], False, None, top_p=None)
    close("ps_free_sample", r["sample"], 1e-4)
    torch.manual_seed(fx.free_seed(tag))
    r = osa.ddim_sample(d, model_fn, x_gen, inp["free_t"], False, None, eta=0.5)
    close("dd_free_sample", r["sample"], 2e-3)   # eps = (.)/sqrt_recipm1 amplifies at small t
    # loops: final latent and tokens exact (rounding snaps every step)
    kw = dict(clip_denoised=True, emb_w=emb_w, mask=mask3, x_start=x_start)
    torch.manual_seed(fx.loop_seed(tag, "ddim50"))
    s = osa.ddim_sample_loop(d, model_fn, (B, L, E), x_gen, gap=40, **kw)
    close("loop_ddim50", s, 0)
    np.testing.assert_array_equal(odn.get_logits(sd, s).argmax(-1).numpy(), g["loop_ddim50_tokens"])
    torch.manual_seed(fx.loop_seed(tag, "p12"))
    s = osa.p_sample_loop(d, model_fn, (B, L, E), x_gen, top_p=1, clamp_step=0, clamp_first=True, t_enc=12, **kw)
    close("loop_p12", s, 0)
    np.testing.assert_array_equal(odn.get_logits(sd, s).argmax(-1).numpy(), g["loop_p12_tokens"])
    torch.manual_seed(fx.loop_seed(tag, "mod"))
    s = osa.ddim_sample_loop(d, model_fn, (B, L, E), x_mod, gap=10, t_enc=fx.NOISING_T, **kw)
    close("loop_mod", s, 0)
    np.testing.assert_array_equal(odn.get_logits(sd, s).argmax(-1).numpy(), g["loop_mod_tokens"])


@pytest.mark.parametrize("tag", ["tiny", "c5s", "tiny_eps", "c5d"])
def test_training_losses(tag):
    """c5s: BASELINE config 5's seq_len 1024 at config 2's width (2 layers, 2 sequences); its fixture keeps every 4th row and
    column of the large gradients (fixtures.slim) and two more of them.  tiny_eps: the reference run with predict_xstart=False
    (`_x0_helper`, diffusion.py:577-592)."""
    g = load_golden("losses_%s.npz" % tag)
    eps = tag.endswith("_eps")
    tag = tag.replace("_eps", "")
    big = tag in ("c5s", "c5d")          # (c5d, round 5: config 5's true depth of 12 layers)
    cfg = fx.CONFIGS[tag]
    li = fx.loss_inputs(tag)
    batch, t, w = li["batch"], li["t"], li["w"]
    d = osc.make_diffusion(predict_xstart=not eps)
    for variant in ("plain", "corrupt"):
        sd = {k: v.clone() for k, v in fx.state_dict(tag).items()}
        names = ("word_embedding.weight", "input_transformers.layer.0.attention.self.query.weight",
                 "time_embed.0.weight", "lm_head.bias")
        keys = ("g_word", "g_q0", "g_te0", "g_lmb")
        if big:
            names += ("input_transformers.layer.1.attention.self.value.weight", "input_transformers.layer.0.output.dense.weight")
            keys += ("g_v1", "g_ff2")
        for n in names:
            sd[n].requires_grad_(True)
        sd["lm_head.weight"] = sd["word_embedding.weight"]  # tied (network.py:56-58)
        torch.manual_seed(fx.loss_seed(tag))
        terms = olo.training_losses(
            d, lambda x, ts: odn.forward(sd, x, ts, cfg["nh"]),
            lambda ids: odn.get_embeds(sd, ids), lambda h: odn.get_logits(sd, h),
            t, batch["input_ids"], batch["input_mask"],
            correct_ids=batch["correct_ids"] if variant == "corrupt" else None)
        for k in ("mse", "nll", "loss"):
            np.testing.assert_allclose(terms[k].detach().numpy(), g["%s_%s" % (variant, k)], rtol=2e-5, atol=2e-5)
        (terms["loss"] * w).mean().backward()
        for n, key in zip(names, keys):
            ref = g["%s_%s" % (variant, key)]
            np.testing.assert_allclose(fx.slim(sd[n].grad).numpy() if big else sd[n].grad.numpy(), ref, rtol=1e-3,
                                       atol=2e-6 if not big else 2e-4 * float(np.abs(ref).max()), err_msg=variant + key)


@pytest.mark.parametrize("tag", ["tiny", "c5s", "c5d"])
def test_training_losses_with_dropout_masks(tag):
    """Train mode: the reference ran with dropout 0.1 at its three kinds of site and the masks of fixtures.dropout_masks
    injected (tools/make_golden.py InjectedDropout); the oracle fed the same masks must give the same losses and gradients."""
    g = load_golden("losses_%s_dropout.npz" % tag)
    big = tag in ("c5s", "c5d")
    cfg = fx.CONFIGS[tag]
    p = float(g["p"])
    assert p == fx.DROPOUT_P
    li = fx.loss_inputs(tag)
    batch, t, w = li["batch"], li["t"], li["w"]
    d = osc.make_diffusion()
    masks = fx.dropout_masks(tag, p)
    for variant in ("plain", "corrupt"):
        sd = {k: v.clone() for k, v in fx.state_dict(tag).items()}
        names = ("word_embedding.weight", "input_transformers.layer.0.attention.self.query.weight",
                 "input_transformers.layer.1.attention.self.value.weight", "input_transformers.layer.0.output.dense.weight",
                 "time_embed.0.weight", "lm_head.bias")
        for n in names:
            sd[n].requires_grad_(True)
        sd["lm_head.weight"] = sd["word_embedding.weight"]
        torch.manual_seed(fx.loss_seed(tag))
        terms = olo.training_losses(
            d, lambda x, ts: odn.forward(sd, x, ts, cfg["nh"], masks=masks, p=p),
            lambda ids: odn.get_embeds(sd, ids), lambda h: odn.get_logits(sd, h),
            t, batch["input_ids"], batch["input_mask"],
            correct_ids=batch["correct_ids"] if variant == "corrupt" else None)
        for k in ("mse", "nll", "loss"):
            np.testing.assert_allclose(terms[k].detach().numpy(), g["%s_%s" % (variant, k)], rtol=2e-5, atol=2e-5)
        (terms["loss"] * w).mean().backward()
        for n, key in zip(names, ("g_word", "g_q0", "g_v1", "g_ff2", "g_te0", "g_lmb")):
            ref = g["%s_%s" % (variant, key)]
            np.testing.assert_allclose(fx.slim(sd[n].grad).numpy() if big else sd[n].grad.numpy(), ref, rtol=1e-3,
                                       atol=2e-6 if not big else 2e-4 * float(np.abs(ref).max()), err_msg=variant + key)
    sd = fx.state_dict(tag)
    inp = fx.case_inputs(tag, sd["word_embedding.weight"])
    y = odn.forward(sd, inp["fwd_x"], inp["fwd_t"], cfg["nh"], masks=masks, p=p)
    np.testing.assert_allclose((y[:, ::8] if big else y).numpy(), g["fwd_y_train"], rtol=0, atol=2e-5)
    # and the masks matter: eval-mode output differs
    assert float((odn.forward(sd, inp["fwd_x"], inp["fwd_t"], cfg["nh"]) - y).abs().max()) > 1e-2


def test_overload_embedding_with_frozen_embedding_and_one_optimizer_step():
    """The reference's pretrained-embedding start (utils/initialization.py:54-68 with freeze_embedding, then train_util.py:246-254):
    the oracle's losses and gradients with the head UNTIED from the replaced embedding, then torch.optim.AdamW's rule restated in
    numpy-style torch ops for the parameters that have a gradient, update_ema for all of them."""
    g = load_golden("overload_freeze_tiny.npz")
    tag = "tiny"
    cfg = fx.CONFIGS[tag]
    li = fx.loss_inputs(tag)
    batch, t, w = li["batch"], li["t"], li["w"]
    d = osc.make_diffusion()
    sd = {k: v.clone() for k, v in fx.state_dict(tag).items()}
    sd["lm_head.weight"] = sd["word_embedding.weight"].clone()          # the old embedding tensor stays the head's weight
    sd["word_embedding.weight"] = torch.from_numpy(g["emb"]).clone()    # the pretrained table; frozen
    watch = {"lmw": "lm_head.weight", "lmb": "lm_head.bias", "q0": "input_transformers.layer.0.attention.self.query.weight",
             "te0": "time_embed.0.weight", "ff2": "input_transformers.layer.0.output.dense.weight", "pos": "position_embeddings.weight"}
    for n in watch.values():
        sd[n].requires_grad_(True)
    torch.manual_seed(fx.loss_seed(tag))
    terms = olo.training_losses(d, lambda x, ts: odn.forward(sd, x, ts, cfg["nh"]), lambda ids: odn.get_embeds(sd, ids),
                                lambda h: odn.get_logits(sd, h), t, batch["input_ids"], batch["input_mask"], correct_ids=batch["correct_ids"])
    for k in ("mse", "nll", "loss"):
        np.testing.assert_allclose(terms[k].detach().numpy(), g[k], rtol=2e-5, atol=2e-5)
    (terms["loss"] * w).mean().backward()
    lr, wd, rate = float(g["lr"]), float(g["wd"]), float(g["ema_rate"])
    for k, n in watch.items():
        np.testing.assert_allclose(sd[n].grad.numpy(), g["g_" + k], rtol=1e-3, atol=2e-6, err_msg=k)
        # torch.optim.AdamW, first step (zero moments): p (1 - lr wd) - lr / (1 - b1) * m / (sqrt(v) / sqrt(1 - b2) + eps)
        p0, gr = sd[n].detach(), torch.from_numpy(g["g_" + k])
        m, v = 0.1 * gr, 0.001 * gr * gr
        p1 = p0 * (1 - lr * wd) - (lr / (1 - 0.9)) * (m / (v.sqrt() / (1 - 0.999) ** 0.5 + 1e-8))
        np.testing.assert_allclose(p1.numpy(), g["p_" + k], rtol=2e-6, atol=2e-7, err_msg=k)
        np.testing.assert_allclose((p0 * rate + p1 * (1 - rate)).numpy(), g["ema_" + k], rtol=2e-6, atol=2e-7, err_msg=k)
    # the frozen embedding: no optimizer state, value untouched, EMA copy = update_ema of an unchanged parameter
    assert int(g["word_index"]) not in set(g["opt_state_keys"].tolist())
    np.testing.assert_array_equal(g["p_word"], g["emb"])
    e = torch.from_numpy(g["emb"])
    np.testing.assert_array_equal(g["ema_word"], e.clone().mul_(rate).add_(e, alpha=1 - rate).numpy())


def test_get_logits_mode2_matches_reference():
    """get_logits with logits_mode 2 (network.py:94-104): the oracle against the reference's own method (tools/make_golden.py logits2),
    rows of the table itself included (distance exactly 0 after the clamp)."""
    g = load_golden("logits_mode2.npz")
    sd = fx.state_dict(str(g["tag"]))
    got = odn.get_logits(sd, T(g["hidden"]), logits_mode=2)
    assert torch.equal(got, T(g["scores"]))
    with pytest.raises(NotImplementedError):
        odn.get_logits(sd, T(g["hidden"]), logits_mode=3)
