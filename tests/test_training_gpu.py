"""Training-path parity (GPU): training_losses of the product (fp32 compute mode) against the reference's
recorded losses and gradients (tests/golden/losses_tiny.npz, made with dropout 0), with the reference's random
draws reproduced by seeding the CPU generator and feeding the same tensors through `torch.randn_like` patches.

Tolerances: losses 5e-4 rel, gradients 2e-3 of the gradient's max-abs (fp32 kernels, different summation
orders than MKL / autograd)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402
from musediffusion_amd.models.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps  # noqa: E402
from musediffusion_amd.models.network import TransformerNetModel  # noqa: E402
from oracle import fixtures as fx  # noqa: E402

DEV = "cuda"


def build(tag, compute_dtype="fp32"):
    c = fx.CONFIGS[tag]
    m = TransformerNetModel(c["E"], c["E"], c["Tt"], c["V"], c["L"], dropout=0.0, bert_hidden=c["H"],
                            bert_layers=c["nL"], bert_heads=c["nh"], bert_ffn=c["F"], compute_dtype=compute_dtype, bert_hidden_dropout=0.0, bert_attention_dropout=0.0)
    m.load_state_dict(fx.state_dict(tag))
    m.train().requires_grad_(True).to(DEV)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    return m, diff, c


class CpuDraws:
    """Makes torch.randn_like(x) on the device return the draws a seeded CPU generator produces (the reference
    fixture was generated on CPU): same call order, same shapes."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.orig = torch.randn_like

    def __enter__(self):
        def fake(x, **kw):
            return torch.randn(x.shape, generator=self.g, dtype=torch.float32).to(x.device)
        torch.randn_like = fake
        return self

    def __exit__(self, *a):
        torch.randn_like = self.orig


def close(name, got, ref, rel):
    got = got.detach().float().cpu()
    ref = torch.from_numpy(np.asarray(ref))
    err = float((got - ref).abs().max())
    scale = float(ref.abs().max()) + 1e-12
    print("%s: max abs err %.3e (ref absmax %.3e)" % (name, err, scale))
    assert err <= rel * scale, "%s: err %.3e > %.1e * %.3e" % (name, err, rel, scale)


@pytest.mark.parametrize("variant", ["plain", "corrupt"])
def test_training_losses_and_grads_match_reference(variant):
    tag = "tiny"
    g = load_golden("losses_tiny.npz")
    m, diff, c = build(tag)
    li = fx.loss_inputs(tag)
    batch, t, w = li["batch"], li["t"].to(DEV), li["w"].to(DEV)
    kw = {k: v for k, v in batch.items() if variant == "corrupt" or k != "correct_ids"}
    # the fixture seeded the GLOBAL cpu generator: torch.manual_seed(s) then randn_like in reference order
    with CpuDraws(fx.loss_seed(tag)) as d:
        d.g = torch.Generator().manual_seed(fx.loss_seed(tag))
        terms = diff.training_losses(m, t, model_kwargs=kw)
    for k in ("mse", "nll", "loss"):
        close("%s %s" % (variant, k), terms[k], g["%s_%s" % (variant, k)], 5e-4)
    (terms["loss"] * w).mean().backward()
    close(variant + " grad word_embedding", m.word_embedding.weight.grad, g[variant + "_g_word"], 2e-3)
    close(variant + " grad layer0.query", m.input_transformers.layer[0].attention.self.query.weight.grad, g[variant + "_g_q0"], 2e-3)
    close(variant + " grad time_embed.0", m.time_embed[0].weight.grad, g[variant + "_g_te0"], 2e-3)
    close(variant + " grad lm_head.bias", m.lm_head.bias.grad, g[variant + "_g_lmb"], 2e-3)
    for n, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n


@pytest.mark.parametrize("variant", ["plain", "corrupt"])
def test_training_losses_with_predict_xstart_false_match_reference(variant):
    """`_x0_helper` (diffusion.py:577-592) with predict_xstart=False: the model output is read as the noise, the t == 0 rows' loss and
    the logged nll use x0 = sqrt(1 / ab_t) x_t - sqrt(1 / ab_t - 1) eps while the t > 0 rows still compare the raw output with the target
    latent.  Fixture: tests/golden/losses_tiny_eps.npz, the reference run with that flag (tools/make_golden.py eps)."""
    tag = "tiny"
    g = load_golden("losses_tiny_eps.npz")
    m, _, c = build(tag)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=False)
    li = fx.loss_inputs(tag)
    batch, t, w = li["batch"], li["t"].to(DEV), li["w"].to(DEV)
    assert int(t[0]) == 0                      # the row that takes the x0-from-eps branch
    kw = {k: v for k, v in batch.items() if variant == "corrupt" or k != "correct_ids"}
    with CpuDraws(fx.loss_seed(tag)):
        terms = diff.training_losses(m, t, model_kwargs=kw)
    for k in ("mse", "nll", "loss"):
        close("eps %s %s" % (variant, k), terms[k], g["%s_%s" % (variant, k)], 5e-4)
    assert not np.allclose(g[variant + "_nll"], load_golden("losses_tiny.npz")[variant + "_nll"])   # the flag changes what is measured
    (terms["loss"] * w).mean().backward()
    close("eps " + variant + " grad word_embedding", m.word_embedding.weight.grad, g[variant + "_g_word"], 2e-3)
    close("eps " + variant + " grad layer0.query", m.input_transformers.layer[0].attention.self.query.weight.grad, g[variant + "_g_q0"], 2e-3)
    close("eps " + variant + " grad time_embed.0", m.time_embed[0].weight.grad, g[variant + "_g_te0"], 2e-3)
    close("eps " + variant + " grad lm_head.bias", m.lm_head.bias.grad, g[variant + "_g_lmb"], 2e-3)


@pytest.mark.parametrize("tag", ["c5s", "c5d"])
@pytest.mark.parametrize("compute_dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("variant", ["plain", "corrupt"])
def test_training_losses_at_config5_shape_match_reference(variant, compute_dtype, tag):
    """c5s = BASELINE config 5's seq_len 1024 at d_model 512 (2 layers, 2 sequences), losses and six gradients recorded from the
    REFERENCE (tools/make_golden.py bench).  fp32 mode: the parity tolerances of the tiny fixture.  bf16 mode - the benchmarked
    training path: streaming attention forward + fused backward, the one-kernel dense + LayerNorm (N = 512), k-major weight
    gradients - is held to: losses 3e-2, every recorded gradient cosine >= 0.99 and max error <= 6 % of its max-abs."""
    # c5d (round 5): the same at config 5's TRUE depth of 12 layers - the tape's error growth through twelve layers against the reference
    g = load_golden("losses_%s.npz" % tag)
    m, diff, c = build(tag, compute_dtype)
    li = fx.loss_inputs(tag)
    batch, t, w = li["batch"], li["t"].to(DEV), li["w"].to(DEV)
    kw = {k: v for k, v in batch.items() if variant == "corrupt" or k != "correct_ids"}
    with CpuDraws(fx.loss_seed(tag)):
        terms = diff.training_losses(m, t, model_kwargs=kw)
    (terms["loss"] * w).mean().backward()
    L0, L1 = m.input_transformers.layer[0], m.input_transformers.layer[1]
    grads = {"g_word": m.word_embedding.weight.grad, "g_q0": L0.attention.self.query.weight.grad, "g_te0": m.time_embed[0].weight.grad,
             "g_lmb": m.lm_head.bias.grad, "g_v1": L1.attention.self.value.weight.grad, "g_ff2": L0.output.dense.weight.grad}
    if compute_dtype == "fp32":
        for k in ("mse", "nll", "loss"):
            close("%s %s" % (variant, k), terms[k], g["%s_%s" % (variant, k)], 5e-4)
        for key, gr in grads.items():
            ref = g["%s_%s" % (variant, key)]
            if float(np.abs(ref).max()) < 1e-6:     # lm_head.bias at this shape: the reference's gradient is rounding noise around 0 (1e-8)
                assert float((fx.slim(gr).detach().float().cpu() - torch.from_numpy(ref)).abs().max()) < 1e-8, key
                continue
            close("%s %s" % (variant, key), fx.slim(gr), ref, 2e-3)
    else:
        for k in ("mse", "nll", "loss"):
            close("%s %s (bf16)" % (variant, k), terms[k], g["%s_%s" % (variant, k)], 3e-2)
        for key, gr in grads.items():
            ref = torch.from_numpy(g["%s_%s" % (variant, key)]).flatten()
            got = fx.slim(gr).detach().float().cpu().flatten()
            if float(ref.abs().max()) < 1e-6:       # lm_head.bias at this shape: rounding noise around 0 in the reference (1e-8)
                assert float(got.abs().max()) < 1e-6, key
                continue
            cos = float(torch.nn.functional.cosine_similarity(got, ref, dim=0))
            err = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
            print("bf16 %s %s: cosine %.5f, max err / absmax %.4f" % (variant, key, cos, err))
            assert cos >= 0.99 and err <= 0.06, (key, cos, err)


def test_private_helpers_of_the_reference_surface():
    """_get_x_start, _token_discrete_loss, _x0_helper (diffusion.py:542-592): the reference calls them on the diffusion object from
    training_losses_seq2seq (:614, :629, :641) and a subclass or notebook may too.  Same signatures; values against the oracle."""
    import inspect
    from musediffusion_amd.models.diffusion import GaussianDiffusion, _extract_into_tensor
    from oracle import denoiser as odn, losses as olo, sampling as osa, schedule as osc
    assert list(inspect.signature(GaussianDiffusion._get_x_start).parameters) == ["x_start_mean", "std"]
    assert list(inspect.signature(GaussianDiffusion._token_discrete_loss).parameters) == ["x_t", "get_logits", "input_ids", "mask"]
    assert list(inspect.signature(GaussianDiffusion._x0_helper).parameters) == ["self", "model_output", "x", "t"]
    tag = "tiny"
    m, diff, c = build(tag)
    sd = fx.state_dict(tag)
    li = fx.loss_inputs(tag)
    ids, mask = li["batch"]["input_ids"].to(DEV), li["batch"]["input_mask"].to(DEV)
    mean = m.get_embeds(ids)
    std = _extract_into_tensor(diff.sqrt_one_minus_alphas_cumprod, torch.tensor([0], device=DEV), mean.shape)     # diffusion.py:610-612
    with CpuDraws(77):
        x0 = diff._get_x_start(mean, std)
    z = torch.randn(mean.shape, generator=torch.Generator().manual_seed(77))
    assert torch.equal(x0.cpu(), mean.cpu() + std.cpu() * z)                            # bit for bit: one product, one sum per element
    with CpuDraws(78):
        x1 = GaussianDiffusion._get_x_start(mean, 0.25 + torch.rand(mean.shape, device=DEV))    # a general elementwise std
    assert x1.shape == mean.shape and torch.isfinite(x1).all()
    # _token_discrete_loss: bound get_logits (tape path) and a foreign callable (generic path), with and without the mask
    logits_fn = lambda h: odn.get_logits(sd, h)
    for mk in (None, mask):
        ref = olo.token_nll(x0.detach().cpu(), logits_fn, ids.cpu(), None if mk is None else mk.cpu().float())
        got = diff._token_discrete_loss(x0.detach(), m.get_logits, ids, mask=mk)
        close("token nll (bound get_logits, mask=%s)" % (mk is not None), got, ref, 2e-5)
        got2 = GaussianDiffusion._token_discrete_loss(x0.detach(), lambda h: m.get_logits(h), ids, mask=mk)
        close("token nll (foreign callable, mask=%s)" % (mk is not None), got2, ref, 2e-5)
    x0g = x0.detach().clone().requires_grad_(True)
    diff._token_discrete_loss(x0g, m.get_logits, ids).sum().backward()                   # differentiable through the kernel tape
    assert x0g.grad is not None and torch.isfinite(x0g.grad).all() and float(x0g.grad.abs().max()) > 0
    # _x0_helper
    d = osc.make_diffusion()
    t = li["t"].to(DEV)
    out = torch.randn(mean.shape, generator=torch.Generator().manual_seed(79)).to(DEV)
    h = diff._x0_helper(out, x0.detach(), t)
    assert set(h) == {"pred_xprev", "pred_xstart"} and h["pred_xstart"] is out
    ref_prev = (osa.extract(d.posterior_mean_coef1, t.cpu(), out.shape) * out.cpu() + osa.extract(d.posterior_mean_coef2, t.cpu(), out.shape) * x0.detach().cpu())
    close("pred_xprev", h["pred_xprev"], ref_prev, 1e-6)


def test_training_forward_matches_inference_engine():
    m, diff, c = build("tiny")
    inp = fx.case_inputs("tiny", fx.state_dict("tiny")["word_embedding.weight"])
    x, t = inp["fwd_x"].to(DEV), inp["fwd_t"].to(DEV)
    y_train = m(x, t)                       # grad enabled -> tape path
    assert y_train.requires_grad
    with torch.no_grad():
        y_inf = m(x, t)                     # engine path
    assert float((y_train - y_inf).abs().max()) < 5e-5


def test_bf16_training_runs_and_tracks_fp32():
    m32, diff, c = build("same", "fp32")
    m16, _, _ = build("same", "bf16")
    li = fx.loss_inputs("same")
    batch, t = li["batch"], li["t"].to(DEV)
    outs = []
    for m in (m32, m16):
        with CpuDraws(5):
            terms = diff.training_losses(m, t, model_kwargs=batch)
        terms["loss"].mean().backward()
        outs.append((terms["loss"].detach().cpu(), m.word_embedding.weight.grad.detach().cpu()))
    assert torch.allclose(outs[0][0], outs[1][0], rtol=3e-2, atol=3e-2)
    cos = torch.nn.functional.cosine_similarity(outs[0][1].flatten(), outs[1][1].flatten(), dim=0)
    assert float(cos) > 0.99, float(cos)


def test_ddp_wrapped_training_step_single_rank_nccl():
    """DistributedDataParallel over RCCL (backend 'nccl') wraps the model exactly like the reference's TrainLoop
    (utils/train_util.py:106-116); gradients through DDP equal the plain ones."""
    import os
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    from conftest import free_port
    os.environ["MASTER_PORT"] = str(free_port())
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        m, diff, c = build("tiny")
        li = fx.loss_inputs("tiny")
        batch, t = li["batch"], li["t"].to(DEV)
        with CpuDraws(9):
            ref_terms = diff.training_losses(m, t, model_kwargs=batch)
        ref_terms["loss"].mean().backward()
        ref_grads = {n: p.grad.clone() for n, p in m.named_parameters()}
        m.zero_grad(set_to_none=True)
        ddp = DDP(m, device_ids=[torch.cuda.current_device()], broadcast_buffers=False, bucket_cap_mb=128,
                  find_unused_parameters=False)
        with CpuDraws(9):
            terms = diff.training_losses(ddp, t, model_kwargs=batch)
        terms["loss"].mean().backward()
        for n, p in m.named_parameters():
            assert torch.allclose(p.grad, ref_grads[n], rtol=1e-5, atol=1e-7), n
        # the loss-aware sampler's packed all_gather on the device
        from musediffusion_amd.models.step_sample import create_named_schedule_sampler
        s = create_named_schedule_sampler("lossaware", diff)
        ts, w = s.sample(4, DEV)
        s.update_with_local_losses(ts, terms["loss"].detach()[:4] if terms["loss"].numel() >= 4 else terms["loss"].detach().repeat(2))
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("L", [512, 528])
def test_fused_attention_training_path_matches_unfused(L):
    """seq_len 512 engages the streaming attention forward + fused backward kernels and the one-node FFN (GELU backward in a
    GEMM epilogue) in training_losses (bf16): same loss and gradients as the unfused tape, up to bf16 rounding"""
    from musediffusion_amd import synthetic, training
    torch.manual_seed(3)
    E, H, B, V = 32, 128, 2, 97
    m = TransformerNetModel(E, E, 32, V, L, dropout=0.0, bert_hidden=H, bert_layers=2, bert_heads=2, bert_ffn=256,
                            compute_dtype="bf16", bert_hidden_dropout=0.0, bert_attention_dropout=0.0)
    m.train().requires_grad_(True).to(DEV)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    gen = torch.Generator().manual_seed(9)
    ids = torch.randint(3, V, (B, L), generator=gen)
    mask = torch.ones(B, L, dtype=torch.long)
    mask[:, :12] = 0
    batch = {"input_ids": ids, "input_mask": mask, "correct_ids": ids.clone()}
    t = torch.tensor([400, 1500], device=DEV)
    res = []
    for fused in (True, False):
        training.FUSED_ATTENTION = training.FUSED_FFN = fused
        m.zero_grad(set_to_none=True)
        with CpuDraws(11):
            terms = diff.training_losses(m, t, model_kwargs=batch)
        terms["loss"].mean().backward()
        res.append((terms["loss"].detach().float().cpu(),
                    {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}))
    training.FUSED_ATTENTION = training.FUSED_FFN = True
    assert torch.allclose(res[0][0], res[1][0], rtol=2e-2, atol=2e-2), (res[0][0], res[1][0])
    for name in ("input_transformers.layer.0.attention.self.query.weight", "input_transformers.layer.0.attention.self.key.weight",
                 "input_transformers.layer.1.attention.self.value.weight", "input_transformers.layer.0.intermediate.dense.weight",
                 "input_transformers.layer.1.output.dense.weight", "input_transformers.layer.0.intermediate.dense.bias",
                 "word_embedding.weight", "time_embed.0.weight"):
        a, b = res[0][1][name].flatten(), res[1][1][name].flatten()
        cos = float(torch.nn.functional.cosine_similarity(a, b, dim=0))
        print("%s: cosine %.5f, |fused| %.3e |unfused| %.3e" % (name, cos, float(a.norm()), float(b.norm())))
        assert cos > 0.995, (name, cos)
        assert abs(float(a.norm()) / float(b.norm()) - 1.0) < 0.03, name


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_dense_dropout_layernorm_as_one_kernel_matches_two(p):
    """d_model 512: BertSelfOutput / BertOutput (dense -> dropout -> + input -> LayerNorm) as ONE kernel (mh_gemm_bias_dropout_res_ln)
    against the dense kernel + the LayerNorm kernel: same Philox masks (same call number), same bf16-rounded pre-LayerNorm rows -
    the losses agree to bf16 rounding and every gradient points the same way"""
    from musediffusion_amd import training
    torch.manual_seed(6)
    E, H, B, V, L = 32, 512, 2, 97, 128
    m = TransformerNetModel(E, E, 32, V, L, dropout=p, bert_hidden=H, bert_layers=2, bert_heads=8, bert_ffn=1024,
                            compute_dtype="bf16", bert_hidden_dropout=p, bert_attention_dropout=0.0)
    m.train().requires_grad_(True).to(DEV)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    gen = torch.Generator().manual_seed(9)
    ids = torch.randint(3, V, (B, L), generator=gen)
    batch = {"input_ids": ids, "input_mask": torch.ones(B, L, dtype=torch.long), "correct_ids": ids.clone()}
    t = torch.tensor([400, 1500], device=DEV)
    res = []
    try:
        for fused in (True, False):
            training.FUSED_DENSE_LN = fused
            m._dropout_calls = 0                       # the same Philox counters in both runs
            m.zero_grad(set_to_none=True)
            with CpuDraws(11):
                terms = diff.training_losses(m, t, model_kwargs=batch)
            terms["loss"].mean().backward()
            res.append((terms["loss"].detach().float().cpu(),
                        {n: q.grad.detach().float().cpu().clone() for n, q in m.named_parameters() if q.grad is not None}))
    finally:
        training.FUSED_DENSE_LN = True
    assert torch.allclose(res[0][0], res[1][0], rtol=5e-3, atol=5e-3), (res[0][0], res[1][0])
    assert res[0][1].keys() == res[1][1].keys()
    for name in res[0][1]:
        a, b = res[0][1][name].flatten(), res[1][1][name].flatten()
        if float(b.norm()) == 0.0:
            assert float(a.norm()) == 0.0, name
            continue
        if name.endswith("attention.self.key.bias"):      # softmax is invariant to a key bias: that gradient is rounding noise
            continue
        cos = float(torch.nn.functional.cosine_similarity(a, b, dim=0))
        assert cos > 0.999, (name, cos)
        assert abs(float(a.norm()) / float(b.norm()) - 1.0) < 0.02, name


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_micro_batch_accumulation_adds_the_separate_gradients_exactly(p):
    """TrainStep.forward_backward (train_util.py:199-226: micro-batches back-propagated one after the other, gradients adding up
    without a division) at d_model 512 / seq_len 512 in bf16: the accumulated gradient of two micro-batches is bit-for-bit the sum of
    the gradients of the two run on their own - the kernels are deterministic (Philox dropout masks included: they are keyed by
    the forward call number) and nothing else touches the gradients"""
    from musediffusion_amd.train_step import TrainStep

    class Fixed:                                            # a sampler with a scripted sequence of timesteps, unit weights
        def __init__(self, ts):
            self.ts = list(ts)

        def sample(self, n, dev):
            t = torch.tensor(self.ts[:n], device=dev)
            self.ts = self.ts[n:]
            return t, torch.ones(n, device=dev)

    class NoOpt:
        def grad_norm(self):
            return torch.zeros(1)

        def step(self, lr=None):
            pass

    torch.manual_seed(8)
    E, H, B, V, L = 32, 512, 4, 97, 512
    m = TransformerNetModel(E, E, 32, V, L, dropout=p, bert_hidden=H, bert_layers=2, bert_heads=8, bert_ffn=1024,
                            compute_dtype="bf16", bert_hidden_dropout=p, bert_attention_dropout=p)
    m.train().requires_grad_(True).to(DEV)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    gen = torch.Generator().manual_seed(10)
    ids = torch.randint(3, V, (B, L), generator=gen)
    cond = {"input_ids": ids, "input_mask": torch.ones(B, L, dtype=torch.long), "correct_ids": ids.clone()}
    ts = [100, 900, 1500, 30]

    def grads():
        return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    parts = []
    m._dropout_calls = 0                                    # (dropout masks are keyed by the forward call number: 1, 2 in both runs)
    with CpuDraws(21):                                      # the two halves on their own, drawing in the accumulated run's order
        for i in (0, 2):
            step = TrainStep(m, diff, microbatch=2, schedule_sampler=Fixed(ts[i:i + 2]), optimizer=NoOpt())
            step.forward_backward({k: v[i:i + 2] for k, v in cond.items()})
            parts.append(grads())
    m._dropout_calls = 0
    with CpuDraws(21):
        step = TrainStep(m, diff, microbatch=2, schedule_sampler=Fixed(ts), optimizer=NoOpt())
        step.forward_backward(cond)
    both = grads()
    assert both.keys() == parts[0].keys() == parts[1].keys()
    for n in both:
        assert torch.equal(both[n], parts[0][n] + parts[1][n]), n


def test_one_launch_weight_copies_give_the_same_tape():
    """training._WeightPrep (one launch makes the bf16 copies and transposes of every encoder weight) against the per-layer casts
    and transposes: identical operands, so losses and every gradient are bit-identical; the copies follow an in-place weight update"""
    from musediffusion_amd import training
    torch.manual_seed(5)
    E, H, B, V, L = 32, 128, 2, 97, 64
    m = TransformerNetModel(E, E, 32, V, L, dropout=0.0, bert_hidden=H, bert_layers=2, bert_heads=2, bert_ffn=256,
                            compute_dtype="bf16", bert_hidden_dropout=0.0, bert_attention_dropout=0.0)
    m.train().requires_grad_(True).to(DEV)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    gen = torch.Generator().manual_seed(9)
    ids = torch.randint(3, V, (B, L), generator=gen)
    batch = {"input_ids": ids, "input_mask": torch.ones(B, L, dtype=torch.long), "correct_ids": ids.clone()}
    t = torch.tensor([400, 1500], device=DEV)

    def run():
        m.zero_grad(set_to_none=True)
        with CpuDraws(11):
            terms = diff.training_losses(m, t, model_kwargs=batch)
        terms["loss"].mean().backward()
        return terms["loss"].detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    try:
        for rnd in range(2):
            training.WEIGHT_PREP = True
            l1, g1 = run()
            assert getattr(m, "_weight_prep", None) is not None
            training.WEIGHT_PREP = False
            l0, g0 = run()
            assert torch.equal(l0, l1)
            assert g0.keys() == g1.keys()
            for n in g0:
                assert torch.equal(g0[n], g1[n]), n
            with torch.no_grad():      # an optimizer-style in-place update: the next forward's copies must see it
                for p in m.parameters():
                    p.add_(0.01 * torch.randn_like(p))
    finally:
        training.WEIGHT_PREP = True


@pytest.mark.parametrize("compute_dtype", ["bf16", "fp32"])
def test_stacked_projection_tape_equals_the_concatenating_tape(compute_dtype):
    """training.TAPE_STACK (round 5): the q | k | v projection's three parameters stacked on the tape only (no fp32 torch.cat per layer and
    step; one bias pack per forward) and the residual branch's gradient added inside the projection's input-gradient GEMM, against round
    4's tape (torch.cat + an autograd add per layer).  Same losses bit for bit (the forward is unchanged); every parameter's gradient
    agrees to the one rounding the new form saves (the sum of the two branches is rounded once instead of twice in bf16; fp32: reorder)."""
    from musediffusion_amd import training
    torch.manual_seed(6)
    E, H, B, V, L = 32, 128, 2, 97, 64
    m = TransformerNetModel(E, E, 32, V, L, dropout=0.0, bert_hidden=H, bert_layers=3, bert_heads=2, bert_ffn=256,
                            compute_dtype=compute_dtype, bert_hidden_dropout=0.0, bert_attention_dropout=0.0)
    m.train().requires_grad_(True).to(DEV)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    gen = torch.Generator().manual_seed(9)
    ids = torch.randint(3, V, (B, L), generator=gen)
    batch = {"input_ids": ids, "input_mask": torch.ones(B, L, dtype=torch.long), "correct_ids": ids.clone()}
    t = torch.tensor([400, 1500], device=DEV)

    def run():
        m.zero_grad(set_to_none=True)
        with CpuDraws(11):
            terms = diff.training_losses(m, t, model_kwargs=batch)
        terms["loss"].mean().backward()
        return terms["loss"].detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    try:
        training.TAPE_STACK = True
        l1, g1 = run()
        training.TAPE_STACK = False
        l0, g0 = run()
    finally:
        training.TAPE_STACK = True
    assert torch.equal(l0, l1) and g0.keys() == g1.keys()
    for n in g0:
        a, b = g1[n].float().flatten(), g0[n].float().flatten()
        if float(b.abs().max()) < 1e-5:          # the key bias: its gradient is zero in exact arithmetic (softmax is shift-invariant per query)
            assert float(a.abs().max()) < 1e-5, n
            continue
        cos = float(torch.nn.functional.cosine_similarity(a, b, dim=0))
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-20)
        assert cos > (0.9995 if compute_dtype == "bf16" else 0.999999) and err < (3e-2 if compute_dtype == "bf16" else 1e-4), (n, cos, err)
    sa = getattr(m.input_transformers.layer[0].attention, "self")
    assert sa.query.weight.grad.data_ptr() % 16 == 0 and sa.key.weight.grad.data_ptr() % 16 == 0       # the optimizer reads 16-byte pieces


@pytest.mark.parametrize("tag,variant", [("c1", "corrupt"), ("same", "plain")])
def test_every_parameter_gradient_matches_oracle(tag, variant):
    """EVERY parameter's gradient (LayerNorm gains / biases, position table, projections, ... - the golden file holds four) of
    training_losses at d_model 128 / 64 against the CPU oracle (torch autograd over the restated reference ops, pinned to the
    reference by tests/test_oracle_golden.py), same draws, fp32 mode: 2e-3 of each gradient's max-abs."""
    from oracle import denoiser as odn, losses as olo, schedule as osc
    m, diff, c = build(tag)
    li = fx.loss_inputs(tag)
    batch, t, w = li["batch"], li["t"], li["w"]
    kw = {k: v for k, v in batch.items() if variant == "corrupt" or k != "correct_ids"}
    with CpuDraws(fx.loss_seed(tag)):
        terms = diff.training_losses(m, t.to(DEV), model_kwargs=kw)
    (terms["loss"] * w.to(DEV)).mean().backward()
    sd = {k: v.clone() for k, v in fx.state_dict(tag).items()}
    pnames = [n for n, _ in m.named_parameters()]
    for n in pnames:
        sd[n].requires_grad_(True)
    sd["lm_head.weight"] = sd["word_embedding.weight"]
    torch.manual_seed(fx.loss_seed(tag))
    ref = olo.training_losses(osc.make_diffusion(), lambda x, ts: odn.forward(sd, x, ts, c["nh"]), lambda ids: odn.get_embeds(sd, ids),
                              lambda h: odn.get_logits(sd, h), t, batch["input_ids"], batch["input_mask"],
                              correct_ids=batch["correct_ids"] if variant == "corrupt" else None)
    (ref["loss"] * w).mean().backward()
    close("loss", terms["loss"], ref["loss"].detach().numpy(), 5e-4)
    bad = []
    for n, p in m.named_parameters():
        r = sd[n].grad
        err = float((p.grad.detach().cpu() - r).abs().max())
        scale = float(r.abs().max())
        # key biases have a zero true gradient (softmax is invariant to a per-query shift): only rounding noise on both sides
        if "attention.self.key.bias" in n:
            assert err < 1e-5, (n, err)
            continue
        if err > 2e-3 * scale + 1e-7:
            bad.append((n, err, scale))
    assert not bad, bad


def test_bf16_fused_training_at_seq_len_1024_against_oracle():
    """BASELINE config 5's sequence length (1024) through the bf16 training kernels that run there - streaming attention forward,
    fused dQ / dK-dV backward, one-node FFN, k-major weight-gradient GEMM - against the fp32 CPU oracle (torch autograd over the
    restated reference ops), same draws, dropout off.  Stated bf16 tolerance: losses 3e-2, every weight-matrix gradient cosine
    > 0.99 against the oracle's (bf16 activations and bf16 gradient tensors through 2 layers)."""
    from oracle import denoiser as odn, losses as olo, schedule as osc
    from musediffusion_amd import synthetic
    E, H, F, nL, nh, V, L, B, Tt = 32, 128, 256, 2, 2, 97, 1024, 2, 32
    sd = odn.random_state_dict(E, H, F, nL, V, L, Tt, seed=31, emb_std=fx.EMB_STD)
    m = TransformerNetModel(E, E, Tt, V, L, dropout=0.0, bert_hidden=H, bert_layers=nL, bert_heads=nh, bert_ffn=F, compute_dtype="bf16",
                            bert_hidden_dropout=0.0, bert_attention_dropout=0.0)
    m.load_state_dict(sd)
    m.train().requires_grad_(True).to(DEV)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    tb = synthetic.training_batch(B, L, seed=6)
    batch = {"input_ids": tb["input_ids"] % V, "correct_ids": tb["correct_ids"] % V, "input_mask": tb["input_mask"]}
    t = torch.tensor([250, 1700])
    with CpuDraws(41):
        terms = diff.training_losses(m, t.to(DEV), model_kwargs=batch)
    terms["loss"].mean().backward()
    ref_sd = {k: v.clone() for k, v in sd.items()}
    names = [n for n, _ in m.named_parameters()]
    for n in names:
        ref_sd[n].requires_grad_(True)
    ref_sd["lm_head.weight"] = ref_sd["word_embedding.weight"]
    torch.manual_seed(41)
    ref = olo.training_losses(osc.make_diffusion(), lambda x, ts: odn.forward(ref_sd, x, ts, nh), lambda ids: odn.get_embeds(ref_sd, ids),
                              lambda h: odn.get_logits(ref_sd, h), t, batch["input_ids"], batch["input_mask"], correct_ids=batch["correct_ids"])
    ref["loss"].mean().backward()
    assert torch.allclose(terms["loss"].detach().cpu(), ref["loss"].detach(), rtol=3e-2, atol=3e-2)
    worst = (1.0, "")
    for n, p in m.named_parameters():
        if p.dim() < 2 or "position_embeddings" in n:
            continue
        cos = float(torch.nn.functional.cosine_similarity(p.grad.detach().cpu().flatten(), ref_sd[n].grad.flatten(), dim=0))
        worst = min(worst, (cos, n))
    print("worst weight-gradient cosine vs the fp32 oracle: %.5f (%s)" % worst)
    assert worst[0] > 0.99, worst


def test_full_size_training_step_is_invariant_to_the_tile_choice(dbg_lib):
    """The benchmarked training shape (BASELINE config 5 per GPU: 32 sequences x seq_len 1024, d_model 512, 12 layers, bf16, dropout 0.1)
    is the only place where the tape's GEMMs take the 256 x 256 tile and the weight-gradient GEMM its 8-wave variant.  Every such form is
    bit-identical with the 256 x 128 form per kernel (tests/test_round3_kernels_gpu.py), so the WHOLE step must be: the same losses and
    the same gradient of every parameter, bit for bit, with the round-3 tile rules on and off (same timesteps, same noise, same Philox
    dropout counters).  Also pins the full-size step's determinism."""
    from musediffusion_amd import _lib, synthetic
    torch.manual_seed(12)
    B, L, E, H, nL = 32, 1024, 128, 512, 12
    m = TransformerNetModel(E, E, 128, synthetic.VOCAB_SIZE, L, dropout=0.1, bert_hidden=H, bert_layers=nL, bert_heads=8, bert_ffn=2048,
                            compute_dtype="bf16")
    m.train().requires_grad_(True).to(DEV)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    batch = {k: v.to(DEV) for k, v in synthetic.training_batch(B, L, seed=5).items()}
    t = torch.arange(B, device=DEV) * 61 % 2000
    L_ = _lib.lib()
    res = []
    try:
        for wide in (1, 0, 1):
            L_.mh_gemm_set_auto_wide(wide)
            L_.mh_gemm_dw_set_wide(wide)
            m._dropout_calls = 0                       # the same Philox counters in every run
            m.zero_grad(set_to_none=True)
            with CpuDraws(21):
                terms = diff.training_losses(m, t, model_kwargs=batch)
            terms["loss"].mean().backward()
            res.append((terms["loss"].detach().clone(), {n: q.grad.detach().clone() for n, q in m.named_parameters() if q.grad is not None}))
    finally:
        L_.mh_gemm_set_auto_wide(1)
        L_.mh_gemm_dw_set_wide(1)
    assert bool(torch.isfinite(res[0][0]).all()) and float(res[0][0].mean()) > 0
    for other in (1, 2):
        assert torch.equal(res[0][0], res[other][0]), "losses differ (run %d)" % other
        assert res[0][1].keys() == res[other][1].keys()
        for name in res[0][1]:
            assert torch.equal(res[0][1][name], res[other][1][name]), (name, other)


def test_lossaware_update_from_device_tensors_is_deferred_and_equals_the_host_update():
    """step_sample.py:90-173: the loss-aware sampler fed device tensors (what TrainStep hands it between forward and backward) only
    enqueues a copy; the state the next `sample()` / `weights()` reads equals, bit for bit, the state a host-side
    update_with_all_losses of the same (timestep, loss) pairs leaves - history windows, counts and weights."""
    from types import SimpleNamespace
    from musediffusion_amd.models.step_sample import LossSecondMomentResampler
    fake = SimpleNamespace(num_timesteps=6)
    a, b = LossSecondMomentResampler(fake, history_per_term=3), LossSecondMomentResampler(fake, history_per_term=3)
    g = torch.Generator().manual_seed(11)
    for it in range(9):
        n = 5 if it % 3 else 70                      # (70 > one padding group of 64)
        ts = torch.randint(0, 6, (n,), generator=g)
        ls = torch.rand(n, generator=g) * 3
        a.update_with_local_losses(ts.to(DEV), ls.to(DEV))
        assert a._pending is not None               # nothing applied yet: no host sync inside the call
        b.update_with_all_losses(ts.tolist(), ls.double().tolist())
        if it % 2:
            np.testing.assert_array_equal(a.weights(), b.weights())
            assert a._pending is None
    np.testing.assert_array_equal(a._loss_history, b._loss_history)
    np.testing.assert_array_equal(a._loss_counts, b._loss_counts)
    assert a._warmed_up() == b._warmed_up()
    np.random.seed(5); ta, wa = a.sample(8, DEV)
    np.random.seed(5); tb, wb = b.sample(8, DEV)
    assert torch.equal(ta, tb) and torch.equal(wa, wb)
