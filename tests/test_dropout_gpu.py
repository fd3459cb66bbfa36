"""Train-mode dropout (GPU): the reference drops units at three kinds of site in training (nn.Dropout after the embedding
LayerNorm, MuseDiffusion/models/network.py:149; HF BertSelfOutput / BertOutput hidden dropout and BertSelfAttention probability
dropout, bert-base's 0.1, network.py:44-46).  Checked here:
  * kernel level: the Philox masks (rate, repeatability, independence of offset), the GEMM epilogue against the standalone
    kernel, the fused attention forward / backward against torch autograd fed the SAME mask (decoded from the keep bits);
  * model level (fp32 parity mode): training_losses + gradients with the oracle's masks INJECTED against the reference's
    recorded results (tests/golden/losses_tiny_dropout.npz; tolerances of tests/test_training_gpu.py);
  * eval mode and p = 0 leave every existing golden untouched (those tests build their models with the dropouts at 0)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402
from musediffusion_amd import _lib, ops  # noqa: E402
from musediffusion_amd._lib import MH_BF16, MH_F32, check, current_stream, lib, ptr  # noqa: E402
from musediffusion_amd.models.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps  # noqa: E402
from musediffusion_amd.models.network import TransformerNetModel  # noqa: E402
from oracle import fixtures as fx  # noqa: E402
from test_training_gpu import CpuDraws, close  # noqa: E402

DEV = "cuda"


def desc(p, seed=1234, offset=7, mask=None):
    d = _lib.Dropout()
    d.p, d.seed, d.offset, d.mask = p, seed, offset, ptr(mask)
    return d


def dropout_fwd(x, d, dt):
    out = torch.empty_like(x)
    check(lib().mh_dropout_fwd(ptr(x), x.shape[1], ptr(out), out.shape[1], x.shape[0], x.shape[1], dt, C.byref(d), current_stream()))
    return out


def _bit_coords(L, device):
    """(word index within a (batch, head), bit) of every probability (q, k): the lane-native layout of csrc/common.h drop_word_index"""
    nb = (L + 31) // 32
    nkp = (nb + 1) // 2
    q = torch.arange(L, device=device).view(L, 1)
    k = torch.arange(L, device=device).view(1, L)
    kk = k & 31
    h = (kk >> 2) & 1
    r = (kk & 3) + 4 * (kk >> 3)
    word = (((q >> 5) * nkp + (k >> 6)) << 6) + (q & 31) + 32 * h
    bit = r + 16 * ((k >> 5) & 1)
    return word.expand(L, L), bit.expand(L, L), nb * nkp * 64


def bits_to_mask(bits, BH, L):
    """keep-bit tensor -> bool [BH, L(q), L(k)]"""
    word, bit, per_bh = _bit_coords(L, bits.device)
    w = bits.view(BH, per_bh).to(torch.int64) & 0xffffffff
    return ((w[:, word.reshape(-1)] >> bit.reshape(-1)) & 1).bool().view(BH, L, L)


def mask_to_bits(mask):
    """bool [BH, L, L] -> int32 keep-bit tensor (test helper: how the oracle's attention masks are injected)"""
    BH, L, _ = mask.shape
    word, bit, per_bh = _bit_coords(L, mask.device)
    out = torch.zeros(BH, per_bh, dtype=torch.int64, device=mask.device)
    out.scatter_add_(1, word.reshape(1, -1).expand(BH, -1), mask.reshape(BH, -1).to(torch.int64) << bit.reshape(1, -1))
    out = torch.where(out >= 2 ** 31, out - 2 ** 32, out)
    return out.to(torch.int32).reshape(-1).contiguous()


@pytest.mark.parametrize("dt,td", [(MH_F32, torch.float32), (MH_BF16, torch.bfloat16)])
def test_dense_mask_rate_repeatability_and_injection(dt, td):
    rows, cols, p = 4096, 512, 0.1
    x = torch.ones(rows, cols, device=DEV, dtype=td)
    y = dropout_fwd(x, desc(p), dt).float()
    kept = y != 0
    rate = 1.0 - float(kept.float().mean())
    assert abs(rate - p) < 3e-3, rate                                # 2M draws: sigma = 2e-4
    assert torch.allclose(y[kept], torch.full_like(y[kept], 1 / (1 - p)), rtol=1e-2 if dt == MH_BF16 else 1e-6)
    assert torch.equal(y, dropout_fwd(x, desc(p), dt).float())        # counter-based: same descriptor, same mask
    y2 = dropout_fwd(x, desc(p, offset=8), dt).float()
    agree = float(((y2 != 0) == kept).float().mean())
    assert abs(agree - (0.9 * 0.9 + 0.1 * 0.1)) < 5e-3, agree        # another offset: an independent mask
    # rows / columns are not correlated: per-column and per-row keep rates are all near 1 - p
    assert float((kept.float().mean(0) - 0.9).abs().max()) < 0.03 and float((kept.float().mean(1) - 0.9).abs().max()) < 0.06
    inj = (torch.rand(rows, cols, device=DEV) > 0.3).to(torch.uint8)
    y3 = dropout_fwd(x, desc(0.3, mask=inj), dt).float()
    assert torch.equal(y3 != 0, inj.bool())
    assert torch.equal(dropout_fwd(x, desc(0.0), dt), x)              # p = 0: identity


@pytest.mark.parametrize("dt,td,M,N,K", [(MH_F32, torch.float32, 96, 64, 64), (MH_BF16, torch.bfloat16, 1024, 512, 256),
                                         (MH_BF16, torch.bfloat16, 300, 128, 2048)])
def test_gemm_dropout_epilogue_matches_standalone_mask(dt, td, M, N, K):
    torch.manual_seed(0)
    A = torch.randn(M, K, device=DEV).to(td)
    W = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(td)
    b = torch.randn(N, device=DEV)
    R = torch.randn(M, N, device=DEV).to(td)
    d = desc(0.1, seed=99, offset=(5 << 16) | 3)
    out = torch.empty(M, N, device=DEV, dtype=td)
    check(lib().mh_gemm_bias_dropout_res(ptr(A), K, ptr(W), K, ptr(b), ptr(R), N, ptr(out), N, M, N, K, dt, C.byref(d), current_stream()))
    keep = dropout_fwd(torch.ones(M, N, device=DEV, dtype=td), d, dt).float() != 0      # the mask the backward re-creates
    ref = (A.float() @ W.float().t() + b) * keep / 0.9 + R.float()
    tol = 2e-2 if dt == MH_BF16 else 1e-4
    assert float((out.float() - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))
    assert 0.07 < 1 - float(keep.float().mean()) < 0.13


@pytest.mark.parametrize("M,K,p", [(1024, 2048, 0.1), (300, 512, 0.1), (640, 256, 0.0)])
def test_dense_dropout_residual_layernorm_one_kernel(M, K, p):
    """mh_gemm_bias_dropout_res_ln (HF BertSelfOutput / BertOutput in train mode as one kernel, N = 512): its un-normalised rows are
    bit-identical with the dense + dropout + residual kernel's (same Philox mask, same arithmetic order), its normalised rows are
    the LayerNorm of those bf16 rows up to one rounding of the result"""
    torch.manual_seed(1)
    N, td = 512, torch.bfloat16
    A = torch.randn(M, K, device=DEV).to(td)
    W = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(td)
    b = torch.randn(N, device=DEV)
    R = torch.randn(M, N, device=DEV).to(td)
    g, bt = 1 + 0.1 * torch.randn(N, device=DEV), 0.1 * torch.randn(N, device=DEV)
    d = desc(p, seed=77, offset=(3 << 16) | 5)
    two = torch.empty(M, N, device=DEV, dtype=td)
    check(lib().mh_gemm_bias_dropout_res(ptr(A), K, ptr(W), K, ptr(b), ptr(R), N, ptr(two), N, M, N, K, MH_BF16, C.byref(d), current_stream()))
    pre = torch.full((M, N), 7.0, device=DEV, dtype=td)
    out = torch.full((M, N), 7.0, device=DEV, dtype=td)
    check(lib().mh_gemm_bias_dropout_res_ln(ptr(A), K, ptr(W), K, ptr(b), ptr(R), N, ptr(g), ptr(bt), 1e-12, ptr(pre), ptr(out), N, M, N, K,
                                            C.byref(d), current_stream()))
    assert torch.equal(pre, two)
    ref = torch.nn.functional.layer_norm(pre.float(), (N,), g, bt, 1e-12)
    err = (out.float() - ref).abs()
    assert float((err / (ref.abs() + 1.0)).max()) < 2 ** -7, float(err.max())


def stream_inputs(B, L, nh, dh, seed=0):
    torch.manual_seed(seed)
    H = nh * dh
    qkv = (torch.randn(B * L, 3 * H, device=DEV) * 0.7).to(torch.bfloat16)
    return qkv, H


def stream_fwd(qkv, B, L, nh, dh, d=None, bits=None, bits_in=0):
    H = nh * dh
    es, ld, L_, st = 2, 3 * H, lib(), current_stream()
    vt = torch.zeros(B * nh * dh * L + 256, device=DEV, dtype=torch.bfloat16)
    check(L_.mh_head_permute(qkv.data_ptr() + 2 * H * es, ptr(vt), ld, B, L, nh, dh, 3, MH_BF16, st))
    out = torch.empty(B * L, H, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B * nh * L, device=DEV, dtype=torch.float32)
    check(L_.mh_attention_stream_fwd_drop(qkv.data_ptr(), qkv.data_ptr() + H * es, ptr(vt), ptr(out), H, 0, B, L, nh, dh,
                                          1 / math.sqrt(dh), ptr(lse), L * ld, dh, ld, C.byref(d) if d is not None else None, ptr(bits), bits_in, st))
    return out, lse


def torch_attention(qkv, B, L, nh, dh, keep, p):
    H = nh * dh
    q, k, v = [t.float().view(B, L, nh, dh).permute(0, 2, 1, 3) for t in qkv.split(H, dim=1)]
    P = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), dim=-1)
    if keep is not None:
        P = P * keep.view(B, nh, L, L) / (1 - p)
    return (P @ v).permute(0, 2, 1, 3).reshape(B * L, H)


@pytest.mark.parametrize("L,dh", [(512, 64), (528, 32), (1024, 64)])
def test_streaming_attention_dropout_forward_bits_and_values(L, dh):
    B, nh, p = 2, 2, 0.1
    qkv, H = stream_inputs(B, L, nh, dh)
    d = desc(p, seed=4242, offset=(3 << 16) | 2)
    nwords = int(lib().mh_dropout_bits_words(B * nh, L))
    bits = torch.zeros(nwords, device=DEV, dtype=torch.int32)
    out, _ = stream_fwd(qkv, B, L, nh, dh, d, bits, 0)
    # the standalone generator produces the same bits (what the materialised path uses)
    bits2 = torch.zeros(nwords, device=DEV, dtype=torch.int32)
    check(lib().mh_dropout_bits(ptr(bits2), B * nh, L, C.byref(d), current_stream()))
    keep, keep2 = bits_to_mask(bits, B * nh, L), bits_to_mask(bits2, B * nh, L)
    assert torch.equal(keep, keep2)
    rate = 1 - float(keep.float().mean())
    assert abs(rate - p) < 4e-3, rate
    ref = torch_attention(qkv, B, L, nh, dh, keep, p)
    assert float((out.float() - ref).abs().max()) < 3e-2
    # reading the bits back (pre-pass / injection mode) gives the same output up to the summation order of the two block geometries
    # (the generator runs on 8 waves x 128-key stages, the reader on 16 x 256); p = 0 equals the plain kernel
    out2, _ = stream_fwd(qkv, B, L, nh, dh, d, bits, 1)
    assert float((out.float() - out2.float()).abs().max()) < 2e-2
    assert float((out2.float() - ref).abs().max()) < 3e-2
    out0, _ = stream_fwd(qkv, B, L, nh, dh, None, None, 0)
    assert float((out0.float() - torch_attention(qkv, B, L, nh, dh, None, 0)).abs().max()) < 3e-2
    assert float((out0.float() - out.float()).abs().max()) > 1e-2


@pytest.mark.parametrize("L,dh", [(512, 64), (528, 32)])
def test_streaming_attention_dropout_backward_matches_autograd(L, dh):
    from musediffusion_amd import training
    B, nh, p = 2, 2, 0.1
    qkv, H = stream_inputs(B, L, nh, dh, seed=1)
    torch.manual_seed(5)
    keep = torch.rand(B * nh, L, L, device=DEV) >= p
    drop = training._Drop(p, 1, 1, bits=mask_to_bits(keep))
    x = qkv.clone().requires_grad_(True)
    out = training._Attention.apply(x, B, L, nh, MH_BF16, drop)
    g = (torch.randn(B * L, H, device=DEV) * 0.5).to(torch.bfloat16)
    out.backward(g)
    xr = qkv.float().clone().requires_grad_(True)
    ref = torch_attention(xr, B, L, nh, dh, keep, p)
    ref.backward(g.float())
    assert float((out.float() - ref).abs().max()) < 3e-2
    err = float((x.grad.float() - xr.grad).abs().max())
    scale = float(xr.grad.abs().max())
    cos = float(torch.nn.functional.cosine_similarity(x.grad.float().flatten(), xr.grad.flatten(), dim=0))
    print("dqkv err %.3e (max %.3e), cos %.6f" % (err, scale, cos))
    assert cos > 0.999 and err < 0.03 * scale


def build(tag, compute_dtype="fp32", p=fx.DROPOUT_P):
    c = fx.CONFIGS[tag]
    m = TransformerNetModel(c["E"], c["E"], c["Tt"], c["V"], c["L"], dropout=p, bert_hidden=c["H"], bert_layers=c["nL"],
                            bert_heads=c["nh"], bert_ffn=c["F"], compute_dtype=compute_dtype, bert_hidden_dropout=p,
                            bert_attention_dropout=p)
    m.load_state_dict(fx.state_dict(tag))
    m.train().requires_grad_(True).to(DEV)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    return m, diff, c


def inject(m, tag, p=fx.DROPOUT_P):
    c = fx.CONFIGS[tag]
    masks = {}
    for name, mk in fx.dropout_masks(tag, p).items():
        mk = mk.to(DEV)
        masks[name] = mask_to_bits(mk.reshape(-1, c["L"], c["L"])) if name.endswith(".attn") else mk.reshape(-1, c["H"]).to(torch.uint8)
    m.dropout_masks = masks


@pytest.mark.parametrize("variant", ["plain", "corrupt"])
def test_training_losses_with_injected_masks_match_reference(variant):
    tag = "tiny"
    g = load_golden("losses_tiny_dropout.npz")
    m, diff, c = build(tag)
    inject(m, tag)
    li = fx.loss_inputs(tag)
    batch, t, w = li["batch"], li["t"].to(DEV), li["w"].to(DEV)
    kw = {k: v for k, v in batch.items() if variant == "corrupt" or k != "correct_ids"}
    with CpuDraws(fx.loss_seed(tag)):
        terms = diff.training_losses(m, t, model_kwargs=kw)
    for k in ("mse", "nll", "loss"):
        close("%s %s" % (variant, k), terms[k], g["%s_%s" % (variant, k)], 5e-4)
    (terms["loss"] * w).mean().backward()
    L0, L1 = m.input_transformers.layer[0], m.input_transformers.layer[1]
    close(variant + " grad word_embedding", m.word_embedding.weight.grad, g[variant + "_g_word"], 2e-3)
    close(variant + " grad layer0.query", L0.attention.self.query.weight.grad, g[variant + "_g_q0"], 2e-3)
    close(variant + " grad layer1.value", L1.attention.self.value.weight.grad, g[variant + "_g_v1"], 2e-3)
    close(variant + " grad layer0.output.dense", L0.output.dense.weight.grad, g[variant + "_g_ff2"], 2e-3)
    close(variant + " grad time_embed.0", m.time_embed[0].weight.grad, g[variant + "_g_te0"], 2e-3)
    close(variant + " grad lm_head.bias", m.lm_head.bias.grad, g[variant + "_g_lmb"], 2e-3)


@pytest.mark.parametrize("tag", ["c5s", "c5d"])
@pytest.mark.parametrize("compute_dtype", ["fp32", "bf16"])
def test_training_losses_with_injected_masks_at_config5_shape(compute_dtype, tag):
    """c5s (seq_len 1024, d_model 512): the reference ran in TRAIN mode with the masks of fixtures.dropout_masks; the product fed the
    same masks - in bf16 through the streaming attention's keep-bit reader, the dropout epilogue of the one-kernel dense + LayerNorm
    and the fused backward kernels - against the reference's recorded losses and gradients ("corrupt" variant)."""
    # c5d (round 5): config 5's true depth of 12 layers
    variant = "corrupt"
    g = load_golden("losses_%s_dropout.npz" % tag)
    m, diff, c = build(tag, compute_dtype)
    inject(m, tag)
    li = fx.loss_inputs(tag)
    batch, t, w = li["batch"], li["t"].to(DEV), li["w"].to(DEV)
    with CpuDraws(fx.loss_seed(tag)):
        terms = diff.training_losses(m, t, model_kwargs=dict(batch))
    (terms["loss"] * w).mean().backward()
    L0, L1 = m.input_transformers.layer[0], m.input_transformers.layer[1]
    grads = {"g_word": m.word_embedding.weight.grad, "g_q0": L0.attention.self.query.weight.grad, "g_v1": L1.attention.self.value.weight.grad,
             "g_ff2": L0.output.dense.weight.grad, "g_te0": m.time_embed[0].weight.grad, "g_lmb": m.lm_head.bias.grad}
    for k in ("mse", "nll", "loss"):
        close("%s %s" % (variant, k), terms[k], g["%s_%s" % (variant, k)], 5e-4 if compute_dtype == "fp32" else 3e-2)
    for key, gr in grads.items():
        if compute_dtype == "fp32":
            ref = g["%s_%s" % (variant, key)]
            if float(np.abs(ref).max()) < 1e-6:     # lm_head.bias: the reference's gradient is rounding noise around 0 (1e-8)
                assert float((fx.slim(gr).detach().float().cpu() - torch.from_numpy(ref)).abs().max()) < 1e-8, key
                continue
            close("%s %s" % (variant, key), fx.slim(gr), ref, 2e-3)
        else:
            ref = torch.from_numpy(g["%s_%s" % (variant, key)]).flatten()
            got = fx.slim(gr).detach().float().cpu().flatten()
            if float(ref.abs().max()) < 1e-6:       # lm_head.bias: rounding noise around 0 in the reference (1e-8) - a direction of noise means nothing
                assert float(got.abs().max()) < 1e-6, key
                continue
            cos = float(torch.nn.functional.cosine_similarity(got, ref, dim=0))
            print("bf16 dropout %s: cosine %.5f" % (key, cos))
            assert cos >= 0.99, (key, cos)


def test_train_forward_with_injected_masks_and_eval_identity():
    tag = "tiny"
    g = load_golden("losses_tiny_dropout.npz")
    m, diff, c = build(tag)
    inject(m, tag)
    inp = fx.case_inputs(tag, fx.state_dict(tag)["word_embedding.weight"])
    x, t = inp["fwd_x"].to(DEV), inp["fwd_t"].to(DEV)
    y = m(x, t)
    close("train-mode forward", y, g["fwd_y_train"], 2e-4)
    m.eval()
    y_eval = m(x, t)                                                   # tape path, eval: no dropout
    with torch.no_grad():
        y_inf = m(x, t)                                                # engine path
    assert float((y_eval - y_inf).abs().max()) < 5e-5
    assert float((y_eval - y).abs().max()) > 1e-2


def test_philox_dropout_in_training_step_statistics_and_repeatability():
    """Production mode (no injection): masks from Philox keyed by torch's seed; two models seeded alike drop alike, successive
    calls drop differently, and the loss stays finite / close to the dropout-free loss."""
    tag = "tiny"
    li = fx.loss_inputs(tag)
    batch, t = {k: v for k, v in li["batch"].items() if k != "correct_ids"}, li["t"].to(DEV)
    outs = []
    for rep in range(2):
        torch.manual_seed(77)
        m, diff, c = build(tag)
        runs = []
        for call in range(2):
            m.zero_grad(set_to_none=True)
            with CpuDraws(fx.loss_seed(tag)):
                terms = diff.training_losses(m, t, model_kwargs=batch)
            terms["loss"].mean().backward()
            runs.append((terms["loss"].detach().clone(), m.word_embedding.weight.grad.detach().clone()))
        outs.append(runs)
    assert torch.equal(outs[0][0][0], outs[1][0][0]) and torch.equal(outs[0][0][1], outs[1][0][1])     # same seed, same call number
    assert not torch.equal(outs[0][0][0], outs[0][1][0])                                                # next call: new masks
    m0, diff, c = build(tag, p=0.0)
    with CpuDraws(fx.loss_seed(tag)):
        base = diff.training_losses(m0, t, model_kwargs=batch)["loss"]
    assert torch.isfinite(outs[0][0][0]).all()
    assert float((outs[0][0][0] - base).abs().max()) < 0.5 * float(base.abs().max())


def test_bf16_fused_path_matches_unfused_with_same_masks():
    """seq_len 512 engages the streaming attention (dropout inside the kernel) and the one-node FFN; with the SAME injected masks
    the fused bf16 tape and the materialised bf16 tape agree up to bf16 rounding."""
    from musediffusion_amd import synthetic, training
    torch.manual_seed(3)
    E, H, B, V, L, p = 32, 128, 2, 97, 512, 0.1
    m = TransformerNetModel(E, E, 32, V, L, dropout=p, bert_hidden=H, bert_layers=2, bert_heads=2, bert_ffn=256, compute_dtype="bf16",
                            bert_hidden_dropout=p, bert_attention_dropout=p)
    m.train().requires_grad_(True).to(DEV)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    gen = torch.Generator().manual_seed(9)
    masks = {"emb": (torch.rand(B * L, H, generator=gen) >= p).to(torch.uint8).to(DEV)}
    for i in range(2):
        masks["l%d.attn" % i] = mask_to_bits((torch.rand(B * 2, L, L, generator=gen) >= p).to(DEV))
        masks["l%d.ao" % i] = (torch.rand(B * L, H, generator=gen) >= p).to(torch.uint8).to(DEV)
        masks["l%d.ffn" % i] = (torch.rand(B * L, H, generator=gen) >= p).to(torch.uint8).to(DEV)
    m.dropout_masks = masks
    batch = {k: v % V for k, v in synthetic.training_batch(B, L, seed=4).items() if k != "length"}
    batch["input_mask"] = synthetic.training_batch(B, L, seed=4)["input_mask"]
    t = torch.tensor([100, 1500], device=DEV)
    outs = []
    for fused in (True, False):
        training.FUSED_ATTENTION = training.FUSED_FFN = fused
        try:
            m.zero_grad(set_to_none=True)
            with CpuDraws(5):
                terms = diff.training_losses(m, t, model_kwargs=batch)
            terms["loss"].mean().backward()
            outs.append((terms["loss"].detach().cpu(), torch.cat([p_.grad.flatten() for p_ in m.parameters()]).cpu()))
        finally:
            training.FUSED_ATTENTION = training.FUSED_FFN = True
    assert torch.allclose(outs[0][0], outs[1][0], rtol=3e-2, atol=3e-2)
    cos = float(torch.nn.functional.cosine_similarity(outs[0][1], outs[1][1], dim=0))
    print("fused vs unfused gradient cosine %.6f" % cos)
    assert cos > 0.995
