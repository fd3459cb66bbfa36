"""Round 6: the training step's encoder layers on K32 panels (csrc/train_layer.hip, training._EncoderLayer) against the op-per-node tape
of rounds 1 - 5 (row-major operands, the same kernels' arithmetic), plus the kernel forms the panel path adds."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from musediffusion_amd import _lib  # noqa: E402
from musediffusion_amd._lib import check, current_stream, lib  # noqa: E402
from musediffusion_amd.models.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps  # noqa: E402
from musediffusion_amd.models.network import TransformerNetModel  # noqa: E402
from test_training_gpu import CpuDraws  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def to_panel(w):
    r, k = w.shape
    return w.bfloat16().reshape(r, k // 32, 32).permute(1, 0, 2).contiguous().to(DEV)


def from_panel(p):
    return p.permute(1, 0, 2).reshape(p.shape[1], -1)


@pytest.mark.parametrize("rows,cols", [(1000, 512), (4096, 2048), (64, 32)])
def test_repack_panel_round_trip(rows, cols):
    x = rnd(rows, cols, seed=1).bfloat16().to(DEV)
    pan = torch.zeros(cols // 32, rows, 32, device=DEV, dtype=torch.bfloat16)
    check(lib().mh_repack_panel(x.data_ptr(), cols, pan.data_ptr(), rows, rows, cols, 1, current_stream()), "mh_repack_panel")
    assert torch.equal(from_panel(pan), x)
    back = torch.zeros_like(x)
    check(lib().mh_repack_panel(pan.data_ptr(), rows, back.data_ptr(), cols, rows, cols, 0, current_stream()), "mh_repack_panel")
    assert torch.equal(back, x)


@pytest.mark.parametrize("K,M,N", [(4096, 512, 2048), (2048, 1536, 512), (8192, 512, 512), (1024, 2048, 512), (3072, 512, 384), (2080, 256, 128)])
def test_weight_gradient_gemm_on_panels_equals_the_row_major_form(K, M, N):
    """gemm_tn_kernel<.., PANEL>: dW = dY^T X with both operands as K32 panels - the same products and sums in the same order as the
    k-major row-major form (bit-identical partials, bias column sums included), and close to fp32 arithmetic on the same operands."""
    A, B = rnd(K, M, seed=2).bfloat16(), rnd(K, N, seed=3).bfloat16()
    S = int(lib().mh_gemm_dw_splits(K, M, N))
    n = M * N + M
    outs = []
    for panel in (0, 1):
        a = to_panel(A) if panel else A.to(DEV)
        b = to_panel(B) if panel else B.to(DEV)
        part = torch.zeros(S, n, device=DEV)
        check(lib().mh_gemm_dw_bias_ex(a.data_ptr(), K if panel else M, b.data_ptr(), K if panel else N, panel, part.data_ptr(), S, K, M, N, 1,
                                       current_stream()), "mh_gemm_dw_bias_ex")
        outs.append(part.clone())
    assert torch.equal(outs[0], outs[1])
    tot = outs[1].sum(0).cpu()
    ref = A.float().T @ B.float()
    assert float((tot[:M * N].view(M, N) - ref).abs().max()) < 2e-3 * float(ref.abs().max()) + 1e-2
    assert float((tot[M * N:] - A.float().sum(0)).abs().max()) < 1e-2 * math.sqrt(K)


@pytest.mark.parametrize("M,K,p", [(1000, 2048, 0.1), (4096, 512, 0.0), (2048, 2048, 0.1)])
def test_dense_dropout_layernorm_kernel_on_panels_equals_the_row_major_form(M, K, p):
    """mh_gemm_desc_launch with the LayerNorm epilogue: A / W / residual / out as panels and pre_out as rows against mh_gemm_bias_dropout_res_ln
    (all rows): the same tile, the same sums - bit-identical pre-LayerNorm rows and outputs."""
    import ctypes as C
    N = 512
    A, W, R = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=1 / math.sqrt(K)), rnd(M, N, seed=6)
    b, g, be = rnd(N, seed=7, scale=0.3).to(DEV), (1 + rnd(N, seed=8, scale=0.1)).to(DEV), rnd(N, seed=9, scale=0.1).to(DEV)
    d = _lib.Dropout()
    d.p, d.seed, d.offset, d.mask = p, 99, 5, None
    Ar, Wr, Rr = A.bfloat16().to(DEV), W.bfloat16().to(DEV), R.bfloat16().to(DEV)
    pre0, out0 = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16), torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    check(lib().mh_gemm_bias_dropout_res_ln(Ar.data_ptr(), K, Wr.data_ptr(), K, b.data_ptr(), Rr.data_ptr(), N, g.data_ptr(), be.data_ptr(), 1e-12,
                                            pre0.data_ptr(), out0.data_ptr(), N, M, N, K, C.byref(d), current_stream()), "mh_gemm_bias_dropout_res_ln")
    Ap, Wp, Rp = to_panel(A), to_panel(W), to_panel(R)
    pre1 = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    out1 = torch.zeros(N // 32, M, 32, device=DEV, dtype=torch.bfloat16)
    gd = _lib.GemmDesc()
    gd.A, gd.lda, gd.a_panel, gd.W, gd.ldw, gd.w_panel, gd.bias = Ap.data_ptr(), M, 1, Wp.data_ptr(), N, 1, b.data_ptr()
    gd.residual, gd.ldr, gd.r_panel, gd.out, gd.ldo, gd.o_panel = Rp.data_ptr(), M, 1, out1.data_ptr(), M, 1
    gd.pre_out, gd.ldp, gd.p_panel, gd.ln_gamma, gd.ln_beta, gd.ln_eps = pre1.data_ptr(), N, 0, g.data_ptr(), be.data_ptr(), 1e-12
    gd.drop = C.pointer(d)
    gd.M, gd.N, gd.K = M, N, K
    check(lib().mh_gemm_desc_launch(C.byref(gd), current_stream()), "mh_gemm_desc_launch")
    assert torch.equal(pre0, pre1)
    assert torch.equal(out0, from_panel(out1))


def _model(nL, L, B, p_drop, seed=5):
    torch.manual_seed(seed)
    E, H, V = 32, 512, 97
    m = TransformerNetModel(E, E, 32, V, L, dropout=p_drop, bert_hidden=H, bert_layers=nL, bert_heads=8, bert_ffn=1024,
                            compute_dtype="bf16", bert_hidden_dropout=p_drop, bert_attention_dropout=p_drop)
    m.train().requires_grad_(True).to(DEV)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    gen = torch.Generator().manual_seed(9)
    ids = torch.randint(3, V, (B, L), generator=gen)
    batch = {"input_ids": ids, "input_mask": torch.ones(B, L, dtype=torch.long), "correct_ids": ids.clone()}
    return m, diff, batch, torch.tensor([400, 1500][:B], device=DEV)


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_panel_layers_equal_the_op_per_node_tape(p_drop):
    """training.PANEL_LAYERS: three encoder layers (d_model 512, seq_len 512) forward and backward through mh_train_layer_fwd / _bwd against
    the tape of rounds 1 - 5.  Both run the same kernels' arithmetic on the same values (the layout changes addresses, not sums; the dropout
    masks come from the same Philox counters): losses and every parameter's gradient agree bit for bit."""
    from musediffusion_amd import training
    m, diff, batch, t = _model(3, 512, 2, p_drop)

    def run():
        m.zero_grad(set_to_none=True)
        m._dropout_calls = 0                 # the same forward-call number -> the same Philox counters in both runs
        with CpuDraws(11):
            terms = diff.training_losses(m, t, model_kwargs=batch)
        terms["loss"].mean().backward()
        return terms["loss"].detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    try:
        training.PANEL_LAYERS = True
        n0 = training.PANEL_LAYER_CALLS
        l1, g1 = run()
        assert training.PANEL_LAYER_CALLS == n0 + 3           # the panel node ran, once per layer
        training.PANEL_LAYERS = False
        l0, g0 = run()
        assert training.PANEL_LAYER_CALLS == n0 + 3           # ... and the other run took the op-per-node tape
    finally:
        training.PANEL_LAYERS = True
    assert torch.isfinite(l1).all()
    assert torch.equal(l0, l1), (l0, l1)
    assert g0.keys() == g1.keys()
    bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
    worst = {n: float((g0[n] - g1[n]).abs().max() / (g0[n].abs().max() + 1e-20)) for n in bad}
    assert not bad, worst


def test_get_logits_mode2_against_the_reference():
    """TransformerNetModel.get_logits with logits_mode 2 (network.py:94-104) against the reference's own output (tests/golden/logits_mode2.npz):
    fp32 GEMM + mh_distance_scores.  Tolerance 2e-5 abs away from zero distance (fp32 dot products in another order); where the
    position IS a table row the distance cancels to ~1e-7 before the square root, so those scores are held to sqrt-of-rounding size."""
    import numpy as np
    from conftest import load_golden
    from oracle import fixtures as fx
    g = load_golden("logits_mode2.npz")
    tag = str(g["tag"])
    c = fx.CONFIGS[tag]
    m = TransformerNetModel(c["E"], c["E"], c["Tt"], c["V"], c["L"], dropout=0.0, logits_mode=2, bert_hidden=c["H"], bert_layers=c["nL"],
                            bert_heads=c["nh"], bert_ffn=c["F"], compute_dtype="fp32")
    m.load_state_dict(fx.state_dict(tag))
    m.eval().requires_grad_(False).to(DEV)
    ref = torch.from_numpy(np.asarray(g["scores"]))
    got = m.get_logits(torch.from_numpy(np.asarray(g["hidden"])).to(DEV)).cpu()
    assert got.shape == ref.shape
    far = ref < -0.05
    assert float((got - ref)[far].abs().max()) < 2e-5
    assert float((got - ref)[~far].abs().max()) < 5e-3 and int((~far).sum()) >= 3
    assert torch.equal(got.argmax(-1), ref.argmax(-1))
    m.logits_mode = 3
    with pytest.raises(NotImplementedError):
        m.get_logits(torch.zeros(1, 2, c["E"], device=DEV))
