"""Deferred LayerNorm (csrc/gemm.hip DeferArgs; HF BertSelfOutput / BertOutput LayerNorms reached from MuseDiffusion/models/
network.py:151): the attention-output / FFN-output GEMMs store raw rows + partial row statistics and their consumers normalise on
the fly.  The three forward modes of the bf16 engine (0 = LayerNorm epilogues / kernels, 1 = default, 2 = deferred everywhere) must
agree with each other within bf16 rounding and each with the fp32 oracle within the stated bf16 tolerance
(tests/test_diffusion_gpu.py::test_bf16_config2_shape_against_oracle); runs are bit-reproducible (no atomics)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from musediffusion_amd import _lib  # noqa: E402
from musediffusion_amd.models.network import TransformerNetModel  # noqa: E402
from oracle import denoiser as odn  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("H,nh,F,L", [(512, 8, 2048, 512), (768, 12, 3072, 512), (256, 4, 1024, 528)])
def test_defer_modes_agree_and_track_oracle(H, nh, F, L, dbg_lib):
    B, E, nL, V, Tt = 2, 128, 3, 729, 64
    sd = odn.random_state_dict(E, H, F, nL, V, L, Tt, seed=21, emb_std=0.5)
    _lib.check(_lib.lib().mh_denoiser_set_defer_ln(2))      # the engine packs the folded operands when mode 2 is on at construction
    try:
        m = TransformerNetModel(E, E, Tt, V, L, dropout=0.0, bert_hidden=H, bert_layers=nL, bert_heads=nh, bert_ffn=F, compute_dtype="bf16")
        m.load_state_dict(sd)
        m.eval().requires_grad_(False).to(DEV)
        m.engine()
    finally:
        _lib.lib().mh_denoiser_set_defer_ln(1)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, L, E, generator=gen)
    t = torch.tensor([900.0, 15.5])
    with torch.no_grad():
        ref = odn.forward(sd, x, t, nh)
    outs = {}
    try:
        for mode in (0, 1, 2):
            _lib.check(_lib.lib().mh_denoiser_set_defer_ln(mode))
            y = m(x.to(DEV), t.to(DEV)).cpu()
            y2 = m(x.to(DEV), t.to(DEV)).cpu()
            assert torch.equal(y, y2), "mode %d is not reproducible" % mode
            d = (y - ref).abs()
            print("H=%d mode %d: mean |delta| %.4f max %.4f" % (H, mode, float(d.mean()), float(d.max())))
            assert float(d.mean()) <= 0.02 and float(d.max()) <= 0.25
            outs[mode] = y
    finally:
        _lib.lib().mh_denoiser_set_defer_ln(1)
    assert float((outs[0] - outs[2]).abs().mean()) < 0.01
    assert not torch.equal(outs[0], outs[2])          # really two different kernel sequences


def test_defer_ln_on_trained_like_statistics_at_bert_base_width(dbg_lib):
    """Deferred LayerNorm is the default exactly at the reference's only real width (d_model 768): its variance is the one-pass
    E[x^2] - mean^2 of per-tile partial sums and the normalisation subtracts mean x c1 from W'x - terms that cancel badly when a row
    has a large mean or a few outlier channels, as trained BERT-style residual streams do.  Trained-like weights: 12 layers, LayerNorm
    gains in [0.5, 2] with a few at 6, shifts up to 0.5, attention / FFN output biases that put a large common offset (mean 3) and
    outlier channels (x 40) on the pre-LayerNorm rows.  Mode 1 (deferred) against mode 0 (LayerNorm kernels, two-pass statistics),
    same bf16 operands: tight; both against the fp32 oracle: the stated bf16 tolerance."""
    B, E, H, nh, F, nL, V, Tt, L = 2, 128, 768, 12, 3072, 12, 729, 64, 512
    sd = odn.random_state_dict(E, H, F, nL, V, L, Tt, seed=33, emb_std=0.5)
    g = torch.Generator().manual_seed(34)
    for l in range(nL):
        p = "input_transformers.layer.%d." % l
        for ln in ("attention.output.LayerNorm", "output.LayerNorm"):
            gain = 0.5 + 1.5 * torch.rand(H, generator=g)
            gain[torch.randint(0, H, (6,), generator=g)] = 6.0
            sd[p + ln + ".weight"] = gain
            sd[p + ln + ".bias"] = torch.rand(H, generator=g) - 0.5
        for dense in ("attention.output.dense", "output.dense"):
            b = 3.0 + 0.3 * torch.randn(H, generator=g)                    # large common offset of the residual stream
            b[torch.randint(0, H, (4,), generator=g)] = 40.0               # outlier channels
            sd[p + dense + ".bias"] = b
    m = TransformerNetModel(E, E, Tt, V, L, dropout=0.0, bert_hidden=H, bert_layers=nL, bert_heads=nh, bert_ffn=F, compute_dtype="bf16")
    m.load_state_dict(sd)
    m.eval().requires_grad_(False).to(DEV)
    x = torch.randn(B, L, E, generator=g)
    t = torch.tensor([900.0, 15.5])
    with torch.no_grad():
        ref = odn.forward(sd, x, t, nh)
    outs = {}
    try:
        for mode in (0, 1):
            _lib.check(_lib.lib().mh_denoiser_set_defer_ln(mode))
            outs[mode] = m(x.to(DEV), t.to(DEV)).cpu()
            d = (outs[mode] - ref).abs()
            print("trained-like H=768 mode %d vs oracle: mean |delta| %.4f max %.4f (ref absmax %.2f)" % (mode, float(d.mean()), float(d.max()), float(ref.abs().max())))
            assert float(d.mean()) <= 0.02 and float(d.max()) <= 0.25, mode
    finally:
        _lib.lib().mh_denoiser_set_defer_ln(1)
    dd = (outs[0] - outs[1]).abs()
    print("mode 1 vs mode 0: mean |delta| %.5f max %.4f" % (float(dd.mean()), float(dd.max())))
    assert float(dd.mean()) < 0.005 and float(dd.max()) < 0.1
