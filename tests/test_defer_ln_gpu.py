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
def test_defer_modes_agree_and_track_oracle(H, nh, F, L):
    B, E, nL, V, Tt = 2, 128, 3, 729, 64
    sd = odn.random_state_dict(E, H, F, nL, V, L, Tt, seed=21, emb_std=0.5)
    m = TransformerNetModel(E, E, Tt, V, L, dropout=0.0, bert_hidden=H, bert_layers=nL, bert_heads=nh, bert_ffn=F, compute_dtype="bf16")
    m.load_state_dict(sd)
    m.eval().requires_grad_(False).to(DEV)
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
