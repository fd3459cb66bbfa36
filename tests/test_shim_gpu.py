"""INTEGRATION.md section 1 executed on the GPU: the block of run/sample.py:185-220 written against the REFERENCE's import names
(tests/shim/MuseDiffusion = the four re-export modules) reproduces the reference's golden tokens."""
import os
import sys
from functools import partial
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402
from oracle import fixtures as fx  # noqa: E402
from oracle import sampling as osa  # noqa: E402

SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim")
DEV = "cuda"


@pytest.mark.parametrize("tag", ["tiny", "c1"])
def test_sampling_block_through_the_reference_import_names(tag):
    sys.path.insert(0, SHIM)
    try:
        from MuseDiffusion.models.rounding import denoised_fn_round                      # run/sample.py:45
        from MuseDiffusion.utils.initialization import create_model_and_diffusion         # run/sample.py:84
        import MuseDiffusion.models.network as shim_net
        assert os.path.dirname(shim_net.__file__).startswith(SHIM)
        c = fx.CONFIGS[tag]
        args = SimpleNamespace(hidden_dim=c["E"], hidden_t_dim=c["Tt"], vocab_size=c["V"], seq_len=c["L"], dropout=0.0,
                               noise_schedule="sqrt", diffusion_steps=2000, timestep_respacing="", rescale_timesteps=True,
                               predict_xstart=True, bert_hidden=c["H"], bert_layers=c["nL"], bert_heads=c["nh"], bert_ffn=c["F"])
        model, diffusion = create_model_and_diffusion(args)
        sd = fx.state_dict(tag)
        model.load_state_dict(sd)                                                         # run/sample.py:85
        model.eval().requires_grad_(False).to(DEV)                                        # run/sample.py:100
        model_emb = torch.nn.Embedding(num_embeddings=c["V"], embedding_dim=c["E"],
                                       _weight=model.word_embedding.weight.clone().cpu()).eval().requires_grad_(False)   # run/sample.py:93-98
        inp = fx.case_inputs(tag, sd["word_embedding.weight"])
        g = load_golden("model_%s.npz" % tag)
        B, L, E = c["B"], c["L"], c["E"]
        x_start = model.get_embeds(inp["batch"]["correct_ids"].to(DEV))                   # run/sample.py:185-186
        mask3 = torch.broadcast_to(inp["batch"]["input_mask"].to(DEV).unsqueeze(-1), x_start.shape)
        x_noised = torch.where(mask3 == 0, x_start, inp["gen_noise0"].to(DEV))            # run/sample.py:188-190
        torch.manual_seed(fx.loop_seed(tag, "ddim50"))
        z = torch.zeros(B, L, E)
        noises = [torch.randn_like(z) for _ in range(50)]                                 # the reference's draws (CPU generator)
        diffusion.noise_fn = lambda k, i, x: noises[k].to(DEV)
        samples = diffusion.ddim_sample_loop(model, (B, L, E), noise=x_noised, clip_denoised=True,
                                             denoised_fn=partial(denoised_fn_round, model_emb.to(DEV), dist=None), model_kwargs={},
                                             top_p=1, clamp_step=0, clamp_first=True, mask=mask3, x_start=x_start, gap=40,
                                             only_last=True)                              # run/sample.py:200-216 (step = 50 -> gap 40)
        sample = samples[-1]
        tokens = torch.argmax(model.get_logits(sample), dim=-1).cpu()                     # run/sample.py:218-220
        assert torch.equal(tokens, torch.from_numpy(g["loop_ddim50_tokens"]).long())
    finally:
        sys.path.remove(SHIM)
        for k in [k for k in sys.modules if k == "MuseDiffusion" or k.startswith("MuseDiffusion.")]:
            del sys.modules[k]
