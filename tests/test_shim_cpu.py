"""The recipe of INTEGRATION.md section 1, executed: tests/shim/MuseDiffusion/ holds exactly the re-export modules a maintainer would
write, and the hot path is imported THROUGH those names.  CPU part: the import surface and - where the reference
checkout exists (the build container) - the reference's REAL utils/initialization.py running over the shim: its import site
(utils/initialization.py:110-112) then binds to this package's classes."""
import importlib
import importlib.util
import os
import sys
from types import SimpleNamespace

import pytest

SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim")
REF_INIT = "/root/reference/MuseDiffusion/utils/initialization.py"


@pytest.fixture()
def shim_path():
    saved = {k: v for k, v in sys.modules.items() if k == "MuseDiffusion" or k.startswith("MuseDiffusion.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, SHIM)
    try:
        yield
    finally:
        sys.path.remove(SHIM)
        for k in [k for k in sys.modules if k == "MuseDiffusion" or k.startswith("MuseDiffusion.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_import_surface_through_the_reference_names(shim_path):
    import musediffusion_amd.models.diffusion as ours_d
    import musediffusion_amd.models.network as ours_n
    net = importlib.import_module("MuseDiffusion.models.network")
    dif = importlib.import_module("MuseDiffusion.models.diffusion")
    rnd = importlib.import_module("MuseDiffusion.models.rounding")
    stp = importlib.import_module("MuseDiffusion.models.step_sample")
    assert os.path.dirname(net.__file__).startswith(SHIM)
    assert net.TransformerNetModel is ours_n.TransformerNetModel
    # utils/initialization.py:110-112, utils/train_util.py (diffusion object), run/sample.py:45, run/train.py:22
    for name in ("GaussianDiffusion", "SpacedDiffusion", "space_timesteps", "get_named_beta_schedule", "unwrap_model", "_WrappedModel",
                 "_extract_into_tensor", "mean_flat", "betas_for_alpha_bar"):
        assert getattr(dif, name) is getattr(ours_d, name), name
    assert callable(rnd.denoised_fn_round) and callable(rnd.get_efficient_knn) and callable(rnd.get_knn)
    for name in ("create_named_schedule_sampler", "LossAwareSampler", "UniformSampler", "LossSecondMomentResampler", "ScheduleSampler", "FixSampler"):
        assert hasattr(stp, name), name
    # the three helpers training_losses_seq2seq calls on the diffusion object (diffusion.py:614, :629, :641)
    for name in ("_get_x_start", "_token_discrete_loss", "_x0_helper", "training_losses_seq2seq", "training_losses_seq2seq_with_corruption"):
        assert hasattr(dif.GaussianDiffusion, name), name
    # north_star alias modules
    assert importlib.import_module("MuseDiffusion.models.denoising_model").TransformerNetModel is ours_n.TransformerNetModel
    assert importlib.import_module("MuseDiffusion.models.gaussian_diffusion").GaussianDiffusion is ours_d.GaussianDiffusion
    assert importlib.import_module("MuseDiffusion.models.nn").timestep_embedding is ours_n.TransformerNetModel.timestep_embedding


@pytest.mark.skipif(not os.path.exists(REF_INIT), reason="the reference checkout exists in the build container only")
def test_the_reference_factory_file_runs_over_the_shim(shim_path):
    """The reference's own create_model_and_diffusion (loaded from its file, unmodified) builds THIS package's classes when the four
    model modules are the re-exports: same constructor keywords, same defaults (bert-base-uncased shape, network.py:44-46)."""
    import torch
    import musediffusion_amd.models.diffusion as ours_d
    import musediffusion_amd.models.network as ours_n
    spec = importlib.util.spec_from_file_location("_reference_initialization", REF_INIT)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mod)
    args = SimpleNamespace(hidden_dim=128, hidden_t_dim=128, vocab_size=729, seq_len=64, dropout=0.1, noise_schedule="sqrt",
                           diffusion_steps=2000, timestep_respacing="", rescale_timesteps=True, predict_xstart=True)
    model, diffusion = mod.create_model_and_diffusion(args)
    assert type(model) is ours_n.TransformerNetModel and type(diffusion) is ours_d.SpacedDiffusion
    assert model.hidden_size == 768 and len(model.input_transformers.layer) == 12 and model.num_heads == 12
    assert diffusion.num_timesteps == 2000 and diffusion.rescale_timesteps and diffusion.predict_xstart
    sd = model.state_dict()
    for k in ("word_embedding.weight", "lm_head.weight", "time_embed.0.weight", "input_up_proj.0.weight", "position_embeddings.weight",
              "input_transformers.layer.11.output.LayerNorm.bias", "output_down_proj.2.bias"):
        assert k in sd, k
    assert sd["lm_head.weight"].data_ptr() == sd["word_embedding.weight"].data_ptr()       # tied (network.py:56-58)
    assert isinstance(model, torch.nn.Module) and model.training


def test_train_side_names_of_run_train_py(shim_path):
    """run/train.py:22-27 imports: create_named_schedule_sampler, the five initialization helpers, TrainLoop - all reachable through
    the reference's module names over the shim, with the reference's call signatures (keyword-only TrainLoop constructor)."""
    import inspect
    tu = importlib.import_module("MuseDiffusion.utils.train_util")
    import musediffusion_amd.utils.initialization as ours_i
    import musediffusion_amd.utils.train_util as ours_t
    assert tu.TrainLoop is ours_t.TrainLoop and callable(tu.update_ema)
    for name in ("create_model_and_diffusion", "seed_all", "fetch_pretrained_embedding", "overload_embedding", "fetch_pretrained_denoiser",
                 "overload_denoiser", "get_latest_model_path"):
        assert callable(getattr(ours_i, name)), name
    assert list(inspect.signature(ours_i.overload_embedding).parameters) == ["model", "emb_weight", "freeze_embedding"]
    assert list(inspect.signature(ours_i.overload_denoiser).parameters) == ["model", "denoiser_state_dict"]
    kinds = {p.kind for n, p in inspect.signature(tu.TrainLoop.__init__).parameters.items() if n != "self"}
    assert kinds == {inspect.Parameter.KEYWORD_ONLY}
