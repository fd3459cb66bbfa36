"""CPU tests of the product's host logic (no kernels run): float64 tables, respacing and samplers
against the reference fixtures; state_dict layout; the C-ABI library loads and exports every symbol
of include/musehip.h; the product path refuses CPU tensors instead of falling back."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden

from musediffusion_amd.models import diffusion as D
from musediffusion_amd.models import step_sample as S
from oracle import fixtures as fx

TABLES = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next",
          "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
          "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
          "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")


def test_tables_match_reference_bit_for_bit():
    g = load_golden("schedules.npz")
    d = D.GaussianDiffusion(betas=D.get_named_beta_schedule("sqrt", 2000), predict_xstart=True)
    for name in TABLES:
        np.testing.assert_array_equal(getattr(d, name)[g["sqrt2000_idx"]], g["sqrt2000_" + name], err_msg=name)
    for sched in ("linear", "cosine", "sqrt", "trunc_cos", "trunc_lin", "pw_lin"):
        b = D.get_named_beta_schedule(sched, 50)
        np.testing.assert_array_equal(b, g["betas50_" + sched])
        gd = D.GaussianDiffusion(betas=b, predict_xstart=True)
        np.testing.assert_array_equal(gd.posterior_mean_coef1, g["pmc1_50_" + sched])
        np.testing.assert_array_equal(gd.posterior_log_variance_clipped, g["plvc_50_" + sched])
    with pytest.raises(NotImplementedError):
        D.get_named_beta_schedule("nope", 10)
    with pytest.raises(AssertionError):
        D.GaussianDiffusion(betas=np.array([[0.1]]), predict_xstart=True)


def test_respacing_matches_reference():
    g = load_golden("schedules.npz")
    for key, (T, spec) in {"ddim50": (2000, "ddim50"), "sec": (300, "10,15,20"), "full": (2000, [2000]),
                           "odd": (100, [7, 3])}.items():
        use = D.space_timesteps(T, spec)
        np.testing.assert_array_equal(np.array(sorted(use)), g["space_" + key])
        sp = D.SpacedDiffusion(use_timesteps=use, betas=D.get_named_beta_schedule("sqrt", T),
                               rescale_timesteps=True, predict_xstart=True)
        np.testing.assert_array_equal(sp.betas, g["spaced_betas_" + key])
        np.testing.assert_array_equal(np.array(sp.timestep_map), g["spaced_map_" + key])
        assert sp.num_timesteps == len(use) and sp.original_num_steps == T
    with pytest.raises(ValueError):
        D.space_timesteps(10, [11])
    with pytest.raises(ValueError):
        D.space_timesteps(2000, "ddim1999")


def test_schedule_samplers_match_reference():
    from types import SimpleNamespace
    g = load_golden("schedules.npz")
    fake = SimpleNamespace(num_timesteps=6)
    rs = S.LossSecondMomentResampler(fake, history_per_term=3)
    ts, ls = g["lsr_ts"].tolist(), g["lsr_ls"].tolist()
    rs.update_with_all_losses(ts[:10], ls[:10])
    np.testing.assert_array_equal(rs.weights(), g["lsr_w_cold"])
    rs.update_with_all_losses(ts[10:], ls[10:])
    np.testing.assert_array_equal(rs.weights(), g["lsr_w_warm"])
    np.random.seed(11)
    idx, w = rs.sample(16, "cpu")
    np.testing.assert_array_equal(idx.numpy(), g["lsr_sample_idx"])
    np.testing.assert_array_equal(w.numpy(), g["lsr_sample_w"])
    np.random.seed(11)
    idx, w = S.UniformSampler(fake).sample(16, "cpu")
    np.testing.assert_array_equal(idx.numpy(), g["uni_sample_idx"])
    np.testing.assert_array_equal(w.numpy(), g["uni_sample_w"])
    np.testing.assert_array_equal(S.FixSampler(SimpleNamespace(num_timesteps=10)).weights(), g["fix_w"])
    with pytest.raises(NotImplementedError):
        S.create_named_schedule_sampler("nope", fake)
    with pytest.raises(RuntimeError):
        S.create_named_schedule_sampler("lossaware", fake)   # needs an initialised process group
    assert isinstance(S.create_named_schedule_sampler("uniform", fake), S.UniformSampler)
    # single-process update path (no process group): same state as update_with_all_losses
    rs2 = S.LossSecondMomentResampler(fake, history_per_term=3)
    rs2.update_with_local_losses(torch.tensor(ts), torch.tensor(ls, dtype=torch.float64))
    np.testing.assert_array_equal(rs2.weights(), g["lsr_w_warm"])


def test_state_dict_layout_is_the_reference_checkpoint_format():
    from musediffusion_amd.models.network import TransformerNetModel
    for tag in ("tiny", "same"):
        c = fx.CONFIGS[tag]
        m = TransformerNetModel(c["E"], c["E"], c["Tt"], c["V"], c["L"], dropout=0.0, bert_hidden=c["H"],
                                bert_layers=c["nL"], bert_heads=c["nh"], bert_ffn=c["F"])
        ref_sd = fx.state_dict(tag)      # key names / shapes recorded from the reference
        sd = m.state_dict()
        assert set(sd) == set(ref_sd), set(sd) ^ set(ref_sd)
        for k in sd:
            assert tuple(sd[k].shape) == tuple(ref_sd[k].shape), k
        assert m.lm_head.weight is m.word_embedding.weight        # tied (network.py:56-58)
        m.load_state_dict(ref_sd)                                  # strict load of a reference checkpoint
        g = load_golden("model_%s.npz" % tag)
        for k, v in m.state_dict().items():
            np.testing.assert_array_equal(v.numpy(), g["sd." + k])
    with pytest.raises(ValueError):
        TransformerNetModel(32, 16, 32, 729, 16)


def test_alias_modules_and_factory():
    from types import SimpleNamespace
    from musediffusion_amd.models import denoising_model, gaussian_diffusion, nn as mnn
    from musediffusion_amd.models.network import TransformerNetModel
    from musediffusion_amd.utils.initialization import create_model_and_diffusion
    assert denoising_model.TransformerNetModel is TransformerNetModel
    assert gaussian_diffusion.GaussianDiffusion is D.GaussianDiffusion
    assert mnn.timestep_embedding is TransformerNetModel.timestep_embedding
    args = SimpleNamespace(hidden_dim=32, hidden_t_dim=32, vocab_size=729, seq_len=16, dropout=0.1,
                           noise_schedule="sqrt", diffusion_steps=2000, timestep_respacing="",
                           rescale_timesteps=True, predict_xstart=True)
    model, diff = create_model_and_diffusion(args, bert_hidden=64, bert_layers=2, bert_heads=4, bert_ffn=256)
    assert isinstance(diff, D.SpacedDiffusion) and diff.num_timesteps == 2000
    assert diff.timestep_map == list(range(2000)) and diff.rescale_timesteps
    assert model.hidden_size == 64 and len(model.input_transformers.layer) == 2
    default = create_model_and_diffusion(SimpleNamespace(**{**vars(args), "seq_len": 8}))[0]
    assert default.hidden_size == 768 and len(default.input_transformers.layer) == 12   # bert-base like network.py:44


def _exports(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("mh_")}


def test_library_exports_every_declared_symbol():
    """include/musehip.h <-> libmusehip.so, include/musehip_dbg.h <-> libmusehip_dbg.so only.  The production library carries no
    process-global switch: nothing named mh_*_set_* (A/B variants, timing-only ablations, staggers, diagnostics) is exported by it."""
    from musediffusion_amd import _lib
    proto = lambda name: set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", open(os.path.join(REPO, "include", name)).read()))
    declared, dbg_declared = proto("musehip.h"), proto("musehip_dbg.h")
    assert declared and dbg_declared and not (declared & dbg_declared)
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert dbg_declared == set(_lib.DBG_SIGNATURES), dbg_declared ^ set(_lib.DBG_SIGNATURES)
    handle = _lib.lib()
    for name in declared:
        assert hasattr(handle, name), name
    assert handle.mh_abi_version() == 1
    prod, dbg = _exports(_lib.LIB_PATH), _exports(_lib.DBG_LIB_PATH)
    assert prod == declared, prod ^ declared
    assert dbg == declared | dbg_declared, dbg ^ (declared | dbg_declared)
    deny = re.compile(r"^mh_.*_set_|skip|debug|ablation|stagger|spread|plain_stores")
    assert not [n for n in prod if deny.search(n)]
    with _lib.debug_library() as d:
        for name in dbg_declared:
            assert hasattr(d, name), name
    assert _lib.lib() is handle


def test_product_path_refuses_cpu_instead_of_falling_back():
    from musediffusion_amd import _lib
    from musediffusion_amd.models.network import TransformerNetModel
    from musediffusion_amd.models.rounding import denoised_fn_round
    c = fx.CONFIGS["tiny"]
    m = TransformerNetModel(c["E"], c["E"], c["Tt"], c["V"], c["L"], bert_hidden=c["H"], bert_layers=c["nL"],
                            bert_heads=c["nh"], bert_ffn=c["F"]).eval().requires_grad_(False)
    x = torch.zeros(2, c["L"], c["E"])
    with pytest.raises(_lib.MuseHipError):
        m(x, torch.zeros(2))
    with pytest.raises(_lib.MuseHipError):
        m.get_embeds(torch.zeros(2, 4, dtype=torch.long))
    d = D.GaussianDiffusion(betas=D.get_named_beta_schedule("sqrt", 50), predict_xstart=True)
    with pytest.raises(_lib.MuseHipError):
        d.q_sample(x, torch.zeros(2, dtype=torch.long))
    with pytest.raises(_lib.MuseHipError):
        denoised_fn_round(torch.nn.Embedding(8, 4), torch.zeros(3, 4), None)
    # nothing under the product package imports the oracle
    import musediffusion_amd
    root = os.path.dirname(musediffusion_amd.__file__)
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dp, f)


def test_batch_and_metric_modules_refuse_cpu_tensors():
    """the rows next to the path have no CPU fallback either: host tensors raise instead of silently running elsewhere"""
    from musediffusion_amd import data as mdata, metric as mm
    from musediffusion_amd._lib import MuseHipError
    from musediffusion_amd.utils import decode_util as md
    v, off = torch.zeros(20, dtype=torch.int32), torch.tensor([0, 20])
    for fn in (lambda: mdata.masking_token(v, off), lambda: mdata.masking_note(v, off), lambda: mdata.randomize_note(v, off),
               lambda: mdata.random_rotating(v, off), lambda: mdata.collate_batches({"input_ids": v}, off, 32),
               lambda: mm.get_vectors(torch.zeros(2, 16, dtype=torch.int32)), lambda: md.validate_tokens(torch.zeros(2, 16, dtype=torch.int32))):
        with pytest.raises(MuseHipError):
            fn()
    c = mdata.Corruptions.from_config("mt,mn", 1, 0.5, "{'p': 0.25}")
    assert c.corr_kwargs == {"p": 0.25} and c.corr_max == 1
    with pytest.raises(AssertionError):
        mdata.Corruptions(("zz",), 1, 0.5)


def test_reference_written_checkpoint_loads_on_cpu():
    """tests/golden/ref_ckpt/ was written by the REFERENCE's model class and torch.optim.AdamW with the reference's own save calls
    (tools/make_golden.py gen_reference_checkpoint; utils/train_util.py:294-319).  Its keys / shapes must load strictly."""
    import os
    from conftest import GOLDEN
    from musediffusion_amd import checkpoint
    from musediffusion_amd.models.network import TransformerNetModel
    d = os.path.join(GOLDEN, "ref_ckpt")
    f = np.load(os.path.join(d, "forward.npz"))
    c = {k[4:]: int(f[k]) for k in f.files if k.startswith("cfg_")}
    m = TransformerNetModel(c["E"], c["E"], c["Tt"], c["V"], c["L"], bert_hidden=c["H"], bert_layers=c["nL"], bert_heads=c["nh"], bert_ffn=c["F"])
    main = checkpoint.find_resume_checkpoint(d)
    assert os.path.basename(main) == "model_000007.pt" and checkpoint.parse_resume_step_from_filename(main) == 7
    sd = torch.load(main, map_location="cpu")
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    assert set(sd) == set(m.state_dict())
    assert checkpoint.find_ema_checkpoint(main, 7, "0.9999").endswith("ema_0.9999_000007.pt")
    osd = torch.load(os.path.join(d, "opt_000007.pt"), map_location="cpu")
    assert len(osd["state"]) == len(list(m.parameters())) and set(osd["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"}


def test_pretrained_weight_helpers_of_the_reference(tmp_path):
    """utils/initialization.py:29-105 as run/train.py:93-100 and scripts/run_train.sh use them: fetch / overload the embedding (the
    Parameter is replaced, so lm_head keeps the old tensor; freeze takes only the new embedding out of training), fetch / overload
    a denoiser state_dict (only keys the model has), newest .pt of the newest sub-directory."""
    import argparse
    import os
    import time
    from types import SimpleNamespace

    from musediffusion_amd.models.network import TransformerNetModel
    from musediffusion_amd.utils import initialization as ini
    m = TransformerNetModel(16, 16, 16, 40, 8, bert_hidden=32, bert_layers=1, bert_heads=2, bert_ffn=64)
    emb = torch.nn.Embedding(40, 16)
    torch.save(emb.state_dict(), tmp_path / "emb.pt")
    args = SimpleNamespace(pretrained_embedding=str(tmp_path / "emb.pt"), hidden_dim=24, freeze_embedding=True, pretrained_denoiser="")
    with pytest.warns(UserWarning):
        w = ini.fetch_pretrained_embedding(args)
    assert args.hidden_dim == 16 and torch.equal(w, emb.weight)        # the file's width overwrites the config's
    with pytest.raises(argparse.ArgumentTypeError):
        ini.fetch_pretrained_embedding(SimpleNamespace(pretrained_embedding="", hidden_dim=16, freeze_embedding=True))
    assert ini.fetch_pretrained_embedding(SimpleNamespace(pretrained_embedding="", hidden_dim=16, freeze_embedding=False)) is None
    old = m.word_embedding.weight
    assert ini.overload_embedding(m, w, True) is m
    assert torch.equal(m.word_embedding.weight, emb.weight) and not m.word_embedding.weight.requires_grad
    assert m.lm_head.weight is old and old.requires_grad               # the head keeps the OLD tensor and keeps training
    assert {"word_embedding.weight", "lm_head.weight"} <= {n for n, _ in m.named_parameters()}
    with pytest.raises(AssertionError):
        ini.overload_embedding(m, torch.zeros(41, 16), False)
    assert ini.fetch_pretrained_denoiser(args) is None
    sd = {"time_embed.0.bias": torch.full((64,), 0.5), "not.a.key": torch.zeros(3)}
    torch.save(sd, tmp_path / "den.pt")
    got = ini.fetch_pretrained_denoiser(SimpleNamespace(pretrained_denoiser=str(tmp_path / "den.pt")))
    before = m.time_embed[2].bias.detach().clone()
    assert ini.overload_denoiser(m, got) is m
    assert float(m.time_embed[0].bias.min()) == 0.5 and torch.equal(m.time_embed[2].bias, before)
    # get_latest_model_path: newest sub-directory, newest .pt in it
    assert ini.get_latest_model_path(str(tmp_path / "none")) is None
    base = tmp_path / "runs"
    for i, d in enumerate(("a", "b")):
        os.makedirs(base / d)
        for j, f in enumerate(("model_000001.pt", "model_000002.pt", "notes.txt")):
            (base / d / f).write_text("x")
            os.utime(base / d / f, (time.time() + 10 * i + j, time.time() + 10 * i + j))
        os.utime(base / d, (time.time() + 100 * i, time.time() + 100 * i))
    assert ini.get_latest_model_path(str(base)) == str(base / "b" / "model_000002.pt")
    os.makedirs(base / "c")
    os.utime(base / "c", (time.time() + 1000, time.time() + 1000))
    assert ini.get_latest_model_path(str(base)) is None                # the newest directory holds no checkpoint


def test_compute_dtype_names():
    """the four compute modes by name / torch dtype / code; anything else is refused (no silent default)"""
    from musediffusion_amd import _lib, ops
    assert [ops.dtype_code(n) for n in ("fp32", "bf16", "bf16x3", "f16x3")] == [_lib.MH_F32, _lib.MH_BF16, _lib.MH_BF16X3, _lib.MH_F16X3]
    assert ops.dtype_code(torch.float32) == _lib.MH_F32 and ops.dtype_code(torch.bfloat16) == _lib.MH_BF16 and ops.dtype_code(_lib.MH_F16X3) == _lib.MH_F16X3
    assert ops.SPLIT_DTYPES == (_lib.MH_BF16X3, _lib.MH_F16X3)
    for bad in ("fp16", "f16", 7, None, torch.float16):
        with pytest.raises(ValueError):
            ops.dtype_code(bad)


def test_ema_file_of_a_tied_model_loads_the_way_the_reference_loads_it():
    """An `ema_*.pt` file of a model whose head is tied to the embedding carries the EMA embedding under `word_embedding.weight` and the
    LIVE embedding under `lm_head.weight` (the reference's `_master_params_to_state_dict` walks named_parameters(), which lists the
    shared Parameter once: utils/train_util.py:321-333; optim.FusedAdamWEMA.ema_state_dict writes the same file).  The reference samples
    from such a file with a plain `model.load_state_dict(torch.load(path))` (run/sample.py), where the later key wins on the shared tensor:
    EMA transformer, live embedding / head.  A drop-in must give the same tensors - pinned here (decide explicitly): the
    product loads through nn.Module.load_state_dict and does NOT prefer either key."""
    import torch
    from musediffusion_amd.models.network import TransformerNetModel
    c = fx.CONFIGS["tiny"]
    mk = lambda: TransformerNetModel(c["E"], c["E"], c["Tt"], c["V"], c["L"], dropout=0.0, bert_hidden=c["H"], bert_layers=c["nL"],
                                     bert_heads=c["nh"], bert_ffn=c["F"])
    m = mk()
    sd = {k: v.clone() for k, v in fx.state_dict("tiny").items()}
    ema_emb, live_emb = sd["word_embedding.weight"] * 0.5, sd["word_embedding.weight"] + 1.0
    sd["word_embedding.weight"], sd["lm_head.weight"] = ema_emb, live_emb
    m.load_state_dict(sd)
    keys = list(m.state_dict())
    assert keys.index("word_embedding.weight") < keys.index("lm_head.weight")      # load order = key order: the head's entry is copied last
    assert m.lm_head.weight is m.word_embedding.weight and torch.equal(m.word_embedding.weight, live_emb)
    # what torch itself does with the same dict on an equally tied pair of stock modules (the reference's model is exactly that)
    emb, head = torch.nn.Embedding(c["V"], c["E"]), torch.nn.Linear(c["E"], c["V"])
    head.weight = emb.weight
    pair = torch.nn.ModuleDict({"word_embedding": emb, "lm_head": head})
    pair.load_state_dict({"word_embedding.weight": ema_emb, "lm_head.weight": live_emb, "lm_head.bias": sd["lm_head.bias"]})
    assert torch.equal(emb.weight, m.word_embedding.weight)
