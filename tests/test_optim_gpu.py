"""Fused AdamW + EMA step against torch.optim.AdamW + the reference's update_ema formula run on CPU, and the
reference-format checkpoint round trip."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from musediffusion_amd import checkpoint  # noqa: E402
from musediffusion_amd.optim import FusedAdamWEMA  # noqa: E402

DEV = "cuda"


def test_fused_adamw_ema_matches_torch():
    g = torch.Generator().manual_seed(0)
    shapes = [(70000,), (33, 65), (7,), (128, 512)]
    cpu_params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    dev_params = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in cpu_params]
    rates = (0.5, 0.9, 0.99)
    ref_opt = torch.optim.AdamW(cpu_params, lr=1e-3, weight_decay=0.01)
    ref_ema = [[p.detach().clone() for p in cpu_params] for _ in rates]
    opt = FusedAdamWEMA(dev_params, lr=1e-3, weight_decay=0.01, ema_rates=rates)
    for step in range(5):
        grads = [torch.randn(*s, generator=g) * (1 + step) for s in shapes]
        for p, q, gr in zip(cpu_params, dev_params, grads):
            p.grad = gr.clone()
            q.grad = gr.clone().to(DEV)
        norm_ref = float(torch.sqrt(sum((gr ** 2).sum() for gr in grads)))
        assert abs(float(opt.grad_norm()) - norm_ref) < 1e-3 * norm_ref
        lr = 1e-3 * (1 - step / 10)                       # linear anneal like _anneal_lr (train_util.py:266-272)
        for gp in ref_opt.param_groups:
            gp["lr"] = lr
        ref_opt.step()
        for rate, copies in zip(rates, ref_ema):
            for targ, src in zip(copies, cpu_params):
                targ.detach().mul_(rate).add_(src.detach(), alpha=1 - rate)
        v0 = dev_params[0]._version
        opt.step(lr=lr)
        assert dev_params[0]._version > v0
        for p, q in zip(cpu_params, dev_params):
            assert torch.allclose(q.detach().cpu(), p.detach(), rtol=2e-6, atol=2e-7)
        for e in range(3):
            for a, b in zip(ref_ema[e], opt.ema[e]):
                assert torch.allclose(b.cpu(), a, rtol=2e-6, atol=2e-7)
    sd, ref_sd = opt.state_dict(), ref_opt.state_dict()
    assert set(sd) == set(ref_sd) and set(sd["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"}
    assert torch.allclose(sd["state"][3]["exp_avg_sq"].cpu(), ref_sd["state"][3]["exp_avg_sq"], rtol=1e-5, atol=1e-8)


def test_checkpoint_round_trip_reference_format(tmp_path):
    from musediffusion_amd.models.network import TransformerNetModel
    from oracle import fixtures as fx
    c = fx.CONFIGS["tiny"]
    mk = lambda: TransformerNetModel(c["E"], c["E"], c["Tt"], c["V"], c["L"], bert_hidden=c["H"], bert_layers=c["nL"],
                                     bert_heads=c["nh"], bert_ffn=c["F"]).to(DEV)
    m = mk()
    opt = FusedAdamWEMA(m.parameters(), lr=1e-4, ema_rates=(0.5, 0.99))
    for p in m.parameters():
        p.grad = torch.randn_like(p) * 0.01
    opt.step()
    checkpoint.save(str(tmp_path), 1234, m, opt, ema_rates=("0.5", "0.99"))
    names = sorted(os.listdir(tmp_path))
    assert names == ["ema_0.5_001234.pt", "ema_0.99_001234.pt", "model_001234.pt", "opt_001234.pt"]
    sd = torch.load(os.path.join(tmp_path, "model_001234.pt"))
    assert set(sd) == set(fx.state_dict("tiny"))            # the reference's key names
    assert checkpoint.parse_resume_step_from_filename(checkpoint.find_resume_checkpoint(str(tmp_path))) == 1234
    m2 = mk()
    opt2 = FusedAdamWEMA(m2.parameters(), lr=1e-4, ema_rates=(0.5, 0.99))
    assert checkpoint.resume(str(tmp_path), m2, opt2, ema_rates=("0.5", "0.99")) == 1234
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)
    assert opt2.step_count == 1 and torch.equal(opt2.exp_avg[5], opt.exp_avg[5]) and torch.equal(opt2.ema[1][3], opt.ema[1][3])
    assert checkpoint.resume(str(tmp_path / "nope"), m2) == 0


def test_resume_from_a_checkpoint_the_reference_wrote():
    """Resume from tests/golden/ref_ckpt/ (written by the reference's own model class + torch.optim.AdamW, tools/make_golden.py):
    parameters, EMA copy and Adam moments arrive bit for bit, and the forward on the loaded weights equals the reference's recorded
    forward (fp32 mode, 1e-4)."""
    import numpy as np
    from conftest import GOLDEN
    from musediffusion_amd.models.network import TransformerNetModel
    d = os.path.join(GOLDEN, "ref_ckpt")
    f = np.load(os.path.join(d, "forward.npz"))
    c = {k[4:]: int(f[k]) for k in f.files if k.startswith("cfg_")}
    m = TransformerNetModel(c["E"], c["E"], c["Tt"], c["V"], c["L"], bert_hidden=c["H"], bert_layers=c["nL"], bert_heads=c["nh"],
                            bert_ffn=c["F"], compute_dtype="fp32").to(DEV)
    opt = FusedAdamWEMA(m.parameters(), lr=3e-4, ema_rates=(0.9999,))
    assert checkpoint.resume(d, m, opt, ema_rates=("0.9999",)) == 7
    sd = torch.load(os.path.join(d, "model_000007.pt"), map_location="cpu")
    for n, p in m.named_parameters():
        assert torch.equal(p.detach().cpu(), sd[n]), n
    esd = torch.load(os.path.join(d, "ema_0.9999_000007.pt"), map_location="cpu")
    for j, (n, _) in enumerate(m.named_parameters()):
        assert torch.equal(opt.ema[0][j].cpu(), esd[n]), n
    osd = torch.load(os.path.join(d, "opt_000007.pt"), map_location="cpu")
    assert opt.step_count == 1 and abs(opt.lr - 1e-4) < 1e-12
    for i in (0, 3, len(opt.params) - 1):
        assert torch.equal(opt.exp_avg[i].cpu(), osd["state"][i]["exp_avg"]) and torch.equal(opt.exp_avg_sq[i].cpu(), osd["state"][i]["exp_avg_sq"])
    m.eval().requires_grad_(False)
    y = m(torch.from_numpy(f["x"]).to(DEV), torch.from_numpy(f["t"]).to(DEV)).cpu().numpy()
    assert float(np.abs(y - f["y"]).max()) < 1e-4


def test_parameters_without_a_gradient_are_skipped_like_torch_adamw():
    """torch.optim.AdamW (train_util.py:95) skips `p.grad is None`; update_ema (:21-31) does not.  A frozen tensor (requires_grad
    False) and one the backward never reached: values untouched, no optimizer state, nothing in the norm, the clip leaves them, their
    EMA copies still take the EMA step; a gradient at an address that is not 16-byte aligned is served too."""
    g = torch.Generator().manual_seed(3)
    shapes = [(300, 7), (70001,), (64, 64), (5,)]
    cpu = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    dev = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in cpu]
    cpu[0].requires_grad_(False); dev[0].requires_grad_(False)
    rates = (0.9,)
    ref = torch.optim.AdamW(cpu, lr=1e-3, weight_decay=0.01)
    ref_ema = [p.detach().clone() for p in cpu]
    opt = FusedAdamWEMA(dev, lr=1e-3, weight_decay=0.01, ema_rates=rates)
    with pytest.raises(RuntimeError):
        opt.step()                                         # nothing has a gradient yet
    for step in range(3):
        for i in (1, 2):                                   # tensor 3 never gets one, tensor 0 is frozen
            gr = torch.randn(*shapes[i], generator=g)
            cpu[i].grad = gr.clone()
            if i == 1:                                     # a view 4 bytes into a larger buffer: not 16-byte aligned
                buf = torch.empty(gr.numel() + 1, device=DEV)
                buf[1:].copy_(gr)
                dev[i].grad = buf[1:]
                assert dev[i].grad.data_ptr() % 16 != 0
            else:
                dev[i].grad = gr.clone().to(DEV)
        norm_ref = float(torch.sqrt(sum((cpu[i].grad ** 2).sum() for i in (1, 2))))
        assert abs(float(opt.clip_grad_norm(0.5)) - norm_ref) < 1e-3 * norm_ref
        torch.nn.utils.clip_grad_norm_(cpu, 0.5)
        v_frozen, v_live = dev[0]._version, dev[1]._version
        ref.step()
        for t, s in zip(ref_ema, cpu):
            t.mul_(rates[0]).add_(s.detach(), alpha=1 - rates[0])
        opt.step()
        assert dev[0]._version == v_frozen and dev[1]._version > v_live
        for p, q in zip(cpu, dev):
            assert torch.allclose(q.detach().cpu(), p.detach(), rtol=2e-6, atol=2e-7)
        for a, b in zip(ref_ema, opt.ema[0]):
            assert torch.allclose(b.cpu(), a, rtol=2e-6, atol=2e-7)
    assert torch.equal(dev[0].detach().cpu(), cpu[0].detach()) and torch.equal(dev[3].detach().cpu(), cpu[3].detach())
    assert sorted(opt.state_dict()["state"]) == sorted(ref.state_dict()["state"]) == [1, 2]


def test_overload_embedding_freeze_and_one_step_match_the_reference():
    """tests/golden/overload_freeze_tiny.npz (tools/make_golden.py overload): the reference's model after overload_embedding with
    freeze_embedding (utils/initialization.py:54-68), its training_losses, backward, one torch.optim.AdamW step and update_ema.
    Here: the same helpers, TrainStep over the fused optimizer.  The embedding stays put (and out of the optimizer state), the
    UNTIED lm_head.weight trains, EMA copies of every parameter - the frozen one included - match; the EMA checkpoint keeps
    the two tensors apart."""
    from conftest import load_golden
    from musediffusion_amd.models.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps
    from musediffusion_amd.models.network import TransformerNetModel
    from musediffusion_amd.train_step import TrainStep
    from musediffusion_amd.utils.initialization import overload_embedding
    from oracle import fixtures as fx
    from test_training_gpu import CpuDraws, close
    g = load_golden("overload_freeze_tiny.npz")
    tag = "tiny"
    c = fx.CONFIGS[tag]
    m = TransformerNetModel(c["E"], c["E"], c["Tt"], c["V"], c["L"], dropout=0.0, bert_hidden=c["H"], bert_layers=c["nL"], bert_heads=c["nh"],
                            bert_ffn=c["F"], compute_dtype="fp32", bert_hidden_dropout=0.0, bert_attention_dropout=0.0)
    m.load_state_dict(fx.state_dict(tag))
    overload_embedding(m, torch.from_numpy(g["emb"]).clone(), True)
    m.train().to(DEV)
    assert [n for n, _ in m.named_parameters()] == list(g["param_names"])
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    li = fx.loss_inputs(tag)
    t_fix, w_fix = li["t"], li["w"]

    class FixedSampler:
        def sample(self, n, device):
            return t_fix.to(device), w_fix.to(device)
    lr, wd, rate = float(g["lr"]), float(g["wd"]), float(g["ema_rate"])
    ts = TrainStep(m, diff, microbatch=-1, lr=lr, weight_decay=wd, ema_rate=rate, schedule_sampler=FixedSampler())
    with CpuDraws(fx.loss_seed(tag)):
        losses = ts.forward_backward(dict(li["batch"]))
    for k in ("mse", "nll", "loss"):
        close(k, losses[k], float((torch.from_numpy(g[k]) * w_fix).mean()), 5e-4)
    by = dict(m.named_parameters())
    watch = {"lmw": "lm_head.weight", "lmb": "lm_head.bias", "q0": "input_transformers.layer.0.attention.self.query.weight",
             "te0": "time_embed.0.weight", "ff2": "input_transformers.layer.0.output.dense.weight", "pos": "position_embeddings.weight"}
    assert m.word_embedding.weight.grad is None
    for k, n in watch.items():
        close("grad " + n, by[n].grad, g["g_" + k], 2e-3)
    ts.optimize()
    names = list(g["param_names"])
    for k, n in watch.items():
        # one AdamW step moves a weight by ~lr whatever the gradient's size: compare the MOVE, within 2 % of lr (sign(g) is what
        # the first step applies; elements whose reference gradient is ~0 are the ones a 2e-3 gradient error can flip)
        p1, ref = by[n].detach().cpu(), torch.from_numpy(g["p_" + k])
        big = torch.from_numpy(np.abs(g["g_" + k]) > 0.05 * np.abs(g["g_" + k]).max())
        assert float((p1 - ref)[big].abs().max()) < 0.02 * lr, n
        e1, eref = ts.opt.ema[0][names.index(n)].cpu(), torch.from_numpy(g["ema_" + k])
        assert float((e1 - eref)[big].abs().max()) < 0.02 * lr, n
    wi = int(g["word_index"])
    assert torch.equal(m.word_embedding.weight.cpu(), torch.from_numpy(g["p_word"]))
    # update_ema of an unchanged tensor: e * rate + p * (1 - rate) to the last bit or two (torch's CPU add_ with alpha may fuse the
    # multiply-add; the kernel keeps them apart) - and NOT the optimizer's business otherwise: no Adam arithmetic touched it
    assert torch.allclose(ts.opt.ema[0][wi].cpu(), torch.from_numpy(g["ema_word"]), rtol=2e-6, atol=2e-7)
    assert sorted(ts.opt.state_dict()["state"]) == list(g["opt_state_keys"])
    esd = ts.opt.ema_state_dict(0, m)
    assert esd["lm_head.weight"].data_ptr() != esd["word_embedding.weight"].data_ptr()
    assert torch.equal(esd["lm_head.weight"], ts.opt.ema[0][names.index("lm_head.weight")])


def test_train_loop_with_the_fused_optimizer_saves_and_resumes(tmp_path):
    """utils/train_util.TrainLoop (the reference's constructor) over the real training path: three optimizer steps with the fused AdamW +
    EMA kernel, a checkpoint in the reference's file layout at step 2 and at the end, and a second TrainLoop on the same directory that
    resumes model, optimizer moments, step count and the EMA copy bit for bit."""
    import itertools
    from musediffusion_amd import synthetic
    from musediffusion_amd.models.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps
    from musediffusion_amd.models.network import TransformerNetModel
    from musediffusion_amd.utils.train_util import TrainLoop
    torch.manual_seed(0)
    np.random.seed(0)
    mk = lambda: TransformerNetModel(32, 32, 32, 729, 64, dropout=0.1, bert_hidden=64, bert_layers=2, bert_heads=2, bert_ffn=128,
                                     compute_dtype="fp32").train().to(DEV)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    batch = synthetic.training_batch(4, 64, seed=3)
    data = (dict(batch) for _ in itertools.count())
    logs = []
    kw = dict(diffusion=diff, data=data, batch_size=4, microbatch=2, lr=1e-3, ema_rate="0.9", log_interval=1, save_interval=3,
              resume_checkpoint="", weight_decay=0.01, checkpoint_path=str(tmp_path), gradient_clipping=1.0, log_fn=logs.append)
    m = mk()
    loop = TrainLoop(model=m, learning_steps=4, **kw)
    loop.run_loop()
    # steps 0 .. 3; the save at step 3 happens BEFORE that iteration's `step += 1`, i.e. after 4 optimizer steps; (4 - 1) % 3 == 0: no closing save
    assert loop.step == 4 and sorted(os.listdir(tmp_path)) == ["ema_0.9_000003.pt", "model_000003.pt", "opt_000003.pt"]
    assert all(np.isfinite(d["loss"]) and np.isfinite(d["grad_norm"]) for d in logs if "loss" in d) and logs[-1]["step"] == 3
    m2 = mk()
    loop2 = TrainLoop(model=m2, learning_steps=6, **kw)
    assert loop2.resume_step == 3 and loop2.opt.step_count == loop.opt.step_count == 4
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)
    for a, b in zip(loop.opt.exp_avg_sq, loop2.opt.exp_avg_sq):
        assert torch.equal(a, b)
    for a, b in zip(loop.opt.ema[0], loop2.opt.ema[0]):
        assert torch.equal(a, b)
    loop2.run_loop()
    assert loop2.step == 3 and "model_000006.pt" in os.listdir(tmp_path)      # steps 3 .. 5 of 6, then the closing save ((3 - 1) % 3 != 0)
