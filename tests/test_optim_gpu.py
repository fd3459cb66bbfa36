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
