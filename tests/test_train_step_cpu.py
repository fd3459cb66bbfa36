"""Host logic of the training step contract (musediffusion_amd/train_step.py = utils/train_util.py:188-272) on CPU with
world-size-2 gloo: micro-batch accumulation sums gradients, DDP all-reduces only on the last micro-batch (`no_sync` before),
the loss-aware sampler is updated once per micro-batch, the learning rate anneals linearly.  The diffusion / model / optimizer
are small host stand-ins: the kernels behind the real ones need a GPU (tests/test_multirank_gpu.py)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import free_port
from musediffusion_amd.train_step import TrainStep


class ToyDiffusion:
    """training_losses with the reference's return contract: dict of [B] tensors, differentiable."""
    num_timesteps = 8

    def training_losses(self, model, t, model_kwargs, noise=None):
        x = model_kwargs["input_ids"].float()
        y = model(x).squeeze(-1)
        loss = (y - t.float()) ** 2
        return {"loss": loss, "mse": loss.detach(), "nll": loss.detach() * 0}


class HostOpt:
    def __init__(self, params):
        self.params, self.lrs = params, []

    def grad_norm(self):
        return torch.sqrt(sum((p.grad ** 2).sum() for p in self.params)).reshape(1)

    def step(self, lr=None):
        self.lrs.append(lr)


class CountingSampler:
    def __init__(self):
        self.calls = []

    def sample(self, n, device):
        self.calls.append(n)
        return torch.arange(n) % 8, torch.linspace(0.5, 1.5, n)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        torch.manual_seed(0)
        model = torch.nn.Linear(4, 1)
        ddp = DDP(model, broadcast_buffers=False, bucket_cap_mb=128, find_unused_parameters=False)
        sampler = CountingSampler()
        loop = TrainStep(model, ToyDiffusion(), microbatch=2, lr=1.0, learning_steps=10, schedule_sampler=sampler, ddp_model=ddp,
                         optimizer=HostOpt(list(model.parameters())))
        g = torch.Generator().manual_seed(10 + rank)
        cond = {"input_ids": torch.randn(5, 4, generator=g)}         # ragged: micro-batches of 2, 2, 1
        losses, gn = loop.run_step(cond)
        grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
        # expected: per rank the SUM over micro-batches of d/dθ (loss * w).mean(); DDP leaves the mean over ranks
        local = torch.zeros_like(grads)
        for i in range(0, 5, 2):
            m2 = torch.nn.Linear(4, 1)
            m2.load_state_dict(model.state_dict())
            x = cond["input_ids"][i:i + 2]
            t, w = torch.arange(x.shape[0]) % 8, torch.linspace(0.5, 1.5, x.shape[0])
            (((m2(x).squeeze(-1) - t.float()) ** 2) * w).mean().backward()
            local += torch.cat([p.grad.reshape(-1) for p in m2.parameters()])
        q.put((rank, grads.numpy(), local.numpy(), sampler.calls, loop.opt.lrs, loop.step, float(gn), sorted(losses)))
    finally:
        dist.destroy_process_group()


def test_accumulation_no_sync_and_anneal_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0, l0, calls0, lrs0, step0, gn0, keys0 = out[0]
    g1, l1, _, _, _, _, _ = out[1]
    np.testing.assert_allclose(g0, g1, rtol=0, atol=0)                 # identical after the all-reduce
    np.testing.assert_allclose(g0, (l0 + l1) / 2, rtol=1e-5, atol=1e-6)   # micro-batch SUM per rank, MEAN over ranks
    assert calls0 == [2, 2, 1]                                        # one sampler draw per micro-batch
    assert lrs0 == [1.0] and step0 == 1                               # step 0 of 10: lr * (1 - 0/10)
    assert keys0 == ["loss", "mse", "nll"]
    assert abs(gn0 - float(np.sqrt((g0 ** 2).sum()))) < 1e-5


def test_single_replica_eval_mode_restored_and_lr_anneal():
    torch.manual_seed(0)
    model = torch.nn.Linear(4, 1)
    loop = TrainStep(model, ToyDiffusion(), microbatch=-1, lr=2.0, learning_steps=4, schedule_sampler=CountingSampler(),
                     optimizer=HostOpt(list(model.parameters())))
    cond = {"input_ids": torch.randn(3, 4)}
    model.train()
    ev = loop.forward_only(cond)
    assert sorted(ev) == ["eval_loss", "eval_mse", "eval_nll"] and model.training and torch.is_grad_enabled()
    assert all(p.grad is None or float(p.grad.abs().sum()) == 0 for p in model.parameters())
    for _ in range(3):
        loop.run_step(cond)
    assert loop.opt.lrs == [2.0, 1.5, 1.0]                             # lr * (1 - step / learning_steps), train_util.py:266-272


def test_ema_rate_takes_the_reference_config_string():
    """config/train.py:19 declares `ema_rate: str` and utils/train_util.py:63-67 parses it with split(','): an integrator forwards
    args.ema_rate as is.  A float, a sequence and None (the reference's empty list) are accepted too."""
    model = torch.nn.Linear(3, 1)
    mk = lambda r: TrainStep(model, ToyDiffusion(), ema_rate=r, optimizer=HostOpt(list(model.parameters())))
    assert mk("0.5,0.9,0.99").ema_rate == [0.5, 0.9, 0.99]
    assert mk("0.9999").ema_rate == [0.9999]
    assert mk(0.99).ema_rate == [0.99]
    assert mk((0.5, 0.9)).ema_rate == [0.5, 0.9]
    assert mk(None).ema_rate == []


def test_optimize_clips_through_the_optimizer_and_returns_a_copy_of_the_norm():
    """train_util.py:246-264: clip (through `opt.clip_grad_norm` when the optimizer has one), log the norm, step.  The returned norm
    must not alias the optimizer's buffer (the fused optimizer overwrites it on the next step)."""
    model = torch.nn.Linear(3, 1)

    class ClipOpt(HostOpt):
        def __init__(self, params):
            super().__init__(params)
            self.buf, self.clipped = torch.zeros(1), []

        def clip_grad_norm(self, max_norm):
            self.clipped.append(max_norm)

        def grad_norm(self):
            self.buf.fill_(float(len(self.lrs) + 1))
            return self.buf
    opt = ClipOpt(list(model.parameters()))
    ts = TrainStep(model, ToyDiffusion(), gradient_clipping=0.5, optimizer=opt)
    g1 = ts.optimize()
    g2 = ts.optimize()
    assert opt.clipped == [0.5, 0.5] and float(g1) == 1.0 and float(g2) == 2.0 and g1.data_ptr() != opt.buf.data_ptr()


def test_clip_falls_back_to_torch_when_the_optimizer_has_no_clip_and_zero_ema_rate_means_none():
    """train_util.py:255-264: an optimizer without `clip_grad_norm` gets torch.nn.utils.clip_grad_norm_; :63-67: a falsy ema_rate
    (0.0, '', None) is the empty list."""
    torch.manual_seed(1)
    model = torch.nn.Linear(3, 1)
    ts = TrainStep(model, ToyDiffusion(), gradient_clipping=0.25, ema_rate=0.0, optimizer=HostOpt(list(model.parameters())))
    assert ts.ema_rate == []
    assert TrainStep(model, ToyDiffusion(), ema_rate="", optimizer=HostOpt(list(model.parameters()))).ema_rate == []
    ts.forward_backward({"input_ids": torch.randn(4, 3) * 10})
    assert float(torch.sqrt(sum((p.grad ** 2).sum() for p in model.parameters()))) > 0.25
    ts.optimize()
    assert abs(float(torch.sqrt(sum((p.grad ** 2).sum() for p in model.parameters()))) - 0.25) < 1e-4


def test_train_loop_has_the_reference_constructor_cadence_and_files(tmp_path):
    """utils/train_util.py:34-186: keyword-only constructor as run/train.py:132-151 calls it; run_loop stops after learning_steps,
    evaluates every eval_interval, saves every save_interval and once more at the end; a second TrainLoop on the same directory
    resumes from the newest model file (step parsed from its name) and anneals from there."""
    import inspect
    import itertools
    from musediffusion_amd.utils.train_util import TrainLoop, update_ema
    want = ["model", "diffusion", "data", "batch_size", "microbatch", "lr", "ema_rate", "log_interval", "save_interval", "resume_checkpoint",
            "schedule_sampler", "weight_decay", "learning_steps", "checkpoint_path", "gradient_clipping", "eval_data", "eval_interval",
            "eval_callbacks"]
    sig = inspect.signature(TrainLoop.__init__)
    assert [n for n in sig.parameters if n != "self"][:len(want)] == want
    assert all(sig.parameters[n].kind is inspect.Parameter.KEYWORD_ONLY for n in want)
    torch.manual_seed(0)
    model = torch.nn.Linear(4, 1)
    data = ({"input_ids": torch.randn(4, 4)} for _ in itertools.count())
    logs, evals = [], []
    mk = lambda m, steps, resume="": TrainLoop(
        model=m, diffusion=ToyDiffusion(), data=data, batch_size=4, microbatch=2, lr=1.0, ema_rate="0.9", log_interval=2,
        save_interval=3, resume_checkpoint=resume, schedule_sampler=CountingSampler(), weight_decay=0.0, learning_steps=steps,
        checkpoint_path=str(tmp_path), gradient_clipping=-1., eval_data=data, eval_interval=4, eval_callbacks=[lambda tl: evals.append(tl.step)],
        optimizer=HostOpt(list(m.parameters())), log_fn=logs.append)
    loop = mk(model, 7)
    assert loop.microbatch == 2 and loop.global_batch == 4 and loop.ema_rate == [0.9] and not loop.use_ddp and loop.resume_step == 0
    loop.run_loop()
    assert loop.step == 7 and loop.opt.lrs == [1.0 * (1 - k / 7) for k in range(7)]
    assert evals == [0, 4]                                             # step % eval_interval == 0
    assert sorted(os.listdir(tmp_path)) == ["model_000003.pt", "model_000006.pt"]   # steps 3 and 6; (7 - 1) % 3 == 0: no extra save
    assert any("eval_loss" in d for d in logs) and logs[0]["step"] == 0 and logs[0]["samples"] == 4 and "grad_norm" in logs[0]
    # resume: newest model file wins over the command-line checkpoint, lr anneals from its step
    m2 = torch.nn.Linear(4, 1)
    loop2 = mk(m2, 9, resume=str(tmp_path / "model_000003.pt"))
    assert loop2.resume_step == 6 and all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), model.state_dict().values()))
    loop2.run_loop()
    assert loop2.step == 3 and loop2.opt.lrs == [1.0 * (1 - (6 + k) / 9) for k in range(3)]
    assert "model_000009.pt" in os.listdir(tmp_path)                   # (step - 1) % save_interval != 0 -> the closing save
    assert TrainLoop.parse_resume_step_from_filename("x/model_000123.pt") == 123
    a, b = [torch.ones(2)], [torch.zeros(2)]
    update_ema(a, b, rate=0.75)
    assert torch.allclose(a[0], torch.full((2,), 0.75))


def test_checkpoint_save_writes_optimizer_state_without_ema_support_and_warns(tmp_path):
    """train_util.py:310-315 always writes opt_{step}.pt; an injected optimizer that has state but keeps no EMA copies still gets
    its file, and configured ema rates that nothing can serve are reported instead of silently skipped."""
    import pytest
    from musediffusion_amd import checkpoint as ckpt

    class StatefulOpt(HostOpt):
        def state_dict(self):
            return {"state": {0: {"step": torch.tensor(3.0)}}, "param_groups": [{"lr": 0.5}]}
    model = torch.nn.Linear(2, 1)
    with pytest.warns(UserWarning, match="keeps no EMA copies"):
        ckpt.save(str(tmp_path), 5, model, StatefulOpt(list(model.parameters())), [0.9])
    assert sorted(os.listdir(tmp_path)) == ["model_000005.pt", "opt_000005.pt"]
    assert float(torch.load(tmp_path / "opt_000005.pt")["state"][0]["step"]) == 3.0
