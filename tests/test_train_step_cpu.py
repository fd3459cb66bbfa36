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
