"""Worker of the multi-rank GPU tests (tests/test_multirank_gpu.py): `python tests/dist_worker.py CASE` with
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment.  One process per GPU over RCCL ("nccl") when the box has
at least WORLD_SIZE GPUs; on a one-GPU box every rank uses cuda:0 and the collectives run over gloo on the device tensors
(same code path above the backend).  Each rank asserts for itself and exits non-zero on failure."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import fixtures as fx  # noqa: E402


def setup():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ngpu = torch.cuda.device_count()
    if ngpu >= world:
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, torch.device("cuda", torch.cuda.current_device())


def build(tag, dev, rank, compute_dtype, train=False):
    from musediffusion_amd.models.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps
    from musediffusion_amd.models.network import TransformerNetModel
    c = fx.CONFIGS[tag]
    torch.manual_seed(1234 + rank)
    m = TransformerNetModel(c["E"], c["E"], c["Tt"], c["V"], c["L"], dropout=0.0, bert_hidden=c["H"], bert_layers=c["nL"],
                            bert_heads=c["nh"], bert_ffn=c["F"], compute_dtype=compute_dtype, bert_hidden_dropout=0.0, bert_attention_dropout=0.0)
    if rank == 0:
        m.load_state_dict(fx.state_dict(tag))          # only rank 0 holds the fixture weights
    m = (m.train().requires_grad_(True) if train else m.eval().requires_grad_(False)).to(dev)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                           rescale_timesteps=True, predict_xstart=True)
    return m, diff, c


def case_generate(rank, world, dev):
    """rank-0 weights -> one packed broadcast -> contiguous shards -> one token all-gather; the gathered tokens must be the
    reference's golden tokens of the UNSHARDED loops (sequences are independent), on every rank."""
    from conftest import load_golden
    from musediffusion_amd import sampling, sharding
    from oracle import sampling as osa
    tag = sys.argv[2] if len(sys.argv) > 2 else "tiny"
    m, diff, c = build(tag, dev, rank, sys.argv[3] if len(sys.argv) > 3 else "fp32")   # (the split-precision modes ship their hi / lo arena the same way)
    sharding.broadcast_weights(m, src=0, packed=True)
    sums = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sums, torch.tensor([sharding.weights_checksum(m)], dtype=torch.int64, device=dev))
    assert len({int(s) for s in sums}) == 1, "arena checksums differ: %s" % sums
    assert (rank == 0) != m.weights_from_arena
    g = load_golden("model_%s.npz" % tag)
    inp = fx.case_inputs(tag, fx.state_dict(tag)["word_embedding.weight"])
    B, L, E = c["B"], c["L"], c["E"]
    cond = {"input_ids": inp["batch"]["correct_ids"], "input_mask": inp["batch"]["input_mask"]}
    lo, hi = sharding.shard_bounds(B)

    def noises(seed, n, top_p):
        torch.manual_seed(seed)
        z = torch.zeros(B, L, E)
        return [(osa.truncated_noise(z, top_p) if top_p else torch.randn_like(z))[lo:hi] for _ in range(n)]
    for use_graph in (False, True):
        diff.use_graph = use_graph
        nz = noises(fx.loop_seed(tag, "ddim50"), 50, None)
        diff.noise_fn = lambda k, i, x: nz[k].to(dev)
        tok = sampling.generate(m, diff, cond, step=50, noise=inp["gen_noise0"])
        assert tok.shape == (B, L) and tok.dtype == torch.int64
        assert np.array_equal(tok.cpu().numpy(), g["loop_ddim50_tokens"]), "ddim50 tokens differ (graph=%s)" % use_graph
        nz2 = noises(fx.loop_seed(tag, "p12"), 12, 1)
        diff.noise_fn = lambda k, i, x: nz2[k].to(dev)
        tok = sampling.generate(m, diff, cond, t_enc=12, noise=inp["gen_noise0"])
        assert np.array_equal(tok.cpu().numpy(), g["loop_p12_tokens"]), "p12 tokens differ (graph=%s)" % use_graph
        nz3 = noises(fx.loop_seed(tag, "mod"), fx.NOISING_T, None)
        diff.noise_fn = lambda k, i, x: nz3[k].to(dev)
        tok = sampling.modify(m, diff, cond, step=200, strength=0.75, noise=inp["mod_noise"])
        assert np.array_equal(tok.cpu().numpy(), g["loop_mod_tokens"]), "mod tokens differ (graph=%s)" % use_graph


class ShardedCpuDraws:
    """torch.randn_like(x) on the device returns rows [lo, hi) of the draw a seeded CPU generator makes for the GLOBAL batch
    (the golden fixture drew full-batch tensors on the CPU generator)."""

    def __init__(self, seed, B, lo, hi):
        self.g, self.B, self.lo, self.hi = torch.Generator().manual_seed(seed), B, lo, hi

    def __enter__(self):
        self.orig = torch.randn_like

        def fake(x, **kw):
            full = torch.randn((self.B,) + tuple(x.shape[1:]), generator=self.g, dtype=torch.float32)
            return full[self.lo:self.hi].to(x.device)
        torch.randn_like = fake
        return self

    def __exit__(self, *a):
        torch.randn_like = self.orig


def case_ddp(rank, world, dev):
    """DDP over the custom-autograd backward: each rank back-propagates its rows of the golden batch (two micro-batches on
    rank layouts that allow it, the first under no_sync); the all-reduced gradients must equal the reference's full-batch
    gradients (tests/golden/losses_tiny.npz) on every rank."""
    from conftest import load_golden
    from torch.nn.parallel import DistributedDataParallel as DDP
    from musediffusion_amd import sharding
    from musediffusion_amd.train_step import TrainStep
    tag = "tiny"
    m, diff, c = build(tag, dev, rank, "fp32", train=True)
    sharding.broadcast_weights(m, src=0)
    ddp = DDP(m, device_ids=[dev.index], broadcast_buffers=False, bucket_cap_mb=128, find_unused_parameters=False)
    g = load_golden("losses_tiny.npz")
    li = fx.loss_inputs(tag)
    B = c["B"]
    lo, hi = sharding.shard_bounds(B)
    batch = {k: v[lo:hi] for k, v in li["batch"].items() if k != "correct_ids"}
    t_all, w_all = li["t"], li["w"]

    class FixedSampler:                      # the golden run fixed t and the weights
        def sample(self, n, device):
            return t_all[lo:hi].to(device), w_all[lo:hi].to(device)

    class NoOpt:
        def grad_norm(self):
            return torch.zeros(1, device=dev)

        def step(self, lr=None):
            pass
    loop = TrainStep(m, diff, microbatch=-1, schedule_sampler=FixedSampler(), ddp_model=ddp, optimizer=NoOpt())
    with ShardedCpuDraws(fx.loss_seed(tag), B, lo, hi):
        losses = loop.forward_backward(batch)
    # DDP averages over ranks: mean_r( (loss_r * w_r).mean() ) = the golden objective (loss * w).mean() when shards are equal
    assert (hi - lo) * world == B

    def close(name, got, ref, rel):
        ref = torch.from_numpy(np.asarray(ref))
        err = float((got.detach().float().cpu() - ref).abs().max())
        scale = float(ref.abs().max()) + 1e-12
        assert err <= rel * scale, "%s: err %.3e > %.1e * %.3e (rank %d)" % (name, err, rel, scale, rank)
    close("grad word_embedding", m.word_embedding.weight.grad, g["plain_g_word"], 2e-3)
    close("grad layer0.query", m.input_transformers.layer[0].attention.self.query.weight.grad, g["plain_g_q0"], 2e-3)
    close("grad time_embed.0", m.time_embed[0].weight.grad, g["plain_g_te0"], 2e-3)
    close("grad lm_head.bias", m.lm_head.bias.grad, g["plain_g_lmb"], 2e-3)
    # every rank holds the same gradients after the all-reduce
    flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    parts = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(parts, flat)
    assert all(torch.equal(parts[0], p) for p in parts[1:]), "gradients differ across ranks after DDP"


def case_untied(rank, world, dev):
    """After the reference's overload_embedding (utils/initialization.py:61-63) the head's weight is no longer the embedding tensor.
    Rank 0 overloads, every other rank starts with the default tie: after the packed broadcast every rank must hold rank 0's
    embedding AND rank 0's (different) lm_head.weight, and argmax_tokens - which reads lm_head.weight - must agree everywhere."""
    from musediffusion_amd import sharding
    tag = "tiny"
    m, diff, c = build(tag, dev, rank, "fp32")
    emb = fx.seeded_randn(4242, c["V"], c["E"]) * fx.EMB_STD
    if rank == 0:
        # what overload_embedding does to the model (utils/initialization.py:61-63; the helper itself ends in a barrier EVERY rank
        # must reach, and here only rank 0 overloads): the embedding's Parameter is replaced, the head keeps the old tensor
        with torch.no_grad():
            m.word_embedding.weight = torch.nn.Parameter(emb.to(dev))
        m.eval().requires_grad_(False)
        assert m.lm_head.weight is not m.word_embedding.weight
    else:
        assert m.lm_head.weight is m.word_embedding.weight
    sharding.broadcast_weights(m, src=0, packed=True)
    assert torch.equal(m.word_embedding.weight.cpu(), emb), "embedding did not arrive (rank %d)" % rank
    assert torch.equal(m.lm_head.weight.cpu(), fx.state_dict(tag)["word_embedding.weight"]), "stale lm_head.weight (rank %d)" % rank
    assert torch.equal(m.lm_head.bias.cpu(), fx.state_dict(tag)["lm_head.bias"])
    x = fx.seeded_randn(99, c["B"], c["L"], c["E"]).to(dev)
    tok = m.argmax_tokens(x)
    ref = torch.argmax(x.cpu() @ fx.state_dict(tag)["word_embedding.weight"].T + fx.state_dict(tag)["lm_head.bias"], dim=-1)
    assert float((tok.cpu() == ref).float().mean()) > 0.999, "argmax does not use rank 0's head (rank %d)" % rank
    toks = [torch.empty_like(tok) for _ in range(world)]
    dist.all_gather(toks, tok)
    assert all(torch.equal(toks[0], t) for t in toks[1:])


def case_lossaware(rank, world, dev):
    """The loss-aware sampler's DEVICE path with ranks whose micro-batches straddle a padding group (64 and 65 rows): with
    the shared bound TrainStep configures and with none (size exchange first) every rank enters the collective with equal buffers
    and ends in the state a host replay of the rank-ordered (timestep, loss) pairs gives."""
    from types import SimpleNamespace
    from musediffusion_amd.models.step_sample import LossSecondMomentResampler
    T = 7
    counts = [64, 65][:world] + [3] * max(0, world - 2)
    gen = torch.Generator().manual_seed(5)
    all_ts = [torch.randint(0, T, (n,), generator=gen) for n in counts]
    all_ls = [torch.rand(n, generator=gen, dtype=torch.float64) for n in counts]
    ref = LossSecondMomentResampler(SimpleNamespace(num_timesteps=T), history_per_term=4)
    for _ in range(2):
        ref.update_with_all_losses([int(v) for t in all_ts for v in t], [float(v) for l in all_ls for v in l])
    for bound in (70, None):
        s = LossSecondMomentResampler(SimpleNamespace(num_timesteps=T), history_per_term=4)
        s.max_local_batch = bound
        for _ in range(2):
            s.update_with_local_losses(all_ts[rank].to(dev), all_ls[rank].to(dev))
        assert np.array_equal(s._loss_counts, ref._loss_counts) and np.array_equal(s._loss_history, ref._loss_history), \
            "sampler state differs from the host replay (rank %d, bound %s)" % (rank, bound)
    s = LossSecondMomentResampler(SimpleNamespace(num_timesteps=T), history_per_term=4)
    s.max_local_batch = 64      # padded size 64: rank 0's 64 rows fit, rank 1's 65 do not - ONE rank over the bound
    try:
        s.update_with_local_losses(all_ts[rank].to(dev), all_ls[rank].to(dev))    # every rank enters the collective ...
        s._loss_counts                                                               # ... and every rank fails when it unpacks the blocks
    except ValueError:
        pass
    else:
        raise AssertionError("a micro-batch beyond max_local_batch was accepted (rank %d)" % rank)


if __name__ == "__main__":
    rank, world, dev = setup()
    try:
        {"generate": case_generate, "ddp": case_ddp, "untied": case_untied, "lossaware": case_lossaware}[sys.argv[1]](rank, world, dev)
        torch.cuda.synchronize()
        dist.barrier()
    finally:
        dist.destroy_process_group()
    print("rank %d ok" % rank)
