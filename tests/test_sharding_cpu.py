"""World-size-2 gloo tests (CPU) of the multi-GPU plumbing: one flat weight broadcast, contiguous
batch shards, ragged token all-gather, and the loss-aware sampler's single packed all_gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from musediffusion_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, fn_name, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, globals()[fn_name](rank, world)))
    finally:
        dist.destroy_process_group()


def _spawn(fn_name, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def _case_broadcast(rank, world):
    from musediffusion_amd.models.network import TransformerNetModel
    torch.manual_seed(100 + rank)     # different weights per rank before the broadcast
    m = TransformerNetModel(32, 32, 32, 50, 16, bert_hidden=64, bert_layers=1, bert_heads=2, bert_ffn=128)
    before = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
    sharding.broadcast_weights(m, src=0)
    after = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    assert m.lm_head.weight is m.word_embedding.weight
    return before.numpy(), after.numpy()


def _case_shard_gather(rank, world):
    B, L = 7, 5                                     # ragged: 4 + 3 rows
    batch = {"input_ids": torch.arange(B * L).view(B, L), "input_mask": torch.ones(B, L, dtype=torch.int)}
    local = sharding.shard_batch(batch)
    lo, hi = sharding.shard_bounds(B)
    assert local["input_ids"].shape[0] == hi - lo
    tokens = local["input_ids"] * 10 + rank         # stand-in for the sampled tokens of this shard
    full = sharding.gather_rows(tokens, B)
    return (lo, hi), full.numpy()


def _case_lossaware(rank, world):
    from types import SimpleNamespace
    from musediffusion_amd.models.step_sample import create_named_schedule_sampler
    s = create_named_schedule_sampler("lossaware", SimpleNamespace(num_timesteps=4))
    s.history_per_term = 2
    s._loss_history = np.zeros([4, 2])
    ts = torch.tensor([0, 1, 2] if rank == 0 else [3, 3])          # ragged local batches
    ls = torch.tensor([1.0, 2.0, 3.0] if rank == 0 else [4.0, 5.0])
    s.update_with_local_losses(ts, ls)
    ts2 = torch.tensor([0, 1] if rank == 0 else [2])
    s.update_with_local_losses(ts2, torch.tensor([6.0, 7.0] if rank == 0 else [8.0]))
    return s._loss_history.copy(), s._loss_counts.copy(), s.weights()


def test_broadcast_weights_single_flat_collective():
    out = _spawn("_case_broadcast")
    b0, a0 = out[0]
    b1, a1 = out[1]
    assert not np.array_equal(b0, b1)
    np.testing.assert_array_equal(a0, b0)       # src keeps its weights
    np.testing.assert_array_equal(a1, b0)       # the other rank received them


def test_shard_bounds_and_ragged_gather():
    out = _spawn("_case_shard_gather")
    assert out[0][0] == (0, 4) and out[1][0] == (4, 7)
    ids = np.arange(35).reshape(7, 5)
    expect = np.concatenate([ids[:4] * 10 + 0, ids[4:] * 10 + 1])
    np.testing.assert_array_equal(out[0][1], expect)
    np.testing.assert_array_equal(out[1][1], expect)
    # single process: identity
    assert sharding.shard_bounds(10, 0, 1) == (0, 10)
    assert [sharding.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_lossaware_sampler_state_identical_on_all_ranks():
    out = _spawn("_case_lossaware")
    h0, c0, w0 = out[0]
    h1, c1, w1 = out[1]
    np.testing.assert_array_equal(h0, h1)
    np.testing.assert_array_equal(c0, c1)
    np.testing.assert_array_equal(w0, w1)
    np.testing.assert_array_equal(c0, [2, 2, 2, 2])
    np.testing.assert_array_equal(h0, [[1, 6], [2, 7], [3, 8], [4, 5]])   # rank order, then arrival order


def _case_eight_ranks(rank, world):
    """BASELINE config 4's plumbing at its real world size: 512 sequences over 8 ranks (64 each, contiguous), one flat weight broadcast from
    rank 0, one token all-gather.  Host tensors over gloo: the collectives' call sites and the rank arithmetic, not the kernels."""
    from musediffusion_amd.models.network import TransformerNetModel
    torch.manual_seed(100 + rank)
    m = TransformerNetModel(32, 32, 32, 50, 16, bert_hidden=64, bert_layers=1, bert_heads=2, bert_ffn=128)
    sharding.broadcast_weights(m, src=0)
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    B, L = 512, 6
    batch = {"input_ids": torch.arange(B * L).view(B, L), "input_mask": torch.ones(B, L, dtype=torch.int)}
    lo, hi = sharding.shard_bounds(B)
    local = sharding.shard_batch(batch)
    assert (lo, hi) == (64 * rank, 64 * (rank + 1)) and local["input_ids"].shape[0] == 64
    tokens = local["input_ids"] * 3 + 1
    full = sharding.gather_rows(tokens, B)
    return float(flat.double().sum()), full.numpy()


def test_eight_rank_broadcast_shards_and_gather():
    out = _spawn("_case_eight_ranks", world=8)
    sums = {round(v[0], 6) for v in out.values()}
    assert len(sums) == 1                                          # every rank holds rank 0's weights
    expect = (torch.arange(512 * 6).view(512, 6) * 3 + 1).numpy()
    for r in range(8):
        np.testing.assert_array_equal(out[r][1], expect)          # every rank holds all 512 rows in batch order
