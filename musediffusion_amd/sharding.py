"""Multi-GPU plumbing of the sampling path: one process per GPU, torch.distributed over RCCL.

The path shards over independent sequences: the reference round-robins whole
batches over ranks and has every rank load the checkpoint itself (run/sample.py:85, :169).  Here
rank `src` owns the weights and ONE flat-buffer broadcast distributes them (instead of ~200
per-tensor broadcasts, utils/dist_util.py:141-152); each rank then samples its contiguous slice of
the batch with no per-step communication, and the int tokens are all-gathered once at the end
(replacing the per-rank broadcast+barrier loop at run/sample.py:222-291).
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def broadcast_weights(model, src=0, packed=False):
    """Make every rank hold rank `src`'s weights with ONE collective (utils/dist_util.py:141-152 does ~200).

    packed=False: one flat fp32 buffer of all parameters; afterwards `model.parameters()` are equal on every rank
    (training start-up, DDP).
    packed=True (sampling ranks, GPU only): the unit is the engine's packed inference arena (compute-dtype
    matrices + fp32 vectors, 78 MB at config 2 instead of 157 MB of fp32 parameters) followed in the same buffer
    by the three fp32 tensors the sampling path reads outside the arena (the embedding table for get_embeds /
    rounding, lm_head.bias and lm_head.weight for the argmax - the same tensor as the embedding unless overload_embedding untied it).  Receivers do NOT re-pack: their engine is pinned to the received arena.
    Their other fp32 `nn.Parameter`s keep their old values, so the model is marked `weights_from_arena` and
    refuses to train or re-pack until `load_state_dict` / a flat broadcast refreshes it."""
    if world() == 1:
        return model
    if packed:
        eng = model.engine()                       # src: packed from its parameters; others: allocated with the same plan
        # lm_head.weight travels too: it is the embedding tensor only until overload_embedding (utils/initialization.py:61-63)
        # replaces the embedding's Parameter, and argmax_tokens / get_logits read lm_head.weight
        extra = [model.word_embedding.weight.data, model.lm_head.bias.data, model.lm_head.weight.data]
        tail = torch.cat([t.reshape(-1) for t in extra]).contiguous().view(torch.uint8)
        flat = torch.cat([eng.arena, tail])
        dist.broadcast(flat, src=src)
        if rank() != src:
            n = eng.arena.numel()
            eng.arena.copy_(flat[:n])
            off, got = n, []
            for t in extra:
                nb = t.numel() * 4
                got.append(flat[off:off + nb].view(torch.float32).view_as(t))
                off += nb
            tied_here = model.lm_head.weight is model.word_embedding.weight
            if tied_here and not torch.equal(got[0], got[2]):     # the source's head left the tie: leave it here as well
                with torch.no_grad():
                    model.lm_head.weight = torch.nn.Parameter(got[2].clone(), requires_grad=model.lm_head.weight.requires_grad)
                extra[2] = model.lm_head.weight.data
            for t, g in zip(extra, got):
                t.copy_(g)
            model.pin_engine()
        return model
    params = [p.data for p in model.parameters()]
    flat = torch.cat([p.reshape(-1) for p in params])
    dist.broadcast(flat, src=src)
    off = 0
    for p in params:
        n = p.numel()
        p.copy_(flat[off:off + n].view_as(p))
        off += n
    return model


def weights_checksum(model):
    """A scalar every rank can compare after the broadcast: the byte sum of the inference arena when the model runs
    on the GPU (that is what the sampling kernels read), the fp32 parameter sum otherwise."""
    if next(model.parameters()).is_cuda:
        a = model.engine().arena
        pad = (-a.numel()) % 8
        if pad:
            a = torch.cat([a, torch.zeros(pad, dtype=torch.uint8, device=a.device)])
        return int(a.view(torch.int64).sum().item())
    return float(sum(p.detach().double().sum() for p in model.parameters()))


def shard_bounds(n, r=None, w=None):
    """Contiguous slice [lo, hi) of n items owned by rank r of w (remainder spread over the first ranks)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    base, extra = divmod(n, w)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def shard_batch(batch, r=None, w=None):
    """Slice every [B, ...] tensor of a model_kwargs batch to this rank's rows."""
    n = next(iter(batch.values())).shape[0]
    lo, hi = shard_bounds(n, r, w)
    return {k: v[lo:hi] for k, v in batch.items()}


def gather_rows(local, total_rows):
    """All-gather row shards produced under `shard_bounds` back into one [total_rows, ...] tensor
    (ragged shards are padded to the largest one for the collective)."""
    w = world()
    if w == 1:
        return local
    sizes = [shard_bounds(total_rows, r, w) for r in range(w)]
    max_rows = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)
