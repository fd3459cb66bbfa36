#!/usr/bin/env python3
"""bench.py -- denoiser-steps/sec of the captured reverse-diffusion step on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one `p_sample` iteration of the reference's sampling loop: denoiser
forward over [B, L, E] + nearest-embedding rounding + clamp + posterior mean + truncated-normal
noise (top_p = 1) + anchoring, i.e. one replay of the hipGraph built by
musediffusion_amd.models.diffusion._ReverseLoop.  Workload = BASELINE.json configs[1]
(seq_len 512, batch 64 per GPU, d_model 512, 12 layers, 8 heads, ffn 2048, E 128, vocab 729,
T = 2000 sqrt schedule), synthetic ComMU-shaped generation batch, torch-default random weights
(seed 0), bf16 compute with fp32 accumulation and fp32 latents.  With N > 1 every rank samples its
own batch of 64 (weak scaling; the reference shards whole batches over ranks, run/sample.py:169);
rank 0 builds the weights and ONE RCCL broadcast of the packed arena distributes them.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (the kernel with the largest time share - the
bf16 MFMA GEMM with the LayerNorm epilogue - timed live with HIP events on the launch streams) and, at N = 1,
`cpu_baseline` (the oracle = a torch-CPU fp32 port of the reference path, timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WORKLOADS = {
    # BASELINE.json configs[1]
    "c2": dict(L=512, B=64, E=128, H=512, nL=12, nh=8, F=2048, V=729, Tt=128, T=2000),
    # the reference-true BERT-base denoiser at the same batch (network.py:44)
    "c2-bertbase": dict(L=512, B=64, E=128, H=768, nL=12, nh=12, F=3072, V=729, Tt=128, T=2000),
    # BASELINE.json configs[0] shape (plumbing)
    "c1": dict(L=128, B=8, E=128, H=128, nL=2, nh=4, F=512, V=729, Tt=128, T=2000),
    # BASELINE.json configs[4]: training_losses, seq_len 1024, global batch 256 = 32 per GPU x 8 (DDP, RCCL all-reduce)
    "train": dict(L=1024, B=32, E=128, H=512, nL=12, nh=8, F=2048, V=729, Tt=128, T=2000),
}
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense peaks, MI355X_MICROARCH.md


def step_flops(c):
    """Algorithmic flops of one reverse step, 2 flops per MAC."""
    N, H, F, L, E, V, Tt, B = c["B"] * c["L"], c["H"], c["F"], c["L"], c["E"], c["V"], c["Tt"], c["B"]
    proj = 4 * (E * H + H * H) if E != H else 0
    per_tok = c["nL"] * (2 * (4 * H * H + 2 * H * F) + 4 * L * H) + proj + 2 * V * E
    return N * per_tok + 2 * B * (4 * Tt * Tt + 4 * Tt * H)


def build(c, dtype, device, seed=0):
    from musediffusion_amd.models.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps
    from musediffusion_amd.models.network import TransformerNetModel
    torch.manual_seed(seed)
    model = TransformerNetModel(c["E"], c["E"], c["Tt"], c["V"], c["L"], dropout=0.1, bert_hidden=c["H"],
                                bert_layers=c["nL"], bert_heads=c["nh"], bert_ffn=c["F"], compute_dtype=dtype)
    model.eval().requires_grad_(False).to(device)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(c["T"], [c["T"]]),
                           betas=get_named_beta_schedule("sqrt", c["T"]), rescale_timesteps=True, predict_xstart=True)
    return model, diff


def time_dominant_kernel(c, dtype, device, reps, branches=1):
    """Average launch duration of the step's dominant kernel (largest time share in profiles/), measured with HIP events
    on the launch streams, launched the way the step launches it.

    bf16 with d_model in {128, 256, 512}: the full-row GEMM with the bias + residual + LayerNorm epilogue
    (gemm_big_kernel<Row tile, EPI=3>), two launches per encoder layer and graph branch: attention-output dense
    (K = d_model) and FFN output dense (K = ffn), both N = d_model, on a branch's batch slice.  With two graph
    branches the step runs two of them side by side, so the timing loop does too (one stream per branch):
    `avg_launch_ms` = wall time of the region / launches (what the chip sustains), `launch_latency_ms` = the span one
    launch occupies on its own stream while it shares the chip with the other branch's (what rocprofv3 lists per kernel).  Otherwise (fp32 parity mode, d_model 768): the FFN intermediate dense with its GELU epilogue."""
    from musediffusion_amd import _lib, ops
    code = ops.dtype_code(dtype)
    td = ops.TORCH_DTYPE[code]
    panel = 1 if code == _lib.MH_BF16 else 0
    N_tok, H, F = (c["B"] // branches) * c["L"], c["H"], c["F"]
    lib = _lib.lib()
    fused = bool(panel and lib.mh_gemm_bias_res_ln_supported(H) and lib.mh_denoiser_get_fuse_ln())
    streams = [torch.cuda.Stream() for _ in range(branches)]
    bufs = []
    for _ in range(branches):
        bufs.append(dict(X=torch.randn(N_tok * H, device=device).to(td), Y=torch.empty(N_tok * H, device=device, dtype=td),
                         Fb=torch.randn(N_tok * F, device=device).to(td), Wa=(torch.randn(H * H, device=device) / 32).to(td),
                         W1=(torch.randn(F * H, device=device) / 32).to(td), W2=(torch.randn(F * H, device=device) / 32).to(td),
                         bias=torch.zeros(max(F, H), device=device), g=torch.ones(H, device=device)))

    def one_pass(bf, st):
        for _ in range(c["nL"]):
            if fused:
                for (A, W, K) in ((bf["X"], bf["Wa"], H), (bf["Fb"], bf["W2"], F)):
                    _lib.check(lib.mh_gemm_bias_res_ln(A.data_ptr(), N_tok, 1, W.data_ptr(), H, 1, bf["bias"].data_ptr(), bf["X"].data_ptr(),
                                                       N_tok, 1, bf["g"].data_ptr(), bf["bias"].data_ptr(), 1e-12, bf["Y"].data_ptr(), N_tok, 1,
                                                       N_tok, H, K, st))
            else:
                _lib.check(lib.mh_gemm_bias_act_ex(bf["X"].data_ptr(), N_tok if panel else H, panel, bf["W1"].data_ptr(), F if panel else H,
                                                   panel, bf["bias"].data_ptr(), None, 0, 0, bf["Fb"].data_ptr(), N_tok if panel else F, panel,
                                                   0, N_tok, F, H, 2, code, st))
    for bf, stream in zip(bufs, streams):
        one_pass(bf, stream.cuda_stream)
    torch.cuda.synchronize()
    # wall time of the whole region (all branches' launches running side by side) and each stream's own span
    cur = torch.cuda.current_stream()
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record(cur)
    evs = []
    for bf, stream in zip(bufs, streams):
        stream.wait_event(w0)
        with torch.cuda.stream(stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                one_pass(bf, stream.cuda_stream)
            e1.record(stream)
            evs.append((e0, e1))
    for _, e1 in evs:
        cur.wait_event(e1)
    w1.record(cur)
    torch.cuda.synchronize()
    per_pass = c["nL"] * (2 if fused else 1)
    latency_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs) / (reps * per_pass)   # one launch, as its stream sees it
    avg_ms = w0.elapsed_time(w1) / (reps * per_pass * branches)                           # wall time per launch
    if fused:
        flops = (2.0 * N_tok * H * H + 2.0 * N_tok * H * F) / 2
        name = ("gemm_big_kernel<128x512 full-row tile, EPI=bias+residual+LayerNorm> bf16: attention-output dense [%d x %d x %d] and FFN "
                "output dense [%d x %d x %d], %d launches/step on %d concurrent stream(s)" % (N_tok, H, H, N_tok, H, F, per_pass * branches, branches))
    else:
        flops = 2.0 * N_tok * F * H
        name = "%s: FFN intermediate dense + GELU [%d x %d x %d], %d launches/step" % (
            "gemm_big_kernel<EPI=0, ACT=gelu> bf16" if panel else "gemm_kernel<float, EPI=0>", N_tok, F, H, per_pass * branches)
    return avg_ms, flops, name, latency_ms


def pmc_traffic(workload, dtype, branches):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE); None when no pass was recorded for this workload."""
    try:
        with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f).get("%s/%s/b%d" % (workload, dtype, branches))
        return None if rec is None else float(rec["traffic_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(c, seconds_budget=25.0):
    """The oracle (torch-CPU fp32 port of the reference path) on a bounded sample of the same workload."""
    from musediffusion_amd import synthetic
    from oracle import denoiser as odn, sampling as osa, schedule as osc
    Bs = max(1, min(c["B"], 8))
    sd = odn.random_state_dict(c["E"], c["H"], c["F"], c["nL"], c["V"], c["L"], c["Tt"], seed=0)
    # give the CPU its best shot: pick the thread count that runs one denoiser forward fastest
    # (all hardware threads is far from optimal for torch's CPU GEMMs on many-core hosts)
    hw = os.cpu_count() or 1
    probe_x, probe_t = torch.randn(Bs, c["L"], c["E"]), torch.full((Bs,), 10.0)
    best, cores = None, hw
    for n in sorted({min(hw, k) for k in (8, 16, 32, 64, 128, hw)}):
        torch.set_num_threads(n)
        with torch.no_grad():
            odn.forward(sd, probe_x, probe_t, c["nh"])
            t0 = time.perf_counter()
            odn.forward(sd, probe_x, probe_t, c["nh"])
            dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, n
    torch.set_num_threads(cores)
    d = osc.make_diffusion(diffusion_steps=c["T"])
    batch = synthetic.generation_batch(Bs, c["L"], seed=1)
    emb_w = sd["word_embedding.weight"]
    x_start = emb_w[batch["input_ids"].long()]
    mask3 = torch.broadcast_to(batch["input_mask"].unsqueeze(-1), x_start.shape)
    torch.manual_seed(105)
    x = osa.start_latent_generation(x_start, mask3)
    fn = lambda xx, ts: odn.forward(sd, xx, ts, c["nh"])
    def step(i, x):
        t = torch.tensor([i] * Bs)
        with torch.no_grad():
            return osa.p_sample(d, fn, x, t, True, emb_w, top_p=1, mask=mask3, x_start=x_start)["sample"]
    x = step(c["T"] - 1, x)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        x = step(c["T"] - 2 - n, x)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 20 or (n >= 2 and el > seconds_budget / 2):
            break
    sps_sample = n / el
    return {"value": sps_sample * Bs / c["B"], "unit": "denoiser-steps/s", "cores": cores, "kind": "port",
            "sample": "oracle p_sample (torch CPU fp32, %d threads) on %d of the %d sequences, %d timed steps in %.1f s "
                      "after 1 warm-up; value = measured steps/s x %d/%d (cost is linear in batch)"
                      % (cores, Bs, c["B"], n, el, Bs, c["B"])}


def train_main(args, world, rank, local_rank, device):
    """`--workload train`: one step = training_losses forward + backward over one micro-batch (the reference's
    _forward_backward_logic, utils/train_util.py:188-232) with DDP gradient all-reduce when N > 1.  Reported
    separately from the headline sampling metric."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from musediffusion_amd import synthetic
    c = WORKLOADS["train"]
    model, diff = build(c, args.dtype, device, seed=0)
    model.train().requires_grad_(True)
    net = model
    if world > 1:
        model = DDP(model, device_ids=[local_rank], broadcast_buffers=False, bucket_cap_mb=128, find_unused_parameters=False)
    batch = {k: v.to(device) for k, v in synthetic.training_batch(c["B"], c["L"], seed=1 + rank).items()}
    g = torch.Generator().manual_seed(7 + rank)

    def step():
        t = torch.randint(0, c["T"], (c["B"],), generator=g).to(device)
        net.zero_grad(set_to_none=True)
        terms = diff.training_losses(model, t, model_kwargs=batch)
        terms["loss"].mean().backward()
        return terms

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        terms = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    assert bool(torch.isfinite(terms["loss"]).all())
    if rank == 0:
        N = c["B"] * c["L"]
        fwd = step_flops(dict(c)) - 2 * N * c["V"] * c["E"] + 2 * 2 * N * c["V"] * c["E"]   # denoiser + two CE heads
        out = {"metric": "training-steps/sec (training_losses fwd+bwd, seq_len=%d, batch=%d/GPU)" % (c["L"], c["B"]),
               "value": round(world * args.steps / elapsed, 3), "unit": "micro-batch steps/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "training_losses_seq2seq_with_corruption fwd+bwd, seq_len=%d batch=%d/GPU d_model=%d "
                                      "layers=%d" % (c["L"], c["B"], c["H"], c["nL"]),
                          "global_batch": c["B"] * world, "seq_len": c["L"], "parallelism": "ddp x%d" % world,
                          "tokens_per_s": round(world * args.steps * N / elapsed, 1),
                          "approx_tflops": round(3 * fwd / (elapsed / args.steps) / 1e12, 2)}}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--rng", default="philox", choices=["philox", "torch"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--split", type=int, default=None, help="batch slices run as concurrent graph branches (default: the library's choice)")
    ap.add_argument("--gemm", type=int, default=None, help="bf16 GEMM kernel variant 0/1/2 (see mh_gemm_set_variant)")
    ap.add_argument("--no-fuse-ln", action="store_true", help="A/B: separate GEMM and LayerNorm kernels")
    ap.add_argument("--no-stream-attn", action="store_true", help="A/B: LDS-resident attention instead of the streaming kernel")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    if args.workload == "train":
        return train_main(args, world, rank, local_rank, device)
    from functools import partial
    from musediffusion_amd import _lib, synthetic
    from musediffusion_amd.models.diffusion import _ReverseLoop
    from musediffusion_amd.models.rounding import denoised_fn_round
    from musediffusion_amd.sharding import broadcast_weights

    if args.gemm is not None:
        _lib.lib().mh_gemm_set_variant(args.gemm)
    if args.no_fuse_ln:
        _lib.lib().mh_denoiser_set_fuse_ln(0)
    if args.no_stream_attn:
        _lib.lib().mh_attention_set_stream(0)
    c = WORKLOADS[args.workload]
    model, diff = build(c, args.dtype, device, seed=0)
    if world > 1:
        broadcast_weights(model, src=0)   # ONE RCCL broadcast of the packed arena
    diff.rng_mode, diff.rng_seed, diff.rng_stream = args.rng, 105, rank
    diff.use_graph = not args.no_graph
    if args.split is not None:
        diff.batch_split = args.split

    batch = synthetic.generation_batch(c["B"], c["L"], seed=1 + rank)
    ids, mask = batch["input_ids"].to(device), batch["input_mask"].to(device)
    x_start = model.get_embeds(ids)
    mask3 = torch.broadcast_to(mask.unsqueeze(-1), x_start.shape)
    torch.manual_seed(105)
    x_noised = torch.where(mask3 == 0, x_start, torch.randn_like(x_start))
    model_emb = torch.nn.Embedding(c["V"], c["E"], _weight=model.word_embedding.weight.clone()).eval().requires_grad_(False)
    fn = partial(denoised_fn_round, model_emb, dist=None)

    total = args.warmup + args.steps
    indices = list(range(c["T"]))[::-1][:total]
    loop = _ReverseLoop.try_build(diff, "p", model, x_noised, True, fn, 1, mask3, x_start, 0.0, indices,
                                  lambda i: fn, False)
    assert loop is not None, "fused reverse loop unavailable"
    with torch.no_grad():
        loop.begin()
        for k in range(args.warmup):
            loop.advance(k)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.warmup, total):
            loop.advance(k)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    tokens = model.argmax_tokens(loop.x)       # the loop's product: discrete tokens (run/sample.py:219-220)
    assert tokens.shape == (c["B"], c["L"]) and bool(torch.isfinite(loop.x).all())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        flops = step_flops(c)
        out = {
            "metric": "denoiser-steps/sec (seq_len=%d, batch=%d)" % (c["L"], c["B"]),
            "value": round(value, 3), "unit": "denoiser-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "p_sample_loop step, %s: seq_len=%d batch=%d/GPU d_model=%d layers=%d heads=%d "
                                   "ffn=%d E=%d vocab=%d T=%d, rounding+clamp+top_p=1 every step"
                                   % (args.workload, c["L"], c["B"], c["H"], c["nL"], c["nh"], c["F"], c["E"], c["V"], c["T"]),
                       "global_batch": c["B"] * world, "seq_len": c["L"], "parallelism": "batch-sharded x%d" % world,
                       "rng": args.rng, "hipgraph": not args.no_graph, "graph_branches": int(loop.nsplit),
                       "step_tflop": round(flops / 1e12, 4),
                       "step_tflops_achieved": round(flops / (ms_per_step * 1e-3) / 1e12, 2),
                       "sequences_steps_per_s": round(value * c["B"], 1)},
        }
        if not args.no_kernel_timing:
            avg_ms, fpl, kname, lat_ms = time_dominant_kernel(c, args.dtype, device, reps=5, branches=int(loop.nsplit))
            ach = fpl / (avg_ms * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.dtype]
            out["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               "traffic": pmc_traffic(args.workload, args.dtype, int(loop.nsplit)), "avg_launch_ms": round(avg_ms, 5),
                               "concurrent_streams": int(loop.nsplit), "launch_latency_ms": round(lat_ms, 5),
                               "flops_per_launch": fpl}
        if not args.no_cpu_baseline and world == 1:     # the host baseline is a rank-0, N = 1 measurement
            out["cpu_baseline"] = cpu_baseline(c)
            out["cpu_baseline"]["value"] = round(out["cpu_baseline"]["value"], 5)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
