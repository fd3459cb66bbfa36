"""Multi-rank runs of the two sharded paths on the GPU: sampling (packed weight broadcast -> batch shards ->
token all-gather, against the reference's golden tokens) and training (DDP gradient all-reduce over the kernel-level
backward, against the reference's golden gradients).  One process per GPU over RCCL when the box has two GPUs; on a one-GPU
box both ranks share cuda:0 and the collectives run over gloo (tests/dist_worker.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

from conftest import REPO, free_port  # noqa: E402


def run_ranks(case, *extra, world=2, timeout=600):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dist_worker.py"), case, *extra], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out[-4000:])
        assert "rank %d ok" % r in out


@pytest.mark.parametrize("tag,mode", [("tiny", "fp32"), ("c1", "fp32"), ("c1", "f16x3")])
def test_two_rank_sharded_sampling_matches_golden_tokens(tag, mode):
    run_ranks("generate", tag, mode)


def test_two_rank_ddp_gradients_match_golden():
    run_ranks("ddp")


def test_two_rank_packed_broadcast_ships_an_untied_head():
    run_ranks("untied", timeout=240)


def test_two_rank_lossaware_gather_with_unequal_micro_batches():
    run_ranks("lossaware", timeout=240)


@pytest.mark.parametrize("workload", ["c2", "c4", "train"])
def test_bench_gpus_2_self_launches_and_reports_two_ranks(workload):
    """`python bench.py --gpus 2` with NO launcher around it (what a SCALE run does): the parent starts two ranks through
    torch.distributed.run, they broadcast rank 0's weights, run the workload, gather, and rank 0 prints one JSON line with
    n_gpus = 2.  Two GPUs: one rank per GPU over RCCL (backend nccl, rccl_ranks 2, two distinct device uuids); one GPU: both ranks
    share it and the collectives run over gloo (MUSE_BENCH_SHARE_GPU) - the code path above the backend is the same, and the line
    must SAY so: backend gloo, rccl_ranks 0, one distinct device (the fields that make an 8-GPU SCALE line un-fakeable)."""
    import json
    import torch
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    if torch.cuda.device_count() < 2:
        env["MUSE_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-kernel-timing"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and len(out["config"]["devices"]) == 2
    if env.get("MUSE_BENCH_SHARE_GPU") == "1":
        assert out["config"]["backend"] == "gloo" and out["config"]["rccl_ranks"] == 0 and out["config"]["distinct_devices"] == 1
    else:
        assert out["config"]["backend"] == "nccl" and out["config"]["rccl_ranks"] == 2 and out["config"]["distinct_devices"] == 2
    if workload != "train":
        assert isinstance(out["config"]["arena_checksum_after_broadcast"], int)
    else:
        assert out["config"]["all_reduce_exposed_ms"] is not None and out["config"]["gradient_bytes_all_reduced_per_step"] > 0
    assert out["scaling"] == "weak" and out["config"]["global_batch"] == 2 * (32 if workload == "train" else 64)


def test_bench_gpus_8_config4_self_launch_at_its_real_world_size():
    """BASELINE configs[3] at the world size it names: `python bench.py --gpus 8 --workload c4` with no launcher around it - the parent
    starts EIGHT ranks through torch.distributed.run (port choice, OMP_NUM_THREADS, one process per rank), rank 0's packed arena is
    broadcast to seven receivers, eight reverse loops capture their graphs concurrently, 512 sequences are sampled as eight contiguous
    shards and the tokens are gathered 8 ways; rank 0 prints ONE line with n_gpus = 8.  On a box with fewer than eight GPUs all ranks
    share cuda:0 and the collectives run over gloo (MUSE_BENCH_SHARE_GPU): the launcher, the rank logic and every collective call
    site of the real entry point run end to end, and the line says what ran (backend gloo, rccl_ranks 0, one distinct device)."""
    import json
    import torch
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    shared = torch.cuda.device_count() < 8
    if shared:
        env["MUSE_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--workload", "c4", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-kernel-timing"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["value"] > 0 and len(out["config"]["devices"]) == 8
    assert out["scaling"] == "weak" and out["config"]["global_batch"] == 8 * 64
    assert isinstance(out["config"]["arena_checksum_after_broadcast"], int)
    if shared:
        assert out["config"]["backend"] == "gloo" and out["config"]["rccl_ranks"] == 0 and out["config"]["distinct_devices"] == 1
    else:
        assert out["config"]["backend"] == "nccl" and out["config"]["rccl_ranks"] == 8 and out["config"]["distinct_devices"] == 8
