"""Round-4 kernels against a torch fp32 reference of the same op on the same (bf16-rounded) operands, and against the forms they
replace: the head of the denoiser (input_up_proj + position / time add + embedding LayerNorm, models/network.py:141-149) and its tail
(output_down_proj, :153-157) as one kernel each (csrc/headtail.hip)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from musediffusion_amd import _lib  # noqa: E402
from musediffusion_amd._lib import check, current_stream, lib  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def q16(t):
    return t.bfloat16().float()


def to_panel(w):
    """[rows, K] fp32 -> bf16 K32 panels [K / 32][rows][32] on the device (the arena's layout, engine._put_mat)"""
    r, k = w.shape
    return w.bfloat16().reshape(r, k // 32, 32).permute(1, 0, 2).contiguous().to(DEV)


def from_panel(p):
    """bf16 [C / 32][rows][32] -> [rows, C] fp32 on the host"""
    return p.permute(1, 0, 2).reshape(p.shape[1], -1).float().cpu()


# H 768: bert-base, the width the reference itself builds (network.py:44-46) - 32 rows per block, 12 waves, two-slot weight ring
@pytest.mark.parametrize("H,E,B,L", [(512, 128, 2, 64), (512, 128, 3, 200), (256, 64, 2, 72), (512, 128, 32, 512),
                                     (768, 128, 2, 64), (768, 128, 3, 200), (768, 96, 2, 72), (768, 128, 16, 512), (768, 128, 1, 8), (512, 128, 1, 8)])
def test_up_proj_ln_fused(H, E, B, L):
    """(pos + (tanh(x W0^T + b0) W2^T + b2)) + emb, LayerNorm: the intermediate rounded to bf16 once (the second GEMM's operand), fp32
    from there to the normalised row.  Rows that do not fill the last row block (3 x 200, 2 x 72) are covered."""
    N = B * L
    x = rnd(N, E, seed=1, scale=0.5)
    W0, b0 = rnd(H, E, seed=2, scale=1 / math.sqrt(E)), rnd(H, seed=3, scale=0.1)
    W2, b2 = rnd(H, H, seed=4, scale=1 / math.sqrt(H)), rnd(H, seed=5, scale=0.1)
    pos, emb = rnd(L, H, seed=6, scale=0.5), rnd(B + 2, H, seed=7, scale=0.5)
    rows = torch.tensor([(b + 2) % (B + 2) for b in range(B)], dtype=torch.int32)
    g, bt = 1 + rnd(H, seed=8, scale=0.2), rnd(H, seed=9, scale=0.2)
    h1 = q16(torch.tanh(q16(x) @ q16(W0).T + b0))
    pre = (pos[None] + (h1 @ q16(W2).T + b2).view(B, L, H)) + emb[rows.long()][:, None]
    ref = torch.nn.functional.layer_norm(pre, (H,), g, bt, 1e-12).view(N, H)
    Ep = (E + 31) // 32 * 32
    assert lib().mh_up_proj_ln_fused_supported(E, Ep, H) == 1 and lib().mh_up_proj_ln_fused_supported(500, 512, H) == 0
    out = torch.zeros(H // 32, N, 32, device=DEV, dtype=torch.bfloat16)
    d = lambda t: t.to(DEV).contiguous()
    xd, w0p, w2p, b0d, b2d, posd, embd, rowsd, gd, btd = d(x), to_panel(W0), to_panel(W2), d(b0), d(b2), d(pos), d(emb), d(rows), d(g), d(bt)
    check(lib().mh_up_proj_ln_fused(xd.data_ptr(), E, Ep, w0p.data_ptr(), b0d.data_ptr(), w2p.data_ptr(), b2d.data_ptr(), posd.data_ptr(),
                                    embd.data_ptr(), rowsd.data_ptr(), gd.data_ptr(), btd.data_ptr(), 1e-12, out.data_ptr(), N, B, L, H,
                                    current_stream()), "mh_up_proj_ln_fused")
    got = from_panel(out)
    err = (got - ref).abs()
    print("head H=%d rows=%d: max |d| %.4f mean %.5f" % (H, N, float(err.max()), float(err.mean())))
    assert float(err.max()) < 4e-2 and float(err.mean()) < 4e-3          # bf16 output rounding of O(1) values + fp32 summation order
    # emb_row = NULL: row b of emb_t
    check(lib().mh_up_proj_ln_fused(xd.data_ptr(), E, Ep, w0p.data_ptr(), b0d.data_ptr(), w2p.data_ptr(), b2d.data_ptr(), posd.data_ptr(),
                                    embd.data_ptr(), None, gd.data_ptr(), btd.data_ptr(), 1e-12, out.data_ptr(), N, B, L, H,
                                    current_stream()), "mh_up_proj_ln_fused")
    pre2 = (pos[None] + (h1 @ q16(W2).T + b2).view(B, L, H)) + emb[:B][:, None]
    ref2 = torch.nn.functional.layer_norm(pre2, (H,), g, bt, 1e-12).view(N, H)
    assert float((from_panel(out) - ref2).abs().max()) < 4e-2


@pytest.mark.parametrize("H,E,N", [(512, 128, 128), (512, 128, 600), (256, 64, 136), (512, 128, 16384),
                                   (768, 128, 128), (768, 128, 600), (768, 128, 8200), (768, 128, 8), (512, 128, 8)])
def test_down_proj_fused(H, E, N):
    X = rnd(N, H, seed=11, scale=1.0)
    W0, b0 = rnd(H, H, seed=12, scale=1 / math.sqrt(H)), rnd(H, seed=13, scale=0.1)
    W2, b2 = rnd(E, H, seed=14, scale=1 / math.sqrt(H)), rnd(E, seed=15, scale=0.1)
    ref = q16(torch.tanh(q16(X) @ q16(W0).T + b0)) @ q16(W2).T + b2
    ld = N + 64                                                             # a row window of a larger panel buffer
    Xp = torch.zeros(H // 32, ld, 32, device=DEV, dtype=torch.bfloat16)
    Xp[:, :N] = X.bfloat16().reshape(N, H // 32, 32).permute(1, 0, 2).to(DEV)
    out, sq = torch.zeros(N, E, device=DEV), torch.zeros(N, device=DEV)
    assert lib().mh_down_proj_fused_supported(E, H) == 1 and lib().mh_down_proj_fused_supported(500, H) == 0
    d = lambda t: t.to(DEV).contiguous()
    w0p, w2p, b0d, b2d = to_panel(W0), to_panel(W2), d(b0), d(b2)
    check(lib().mh_down_proj_fused(Xp.data_ptr(), ld, w0p.data_ptr(), b0d.data_ptr(), w2p.data_ptr(), b2d.data_ptr(), out.data_ptr(), sq.data_ptr(), N, E, H,
                                   current_stream()), "mh_down_proj_fused")
    err = (out.cpu() - ref).abs()
    print("tail H=%d rows=%d: max |d| %.2e" % (H, N, float(err.max())))
    assert float(err.max()) < 2e-3                                          # fp32 output: only the accumulation order differs
    sq_ref = (out.double() ** 2).sum(dim=1)
    assert float(((sq.double() - sq_ref).abs() / sq_ref).max()) < 1e-5       # |row|^2 of the rows the kernel itself wrote


def test_forward_with_fused_head_and_tail_tracks_the_separate_launches():
    """BASELINE config 2's shape (2 layers, 4 sequences): the forward with the one-kernel head / tail against the round-3 launch
    sequence (debug library switch).  Not bit-identical by design - the fused head keeps fp32 between the second dense layer and the
    LayerNorm where the separate launches round to bf16 - so the comparison is at bf16 resolution, and both are compared with the
    fp32 oracle."""
    from musediffusion_amd.models.network import TransformerNetModel
    from oracle import denoiser as odn
    torch.manual_seed(5)
    E, H, L, B = 128, 512, 512, 4
    m = TransformerNetModel(E, E, 128, 729, L, dropout=0.0, bert_hidden=H, bert_layers=2, bert_heads=8, bert_ffn=2048, compute_dtype="bf16")
    m.eval().requires_grad_(False).to(DEV)
    x = rnd(B, L, E, seed=21).to(DEV)
    t = torch.tensor([3.0, 500.5, 999.0, 17.0], device=DEV)
    with torch.no_grad():
        y_fused = m(x, t).cpu()
        with _lib.debug_library():
            _lib.lib().mh_denoiser_set_fuse_headtail(0)
            try:
                m._engine = None
                y_sep = m(x, t).cpu()
            finally:
                _lib.lib().mh_denoiser_set_fuse_headtail(1)
        m._engine = None
        sd = {k: v.detach().cpu().float() for k, v in m.state_dict().items()}
        ref = odn.forward(sd, x.cpu(), t.cpu(), 8)
    d_fs, d_f, d_s = (y_fused - y_sep).abs(), (y_fused - ref).abs(), (y_sep - ref).abs()
    print("fused vs separate: max %.4f mean %.5f; vs oracle: fused mean %.5f max %.4f, separate mean %.5f max %.4f"
          % (float(d_fs.max()), float(d_fs.mean()), float(d_f.mean()), float(d_f.max()), float(d_s.mean()), float(d_s.max())))
    assert float(d_fs.mean()) < 5e-3 and float(d_fs.max()) < 0.1
    assert float(d_f.mean()) < 0.02 and float(d_f.max()) < 0.25                # the stated bf16 tolerance
    assert float(d_f.mean()) <= float(d_s.mean()) * 1.1                          # and no worse than the separate launches


@pytest.mark.parametrize("kind", ["p", "ddim"])
def test_step_with_fused_rounding_pieces_equals_the_separate_launches(kind):
    """One captured step with |row|^2 from the fused down-projection + mh_round_scores + mh_step_epilogue_slots against the round-3
    sequence (row_sqnorm, score GEMM, argbest_reduce, update) on the same bf16 model, same start latent, same Philox noise, 3 steps.
    The only arithmetic difference is the summation order inside |x_n|^2, which enters a score as (|W_v|^2 + |x_n|^2) - 2 x.W_v: rows
    change their nearest embedding only on last-bit ties.  mh_step_advance must leave the loop state where begin + end left it."""
    from functools import partial
    from musediffusion_amd import synthetic
    from musediffusion_amd.models.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps, _ReverseLoop
    from musediffusion_amd.models.network import TransformerNetModel
    from musediffusion_amd.models.rounding import denoised_fn_round
    torch.manual_seed(7)
    E, H, L, B, V = 128, 512, 512, 4, 729
    m = TransformerNetModel(E, E, 128, V, L, dropout=0.0, bert_hidden=H, bert_layers=2, bert_heads=8, bert_ffn=2048, compute_dtype="bf16")
    m.eval().requires_grad_(False).to(DEV)
    batch = synthetic.generation_batch(B, L, seed=1)
    ids, mask = batch["input_ids"].to(DEV), batch["input_mask"].to(DEV)
    x_start = m.get_embeds(ids)
    mask3 = torch.broadcast_to(mask.unsqueeze(-1), x_start.shape)
    torch.manual_seed(105)
    x0 = torch.where(mask3 == 0, x_start, torch.randn_like(x_start))
    emb = torch.nn.Embedding(V, E, _weight=m.word_embedding.weight.clone()).eval().requires_grad_(False)
    fn = partial(denoised_fn_round, emb, dist=None)
    res = {}
    for fused, graph, own_noise in ((True, True, True), (True, False, True), (False, True, True), (False, False, True), (True, True, False),
                                    ("in the forward", True, True), ("in the forward", False, True), ("in the forward", True, "separate update")):
        if True:
            diff = SpacedDiffusion(use_timesteps=space_timesteps(2000, [2000]), betas=get_named_beta_schedule("sqrt", 2000),
                                   rescale_timesteps=True, predict_xstart=True)
            diff.rng_mode, diff.rng_seed, diff.rng_stream, diff.use_graph, diff.fuse_rounding, diff.batch_split = "philox", 105, 0, graph, fused, 2
            diff.fuse_noise = own_noise          # True: the update kernel draws the noise; False: mh_trunc_normal at the head of the step
            diff.round_in_forward = fused == "in the forward"      # the forward's last kernel rounds its rows (split-bf16 scores)
            diff.update_in_forward = own_noise is True             # ... and takes the reverse step of its rows
            if own_noise == "separate update":
                own_noise = True
            fused_name, fused = fused, bool(fused)
            idx = list(range(2000))[::-1][:3]
            loop = _ReverseLoop.try_build(diff, kind, m, x0, True, fn, 1 if kind == "p" else None, mask3, x_start, 0.0, idx, lambda i: fn, False)
            assert loop is not None and loop.fused_round == fused and loop.round_in_tail == (fused_name == "in the forward")
            with torch.no_grad():
                loop.begin()
                for k in range(3):
                    loop.advance(k)
                loop.finish()
            torch.cuda.synchronize()
            for state in (loop.br_state if loop.decoupled else [loop.state]):   # (decoupled chains keep one loop state per batch slice)
                st = state.cpu().tolist()
                assert st[0] == 3 and st[1] == 3 and st[2] == idx[2]          # pos, n_steps, cur_t after three steps
            assert loop.decoupled == bool(graph and (loop.fused_round if diff.decouple_branches is None else diff.decouple_branches))
            key = (fused_name, graph) if own_noise else "separate noise launch"
            if fused_name == "in the forward" and not diff.update_in_forward:
                key = "separate update"
            res[key] = (loop.x.clone(), loop.round_idx.clone(), loop.pred.clone())
    # the noise drawn inside the update kernel is the noise mh_trunc_normal writes: identical samples, bit for bit
    assert all(torch.equal(a, b) for a, b in zip(res[(True, True)], res["separate noise launch"]))
    # the reverse step taken inside the forward's last kernel == the separate update kernel on the same indices, bit for bit
    assert all(torch.equal(a, b) for a, b in zip(res[("in the forward", True)], res["separate update"]))
    # rounding inside the forward (split-bf16 scores) against the exact-fp32 score GEMM: the same rows except on near-ties
    same_f = float((res[("in the forward", True)][1] == res[(True, True)][1]).float().mean())
    print("%s: rounded index agreement, scores inside the forward vs exact-fp32 score GEMM %.6f" % (kind, same_f))
    assert same_f >= 0.999 and all(torch.equal(a, b) for a, b in zip(res[("in the forward", True)], res[("in the forward", False)]))
    for fused in (True, False):                                             # eager and captured step: the same launches
        assert torch.equal(res[(fused, True)][0], res[(fused, False)][0]) and torch.equal(res[(fused, True)][1], res[(fused, False)][1])
    (xa, ia, pa), (xb, ib, pb) = res[(True, True)], res[(False, True)]
    same = float((ia == ib).float().mean())
    print("%s: rounded index agreement fused vs separate %.6f" % (kind, same))
    assert same >= 0.9995
    rows_same = (ia == ib).view(B, L, 1).expand_as(xa)
    assert torch.equal(xa[rows_same], xb[rows_same]) and torch.equal(pa[rows_same], pb[rows_same])


@pytest.mark.parametrize("H", [512, 768])
@pytest.mark.parametrize("N,V", [(128, 729), (1000, 729), (16384, 729), (192, 650)])
def test_down_proj_with_rounding_inside(N, V, H):
    """mh_down_proj_round_fused: the down-projection's rows AND their nearest embedding row (models/rounding.py:21-28) from one kernel,
    the scores on the bf16 matrix pipe as hi / lo parts (x_hi T_hi + x_lo T_hi + x_hi T_lo, fp32 accumulation).  Checked against the
    float64 argmin of |x - T_v|^2 on the rows the kernel itself wrote: every row whose two best distances differ by more than the
    split's resolution must agree (the rest are reported), and planted exact matches (x = a table row) must be found."""
    E = 128
    X = rnd(N, H, seed=31, scale=1.0)
    W0, b0 = rnd(H, H, seed=32, scale=1 / math.sqrt(H)), rnd(H, seed=33, scale=0.1)
    W2, b2 = rnd(E, H, seed=34, scale=1 / math.sqrt(H)), rnd(E, seed=35, scale=0.1)
    table = rnd(V, E, seed=36, scale=1.0)
    Xp = X.bfloat16().reshape(N, H // 32, 32).permute(1, 0, 2).contiguous().to(DEV)
    out, sq, idx = torch.zeros(N, E, device=DEV), torch.zeros(N, device=DEV), torch.full((N,), -1, device=DEV, dtype=torch.int32)
    d = lambda t: t.to(DEV).contiguous()
    assert lib().mh_down_proj_round_supported(E, H, V) == 1 and lib().mh_down_proj_round_supported(E, H, 500) == 0
    td = d(table)
    buf = torch.empty(int(lib().mh_round_split_bytes(E, V)), dtype=torch.uint8, device=DEV)
    check(lib().mh_round_split_table(td.data_ptr(), None, V, E, buf.data_ptr(), current_stream()), "mh_round_split_table")
    w0p, w2p, b0d, b2d = to_panel(W0), to_panel(W2), d(b0), d(b2)
    check(lib().mh_down_proj_round_fused(Xp.data_ptr(), N, w0p.data_ptr(), b0d.data_ptr(), w2p.data_ptr(), b2d.data_ptr(), out.data_ptr(),
                                         sq.data_ptr(), buf.data_ptr(), V, idx.data_ptr(), None, N, E, H, current_stream()), "mh_down_proj_round_fused")
    ref = q16(torch.tanh(q16(X) @ q16(W0).T + b0)) @ q16(W2).T + b2
    assert float((out.cpu() - ref).abs().max()) < 2e-3
    y = out.cpu().double()
    dist = ((y[:, None, :] - table.double()[None]) ** 2).sum(-1) if N <= 1000 else None
    if dist is None:
        t2 = (table.double() ** 2).sum(1)
        dist = (y ** 2).sum(1, keepdim=True) + t2[None] - 2 * y @ table.double().T
    best2 = torch.topk(-dist, 2, dim=1)
    margin = (best2.values[:, 0] - best2.values[:, 1])                     # >= 0: gap between the best and the second-best distance
    got = idx.cpu().long()
    assert int(got.min()) >= 0 and int(got.max()) < V
    agree = got == best2.indices[:, 0]
    clear = margin > 1e-3 * dist.abs().max()                                # split resolution 2^-16 of |x||T| ~ 1e-5 relative: 1e-3 is generous
    print("rows %d: agreement %.5f, clear rows %.5f" % (N, float(agree.float().mean()), float(clear.float().mean())))
    assert bool(agree[clear].all()) and float(agree.float().mean()) > 0.999
    sq_ref = (y ** 2).sum(1)
    assert float(((sq.cpu().double() - sq_ref).abs() / sq_ref).max()) < 1e-5


@pytest.mark.parametrize("L,B,nh", [(512, 3, 5), (1024, 2, 3)])
def test_attention_with_two_query_tiles_per_wave_equals_the_16_wave_kernel(L, B, nh):
    """A/B geometry (debug library, mh_attention_set_stream(7)): 8 waves x 64 queries instead of 16 x 32 - every query sees the same
    arithmetic in the same order (one K / V^T fragment read feeds both of a wave's query tiles), so the context rows are bit-identical."""
    from test_kernels_gpu import _vt_perm
    dh = 64
    qh, kh, vh = (rnd(B, nh, L, dh, seed=330 + i) for i in range(3))
    kh = kh * 1.5
    kh[:, :, 300] *= 4.0
    vtp = _vt_perm(vh.transpose(-1, -2).contiguous())
    vt_dev = torch.zeros(vtp.numel() + 128, device=DEV, dtype=torch.bfloat16)
    vt_dev[: vtp.numel()] = vtp.to(DEV).bfloat16().flatten()
    qd, kd = qh.to(DEV).bfloat16().contiguous(), kh.to(DEV).bfloat16().contiguous()
    outs = []
    with _lib.debug_library():
        for mode in (1, 7):
            _lib.lib().mh_attention_set_stream(mode)
            try:
                for panel in (0, 1):
                    out = torch.zeros(nh * dh // 32, B * L, 32, device=DEV, dtype=torch.bfloat16) if panel else torch.zeros(B * L, nh * dh, device=DEV, dtype=torch.bfloat16)
                    check(_lib.lib().mh_attention_stream_fwd(qd.data_ptr(), kd.data_ptr(), vt_dev.data_ptr(), out.data_ptr(), B * L if panel else nh * dh, panel,
                                                             B, L, nh, dh, 1.0 / math.sqrt(dh), current_stream()))
                    outs.append(out.clone())
            finally:
                _lib.lib().mh_attention_set_stream(1)
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])
    assert float(outs[0].float().abs().max()) > 0.1


@pytest.mark.parametrize("H", [512, 768])
def test_fused_head_and_tail_repeat_launches_are_bit_identical_under_load(H):
    """Race screen for the barrier-free K loops (a wave's LDS-DMA stages are private to it, ordered by its own counted vmcnt / lgkmcnt
    only): 120 launches of each kernel on the same operands while a second stream keeps the chip's L2 / DMA paths busy with GEMMs
    of another size - every launch must reproduce the first one bit for bit (a read that overtakes its DMA shows up as a changed row)."""
    E, V, B, L = 128, 729, 8, 512
    N = B * L
    x = rnd(N, E, seed=41, scale=0.5).to(DEV)
    w0, b0 = to_panel(rnd(H, E, seed=42, scale=1 / math.sqrt(E))), rnd(H, seed=43, scale=0.1).to(DEV)
    w2, b2 = to_panel(rnd(H, H, seed=44, scale=1 / math.sqrt(H))), rnd(H, seed=45, scale=0.1).to(DEV)
    pos, emb = rnd(L, H, seed=46, scale=0.5).to(DEV), rnd(B, H, seed=47, scale=0.5).to(DEV)
    g, bt = (1 + rnd(H, seed=48, scale=0.2)).to(DEV), rnd(H, seed=49, scale=0.2).to(DEV)
    d0, db0 = to_panel(rnd(H, H, seed=50, scale=1 / math.sqrt(H))), rnd(H, seed=51, scale=0.1).to(DEV)
    d2, db2 = to_panel(rnd(E, H, seed=52, scale=1 / math.sqrt(H))), rnd(E, seed=53, scale=0.1).to(DEV)
    table = rnd(V, E, seed=54).to(DEV)
    buf = torch.empty(int(lib().mh_round_split_bytes(E, V)), dtype=torch.uint8, device=DEV)
    check(lib().mh_round_split_table(table.data_ptr(), None, V, E, buf.data_ptr(), current_stream()))
    X = torch.zeros(H // 32, N, 32, device=DEV, dtype=torch.bfloat16)
    out, idx = torch.zeros(N, E, device=DEV), torch.zeros(N, device=DEV, dtype=torch.int32)
    noise_a, noise_b = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16), torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
    side = torch.cuda.Stream()

    def head():
        check(lib().mh_up_proj_ln_fused(x.data_ptr(), E, E, w0.data_ptr(), b0.data_ptr(), w2.data_ptr(), b2.data_ptr(), pos.data_ptr(), emb.data_ptr(), None,
                                        g.data_ptr(), bt.data_ptr(), 1e-12, X.data_ptr(), N, B, L, H, current_stream()))

    def tail():
        check(lib().mh_down_proj_round_fused(X.data_ptr(), N, d0.data_ptr(), db0.data_ptr(), d2.data_ptr(), db2.data_ptr(), out.data_ptr(), None, buf.data_ptr(), V,
                                             idx.data_ptr(), None, N, E, H, current_stream()))
    head(); tail()
    torch.cuda.synchronize()
    X0, out0, idx0 = X.clone(), out.clone(), idx.clone()
    bad = 0
    for rep in range(120):
        with torch.cuda.stream(side):
            for _ in range(2):
                noise_a @ noise_b
        X.zero_(); out.zero_(); idx.zero_()
        head(); tail()
        torch.cuda.synchronize()
        bad += int(not (torch.equal(X, X0) and torch.equal(out, out0) and torch.equal(idx, idx0)))
    assert bad == 0, "%d of 120 repeated launches differ from the first" % bad


def test_concurrent_streams_really_overlap():
    """HIP multiplexes streams onto a few hardware queues: two streams on one queue run back to back (every fourth pool stream against a
    given one).  ops.concurrent_streams returns streams a probe has SEEN overlap - what the loop's decoupled chains run on."""
    from musediffusion_amd import ops
    ss = ops.concurrent_streams(2, torch.device(DEV))
    assert len(ss) == 2 and ss[0] != ss[1]
    assert sum(ops.streams_overlap(ss[0], ss[1]) for _ in range(5)) >= 3        # (hiccups of the timer are tolerated)
    pool = [torch.cuda.Stream() for _ in range(12)]
    clashes = sum(not ops.streams_overlap(pool[0], s) for s in pool[1:])
    print("pool streams that do NOT overlap with the first of 12: %d" % clashes)   # (3 on this runtime; 0 would make the probe moot, not wrong)
