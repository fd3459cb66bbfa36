"""Kernel-level parity tests (GPU): every C-ABI entry point against a plain torch fp32 CPU
reference of the same op (fp32 mode: tight tolerance; bf16 mode: inputs rounded to bf16 first,
tolerance = bf16 output rounding).  Integer / index outputs are compared exactly."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from musediffusion_amd import ops  # noqa: E402
from musediffusion_amd._lib import MH_BF16, MH_F32, lib  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def q(t, dtype):
    """Round a CPU fp32 tensor the way the device storage type does."""
    return t.bfloat16().float() if dtype == MH_BF16 else t


def to_dev(t, dtype):
    return t.to(DEV, ops.TORCH_DTYPE[dtype])


def assert_close(got, ref, atol, rtol=0.0, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax(err - tol))
        raise AssertionError("%s: %d/%d elements off, max err %.3e at flat %d (got %.6f ref %.6f), ref absmax %.3e"
                             % (what, int(bad.sum()), bad.numel(), float(err.max()), i,
                                float(got.flatten()[i]), float(ref.flatten()[i]), float(ref.abs().max())))


def test_library_sees_gfx950():
    import ctypes
    buf = ctypes.create_string_buffer(256)
    assert lib().mh_device_name(0, buf, 256) == 0
    assert b"gfx950" in buf.value, buf.value


def test_cast_pad_and_norms():
    x = rnd(37, 50, seed=1)
    for dt in (MH_F32, MH_BF16):
        out = ops.cast_pad(x.to(DEV), 64, dt, rows_out=40).float().cpu()
        ref = torch.zeros(40, 64)
        ref[:37, :50] = q(x, dt)
        assert torch.equal(out, ref)
    tbl = rnd(729, 128, seed=2)
    assert_close(ops.row_sqnorm(tbl.to(DEV)), (tbl ** 2).sum(-1), 1e-4, what="row_sqnorm")


def test_embed_gather_exact():
    tbl = rnd(729, 32, seed=3)
    ids = torch.randint(0, 729, (4, 24), generator=torch.Generator().manual_seed(4))
    for idt in (torch.int32, torch.int64):
        out = ops.embed_gather(tbl.to(DEV), ids.to(DEV, idt)).cpu()
        assert torch.equal(out, tbl[ids])


@pytest.mark.parametrize("dim", [128, 32, 33])
def test_timestep_embedding(dim):
    from oracle.denoiser import timestep_embedding
    t = torch.tensor([0.0, 0.5, 3.0, 499.5, 999.5, 12.25])
    ref = timestep_embedding(t, dim)
    out = ops.timestep_embedding(t.to(DEV), dim, MH_F32, ld_out=ops.pad64(dim)).cpu()
    assert_close(out[:, :dim], ref, 2e-4, what="timestep_embedding")  # |t f| up to 1e3: cos/sin arg ulp ~6e-5
    assert torch.all(out[:, dim:] == 0)


GEMM_CASES = [
    # M, N, K, act, residual, out_f32
    (200, 128, 64, None, False, False),
    (128, 192, 128, "tanh", False, False),
    (300, 64, 256, "gelu", False, False),
    (77, 320, 64, "silu", False, False),
    (256, 128, 512, None, True, False),
    (130, 729, 128, None, False, True),     # ragged N, fp32 out with odd leading dimension
    (64, 32, 64, None, False, True),        # N smaller than a tile (down-projection to E)
    (1000, 512, 2048, None, True, False),
]


@pytest.mark.parametrize("dtype", [MH_F32, MH_BF16], ids=["f32", "bf16"])
@pytest.mark.parametrize("glds", [0, 2, 4])
@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm_bias_act(case, dtype, glds):
    M, N, K, act, use_res, out_f32 = case
    if dtype == MH_F32 and glds:
        pytest.skip("kernel variants exist for bf16 only")
    # the default kernel choice (2) is tested in the production library; the alternatives need the debug build's switch
    import contextlib
    from musediffusion_amd import _lib as L_
    with (L_.debug_library() if glds != 2 else contextlib.nullcontext()):
        _gemm_bias_act_case(case, dtype, glds)


def _gemm_bias_act_case(case, dtype, glds):
    M, N, K, act, use_res, out_f32 = case
    if glds != 2:
        lib().mh_gemm_set_variant(glds)
    try:
        A = rnd(M, K, seed=10, scale=0.5)
        W = rnd(N, K, seed=11, scale=1.0 / math.sqrt(K))
        b = rnd(N, seed=12, scale=0.1)
        R = rnd(M, N, seed=13) if use_res else None
        ref = q(A, dtype) @ q(W, dtype).T + b
        ref = {None: lambda v: v, "tanh": torch.tanh, "gelu": torch.nn.functional.gelu,
               "silu": torch.nn.functional.silu}[act](ref)
        if use_res:
            ref = ref + q(R, dtype)
        out = ops.gemm_bias_act(to_dev(A, dtype), to_dev(W, dtype), b.to(DEV), to_dev(R, dtype) if use_res else None,
                                act, dtype, out_f32=out_f32)
        atol = 2e-5 * max(1.0, math.sqrt(K) / 8) if dtype == MH_F32 else (2e-5 if out_f32 else 0.0)
        rtol = 1e-5 if dtype == MH_F32 else (1e-4 if out_f32 else 2 ** -8)
        if dtype == MH_BF16:
            atol += 2e-3  # fp32 accumulation-order noise on top of the output rounding
        assert_close(out, ref, atol, rtol, what="gemm %s" % (case,))
    finally:
        if glds != 2:
            lib().mh_gemm_set_variant(2)


@pytest.mark.parametrize("dtype", [MH_F32, MH_BF16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 16, 64, 4), (3, 136, 128, 4), (2, 128, 512, 8)])
def test_gemm_qkv_layout(shape, dtype):
    B, L, H, nh = shape
    dh = H // nh
    X = rnd(B * L, H, seed=20, scale=0.5)
    W = rnd(3 * H, H, seed=21, scale=1.0 / math.sqrt(H))
    b = rnd(3 * H, seed=22, scale=0.1)
    ref = q(X, dtype) @ q(W, dtype).T + b
    rq, rk, rv = (ref[:, i * H:(i + 1) * H].view(B, L, nh, dh).permute(0, 2, 1, 3) for i in range(3))
    qd, kd, vt = ops.gemm_qkv(to_dev(X, dtype), to_dev(W, dtype), b.to(DEV), B, L, nh, dtype)
    atol, rtol = (5e-5, 1e-5) if dtype == MH_F32 else (2e-3, 2 ** -8)
    assert_close(qd, rq, atol, rtol, what="q")
    assert_close(kd, rk, atol, rtol, what="k")
    assert_close(vt, rv.transpose(-1, -2), atol, rtol, what="v^T")


def _attn_ref(qh, kh, vh, scale):
    s = (qh @ kh.transpose(-1, -2)) * scale
    return torch.softmax(s, -1) @ vh


@pytest.mark.parametrize("dtype,dh", [(MH_F32, 16), (MH_F32, 32), (MH_F32, 64), (MH_F32, 128),
                                      (MH_BF16, 32), (MH_BF16, 64), (MH_BF16, 128)])
@pytest.mark.parametrize("L", [64, 136, 512])
def test_attention(dtype, dh, L):
    B, nh = 2, 3
    qh, kh, vh = (rnd(B, nh, L, dh, seed=30 + i) for i in range(3))
    kh = kh * 1.5
    kh[:, :, 5] *= 4.0  # a dominant key: exercises the running-max rescale
    scale = 1.0 / math.sqrt(dh)
    ref = _attn_ref(q(qh, dtype), q(kh, dtype), q(vh, dtype), scale).permute(0, 2, 1, 3).reshape(B * L, nh * dh)
    vt = vh.transpose(-1, -2).contiguous()
    vt_dev = torch.zeros(vt.numel() + 128, device=DEV, dtype=ops.TORCH_DTYPE[dtype])
    vt_dev[: vt.numel()] = to_dev(vt, dtype).flatten()
    out = ops.attention(to_dev(qh, dtype), to_dev(kh, dtype), vt_dev, scale, dtype)
    atol = 2e-5 if dtype == MH_F32 else 2e-2   # bf16: P is rounded to bf16 before P.V
    assert_close(out, ref, atol, what="attention dh=%d L=%d" % (dh, L))


def _vt_perm(vt):
    """[..., dh, L] -> keys of every group of 16 stored as 0-3, 8-11, 4-7, 12-15 (mh_gemm_qkv_vtperm's order)."""
    *lead, L = vt.shape
    return vt.reshape(*lead, L // 16, 4, 4)[..., [0, 2, 1, 3], :].reshape(*lead, L).contiguous()


@pytest.mark.parametrize("dh,L,B,nh", [(64, 512, 3, 5), (32, 512, 2, 3), (64, 1024, 2, 2), (64, 768, 1, 3), (64, 2096, 1, 2),
                                        (64, 528, 2, 3), (32, 1040, 1, 2)])
def test_attention_stream(dh, L, B, nh):
    """streaming kernel (LDS-DMA stages, permuted V^T) == softmax(QK^T/sqrt(dh)) V, row-major and panel output"""
    from musediffusion_amd._lib import check, current_stream
    qh, kh, vh = (rnd(B, nh, L, dh, seed=230 + i) for i in range(3))
    kh = kh * 1.5
    kh[:, :, 300] *= 4.0
    kh[:, :, L - 3] *= 3.0          # a strong key inside the (possibly partial) last tile
    scale = 1.0 / math.sqrt(dh)
    ref = _attn_ref(q(qh, MH_BF16), q(kh, MH_BF16), q(vh, MH_BF16), scale).permute(0, 2, 1, 3).reshape(B * L, nh * dh)
    assert lib().mh_attention_stream_supported(L, dh) == 1 and lib().mh_attention_stream_supported(136, dh) == 0
    vtp = _vt_perm(vh.transpose(-1, -2).contiguous())
    vt_dev = torch.zeros(vtp.numel() + 128, device=DEV, dtype=torch.bfloat16)
    vt_dev[: vtp.numel()] = vtp.to(DEV).bfloat16().flatten()
    qd, kd = qh.to(DEV).bfloat16().contiguous(), kh.to(DEV).bfloat16().contiguous()
    out = torch.zeros(B * L, nh * dh, device=DEV, dtype=torch.bfloat16)
    check(lib().mh_attention_stream_fwd(qd.data_ptr(), kd.data_ptr(), vt_dev.data_ptr(), out.data_ptr(), nh * dh, 0, B, L, nh, dh,
                                        scale, current_stream()))
    assert_close(out, ref, 2e-2, what="attention_stream dh=%d L=%d" % (dh, L))
    from musediffusion_amd import _lib as L_
    with L_.debug_library():      # (the geometry switch lives in the debug build; everything else here runs the production library)
        lib().mh_attention_set_stream(2)          # 8-wave / 128-key-stage blocks: same arithmetic per query tile, same bits
        try:
            out2 = torch.zeros_like(out)
            check(lib().mh_attention_stream_fwd(qd.data_ptr(), kd.data_ptr(), vt_dev.data_ptr(), out2.data_ptr(), nh * dh, 0, B, L, nh, dh,
                                                scale, current_stream()))
            assert torch.equal(out2, out)
        finally:
            lib().mh_attention_set_stream(1)
    outp = torch.zeros(nh * dh // 32, B * L, 32, device=DEV, dtype=torch.bfloat16)
    check(lib().mh_attention_stream_fwd(qd.data_ptr(), kd.data_ptr(), vt_dev.data_ptr(), outp.data_ptr(), B * L, 1, B, L, nh, dh,
                                        scale, current_stream()))
    assert torch.equal(outp.permute(1, 0, 2).reshape(B * L, nh * dh), out)


@pytest.mark.parametrize("dh,L,B,nh", [(64, 512, 3, 5), (32, 512, 2, 3), (64, 1024, 2, 2), (64, 768, 1, 3)])
def test_attention_stream_prescaled(dh, L, B, nh):
    """queries that carry scale x log2(e) (the QKV epilogue's form): the reference-in-the-accumulator softmax == softmax(QK^T/sqrt(dh)) V.
    Forces the re-basing branch: one key row per head beats its predecessors by far more than 2^8, late in the sequence, and one (batch,
    head) has every score far below zero (the first sub-tile's reference is then a large negative number, not 0)."""
    from musediffusion_amd._lib import check, current_stream
    qh, kh, vh = (rnd(B, nh, L, dh, seed=430 + i) for i in range(3))
    kh = kh * 1.5
    kh[:, :, 300] *= 6.0
    kh[:, :, L - 3] = 40.0 * qh[:, :, 17] / qh[:, :, 17].norm(dim=-1, keepdim=True)   # aligned with query 17: its score jumps by > 2^8
    qh[0, 0] = qh[0, 0].abs() + 2.0                                                       # (batch 0, head 0): all scores << 0
    kh[0, 0] = -(kh[0, 0].abs() + 2.0)
    scale = 1.0 / math.sqrt(dh)
    sl2 = scale * math.log2(math.e)
    assert lib().mh_attention_stream_prescaled_supported(L, dh) == (1 if L % 256 == 0 else 0)
    q_pre = (q(qh, MH_BF16) * sl2).bfloat16()                                             # what mh_gemm_qkv_vtperm_qs stores
    ref = _attn_ref(q_pre.float() / sl2, q(kh, MH_BF16), q(vh, MH_BF16), scale).permute(0, 2, 1, 3).reshape(B * L, nh * dh)
    vtp = _vt_perm(vh.transpose(-1, -2).contiguous())
    vt_dev = torch.zeros(vtp.numel() + 128, device=DEV, dtype=torch.bfloat16)
    vt_dev[: vtp.numel()] = vtp.to(DEV).bfloat16().flatten()
    qd, kd = q_pre.to(DEV).contiguous(), kh.to(DEV).bfloat16().contiguous()
    out = torch.zeros(B * L, nh * dh, device=DEV, dtype=torch.bfloat16)
    check(lib().mh_attention_stream_fwd_prescaled(qd.data_ptr(), kd.data_ptr(), vt_dev.data_ptr(), out.data_ptr(), nh * dh, 0, B, L, nh, dh,
                                                  current_stream()))
    assert torch.isfinite(out).all()
    assert_close(out, ref, 2e-2, what="attention_stream_prescaled dh=%d L=%d" % (dh, L))
    outp = torch.zeros(nh * dh // 32, B * L, 32, device=DEV, dtype=torch.bfloat16)
    check(lib().mh_attention_stream_fwd_prescaled(qd.data_ptr(), kd.data_ptr(), vt_dev.data_ptr(), outp.data_ptr(), B * L, 1, B, L, nh, dh,
                                                  current_stream()))
    assert torch.equal(outp.permute(1, 0, 2).reshape(B * L, nh * dh), out)


def test_gemm_qkv_vtperm_scaled_queries():
    """mh_gemm_qkv_vtperm_qs: q = (x Wq^T + bq) * q_scale rounded ONCE from the fp32 accumulator; k and V^T as mh_gemm_qkv_vtperm"""
    from musediffusion_amd._lib import check, current_stream
    B, L, H, nh = 2, 256, 128, 2
    X, Wq, bq = rnd(B * L, H, seed=440), rnd(3 * H, H, seed=441, scale=1.0 / 11), rnd(3 * H, seed=442, scale=0.1)
    Xp, Wp, bd = dev_panel(X), dev_panel(Wq), bq.to(DEV)
    q0, k0 = (torch.empty(B, nh, L, H // nh, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    q1, k1 = torch.empty_like(q0), torch.empty_like(k0)
    vt0, vt1 = (torch.zeros(B * H * L + 128, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    st = current_stream()
    check(lib().mh_gemm_qkv_vtperm(Xp.data_ptr(), B * L, 1, Wp.data_ptr(), 3 * H, 1, bd.data_ptr(), q0.data_ptr(), k0.data_ptr(), vt0.data_ptr(),
                                   B, L, H, nh, st))
    qs = 0.18033688
    check(lib().mh_gemm_qkv_vtperm_qs(Xp.data_ptr(), B * L, Wp.data_ptr(), 3 * H, bd.data_ptr(), q1.data_ptr(), k1.data_ptr(), vt1.data_ptr(),
                                      B, L, H, nh, qs, None, st))
    assert torch.equal(k0, k1) and torch.equal(vt0, vt1)
    acc = (q(X, MH_BF16) @ q(Wq[:H], MH_BF16).t() + bq[:H]) * qs                      # fp32 accumulator, then one rounding
    ref = acc.view(B, L, nh, H // nh).permute(0, 2, 1, 3)
    assert_close(q1, ref, 1e-3, 2 ** -8, what="scaled queries")
    assert not torch.equal(q1, q0)


@pytest.mark.parametrize("dh,L,B,nh", [(64, 512, 2, 3), (64, 1024, 1, 2), (32, 512, 2, 2), (64, 528, 1, 2), (64, 2096, 1, 1), (32, 784, 1, 2)])
def test_attention_stream_backward(dh, L, B, nh):
    """fused backward kernels (dQ; dK, dV) == torch autograd through softmax(QK^T/sqrt(dh)) V on the bf16-rounded inputs"""
    from musediffusion_amd._lib import check, current_stream
    H = nh * dh
    qkv = rnd(B * L, 3 * H, seed=260, scale=0.8)
    qkv[:, H + 7] *= 3.0
    dctx = rnd(B * L, H, seed=261, scale=0.5)
    scale = 1.0 / math.sqrt(dh)
    qb = q(qkv, MH_BF16).view(B, L, 3, nh, dh).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(True)   # [3, B, nh, L, dh]
    out_ref = torch.softmax((qb[0] @ qb[1].transpose(-1, -2)) * scale, -1) @ qb[2]
    dO_ref = q(dctx, MH_BF16).view(B, L, nh, dh).permute(0, 2, 1, 3)
    out_ref.backward(dO_ref)
    gref = qb.grad                                                                                          # [3, B, nh, L, dh]
    qkv_d, dctx_d = qkv.to(DEV).bfloat16().contiguous(), dctx.to(DEV).bfloat16().contiguous()
    st = current_stream()

    def perm(src, col0, ld, mode):
        shape = (B, nh, dh, L) if mode >= 2 else (B, nh, L, dh)
        out = torch.zeros(B * nh * L * dh + 256, device=DEV, dtype=torch.bfloat16)
        check(lib().mh_head_permute(src.data_ptr() + col0 * 2, out.data_ptr(), ld, B, L, nh, dh, mode, MH_BF16, st))
        return out
    qr, kr, vr = (perm(qkv_d, i * H, 3 * H, 0) for i in range(3))
    qT, kT, vT = (perm(qkv_d, i * H, 3 * H, 3) for i in range(3))
    if L % 64 == 0:      # mode 4 = mode 3 + the 256-element slack behind the last row zeroed by the same launch
        dirty = torch.full((B * nh * L * dh + 256,), 7.0, device=DEV, dtype=torch.bfloat16)
        check(lib().mh_head_permute(qkv_d.data_ptr() + H * 2, dirty.data_ptr(), 3 * H, B, L, nh, dh, 4, MH_BF16, st))
        assert torch.equal(dirty, kT)
    assert torch.equal(vT[: B * H * L].view(B, nh, dh, L), _vt_perm(vr[: B * H * L].view(B, nh, L, dh).transpose(-1, -2).contiguous()))
    ctx = torch.zeros(B * L, H, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B * nh * L, device=DEV)
    check(lib().mh_attention_stream_fwd_lse(qr.data_ptr(), kr.data_ptr(), vT.data_ptr(), ctx.data_ptr(), H, 0, B, L, nh, dh, scale,
                                            lse.data_ptr(), st))
    s_ref = (qb[0].detach() @ qb[1].detach().transpose(-1, -2)) * scale
    assert_close(lse.view(B, nh, L).cpu() / math.log2(math.e), torch.logsumexp(s_ref, -1), 2e-3, what="lse")
    assert_close(ctx, out_ref.detach().permute(0, 2, 1, 3).reshape(B * L, H), 2e-2, what="ctx")
    dOr, dOT = perm(dctx_d, 0, H, 0), perm(dctx_d, 0, H, 3)
    D = torch.zeros(B * nh * L, device=DEV)
    check(lib().mh_attention_bwd_rowdot(dctx_d.data_ptr(), ctx.data_ptr(), H, D.data_ptr(), B, L, nh, dh, st))
    D_ref = (dO_ref * out_ref.detach()).sum(-1)
    assert_close(D.view(B, nh, L), D_ref, 2e-2, 2e-2, what="D")
    dqkv = torch.zeros(B * L, 3 * H, device=DEV, dtype=torch.bfloat16)
    ctx_rows = perm(ctx, 0, H, 0)
    D2 = torch.zeros(B * nh * L, device=DEV)
    check(lib().mh_attention_stream_bwd(qr.data_ptr(), kr.data_ptr(), vr.data_ptr(), qT.data_ptr(), kT.data_ptr(), dOr.data_ptr(),
                                        dOT.data_ptr(), ctx_rows.data_ptr(), lse.data_ptr(), D2.data_ptr(), dqkv.data_ptr(),
                                        dqkv.data_ptr() + H * 2, dqkv.data_ptr() + 2 * H * 2, 3 * H, B, L, nh, dh, scale, st))
    assert_close(D2.view(B, nh, L), D_ref, 2e-2, 2e-2, what="D from the dQ kernel")
    got = dqkv.float().cpu().view(B, L, 3, nh, dh).permute(2, 0, 3, 1, 4)
    for i, nm in enumerate(("dQ", "dK", "dV")):
        ref = gref[i]
        err = (got[i] - ref).abs().max().item()
        tol = 2e-2 * ref.abs().max().item() + 1e-3
        assert err <= tol, "%s: max err %.4g > %.4g (ref max %.3g)" % (nm, err, tol, ref.abs().max().item())


@pytest.mark.parametrize("K,M,N", [(4096, 512, 512), (2048, 264, 136), (8192, 2048, 512), (1024, 72, 64)])
def test_gemm_dw_k_major(K, M, N):
    """dW = dY^T X straight from the k-major operands (transposing LDS reads) == fp32 matmul of the bf16-rounded inputs"""
    from musediffusion_amd._lib import check, current_stream
    A, B = rnd(K, M, seed=270, scale=0.5), rnd(K, N, seed=271, scale=0.5)
    ref = q(A, MH_BF16).T @ q(B, MH_BF16)
    Ad, Bd = A.to(DEV).bfloat16().contiguous(), B.to(DEV).bfloat16().contiguous()
    S = int(lib().mh_gemm_dw_splits(K, M, N))
    part = torch.zeros(S, M, N, device=DEV)
    check(lib().mh_gemm_dw(Ad.data_ptr(), M, Bd.data_ptr(), N, part.data_ptr(), S, K, M, N, current_stream()))
    out = torch.empty(M, N, device=DEV)
    check(lib().mh_sum_slices(part.data_ptr(), S, M * N, out.data_ptr(), current_stream()))
    assert_close(out, ref, 2e-3 * math.sqrt(K / 1024), 1e-3, what="gemm_dw K=%d M=%d N=%d splits=%d" % (K, M, N, S))


@pytest.mark.parametrize("K,M,N", [(4096, 512, 512), (2048, 264, 136), (8192, 2048, 512), (8192, 512, 2048), (1024, 72, 64), (2048, 128, 384)])
def test_gemm_dw_with_bias_gradient(K, M, N):
    """the fused form: each split slice carries dW and, behind it, the column sums of dY (all-ones MFMA); one mh_sum_slices folds both"""
    from musediffusion_amd._lib import check, current_stream
    A, B = rnd(K, M, seed=272, scale=0.5), rnd(K, N, seed=273, scale=0.5)
    ref, ref_b = q(A, MH_BF16).T @ q(B, MH_BF16), q(A, MH_BF16).sum(0)
    Ad, Bd = A.to(DEV).bfloat16().contiguous(), B.to(DEV).bfloat16().contiguous()
    S = int(lib().mh_gemm_dw_splits(K, M, N))
    n = M * N + M
    part = torch.full((S, n), float("nan"), device=DEV)
    check(lib().mh_gemm_dw_bias(Ad.data_ptr(), M, Bd.data_ptr(), N, part.data_ptr(), S, K, M, N, 1, current_stream()))
    out = torch.empty(n, device=DEV)
    check(lib().mh_sum_slices(part.data_ptr(), S, n, out.data_ptr(), current_stream()))
    assert_close(out[:M * N].view(M, N), ref, 2e-3 * math.sqrt(K / 1024), 1e-3, what="gemm_dw_bias dW K=%d M=%d N=%d splits=%d" % (K, M, N, S))
    assert_close(out[M * N:], ref_b, 2e-3 * math.sqrt(K / 1024), 1e-3, what="gemm_dw_bias db K=%d M=%d N=%d" % (K, M, N))


def test_weight_prep_one_launch_copies_and_transposes():
    """mh_weight_prep: bf16 copies + transposes of several fp32 matrices from one device table, incl. the QKV case (three sources
    into row / column blocks of one destination): bit-identical with torch's casts"""
    import ctypes as C
    from musediffusion_amd import _lib
    from musediffusion_amd._lib import check, current_stream
    H, F = 128, 256
    Ws = [rnd(H, H, seed=300 + i).to(DEV) for i in range(3)] + [rnd(F, H, seed=310).to(DEV), rnd(H, F, seed=311).to(DEV)]
    qkv = torch.full((3 * H, H), 9.0, device=DEV, dtype=torch.bfloat16)
    qkv_t = torch.full((H, 3 * H), 9.0, device=DEV, dtype=torch.bfloat16)
    w1, w1_t = torch.empty(F, H, device=DEV, dtype=torch.bfloat16), torch.empty(H, F, device=DEV, dtype=torch.bfloat16)
    w2 = torch.empty(H, F, device=DEV, dtype=torch.bfloat16)
    items, tiles = [], 0
    for W, dst, dst_t, ld_dst, ld_t in ([(Ws[j], qkv.data_ptr() + j * H * H * 2, qkv_t.data_ptr() + j * H * 2, H, 3 * H) for j in range(3)]
                                        + [(Ws[3], w1.data_ptr(), w1_t.data_ptr(), H, F), (Ws[4], w2.data_ptr(), None, F, 0)]):
        items.append(_lib.WPrepItem(W.data_ptr(), dst, dst_t, W.shape[0], W.shape[1], ld_dst, ld_t, tiles, 0))
        tiles += (W.shape[0] // 64) * (W.shape[1] // 64)
    raw = (_lib.WPrepItem * len(items))(*items)
    table = torch.frombuffer(bytearray(C.string_at(C.addressof(raw), C.sizeof(raw))), dtype=torch.uint8).to(DEV)
    check(lib().mh_weight_prep(table.data_ptr(), len(items), tiles, current_stream()))
    cat = torch.cat(Ws[:3], 0).bfloat16()
    assert torch.equal(qkv, cat) and torch.equal(qkv_t, cat.t().contiguous())
    assert torch.equal(w1, Ws[3].bfloat16()) and torch.equal(w1_t, Ws[3].bfloat16().t().contiguous())
    assert torch.equal(w2, Ws[4].bfloat16())


def test_gemm_qkv_vtperm():
    from musediffusion_amd._lib import check, current_stream
    B, L, H, nh = 2, 48, 128, 2
    X, Wq, bq = rnd(B * L, H, seed=240), rnd(3 * H, H, seed=241, scale=1.0 / 11), rnd(3 * H, seed=242, scale=0.1)
    Xd, Wd, bd = X.to(DEV).bfloat16(), Wq.to(DEV).bfloat16(), bq.to(DEV)
    q0, k0, vt0 = ops.gemm_qkv(Xd, Wd, bd, B, L, nh, MH_BF16)
    q1, k1 = torch.empty_like(q0), torch.empty_like(k0)
    vt1 = torch.zeros(B * H * L + 128, device=DEV, dtype=torch.bfloat16)
    check(lib().mh_gemm_qkv_vtperm(Xd.data_ptr(), H, 0, Wd.data_ptr(), H, 0, bd.data_ptr(), q1.data_ptr(), k1.data_ptr(),
                                   vt1.data_ptr(), B, L, H, nh, current_stream()))
    assert torch.equal(q0, q1) and torch.equal(k0, k1)
    assert torch.equal(vt1[: B * H * L].view(B, nh, H // nh, L), _vt_perm(vt0))


@pytest.mark.parametrize("dtype", [MH_F32, MH_BF16], ids=["f32", "bf16"])
@pytest.mark.parametrize("H", [64, 512, 768])
def test_layernorm(H, dtype):
    x = rnd(50, H, seed=40, scale=2.0) + 0.3
    g, b = 1 + 0.1 * rnd(H, seed=41), 0.1 * rnd(H, seed=42)
    ref = torch.nn.functional.layer_norm(q(x, dtype), (H,), g, b, 1e-12)
    out = ops.layernorm(to_dev(x, dtype), g.to(DEV), b.to(DEV), 1e-12, dtype)
    assert_close(out, ref, 5e-6 if dtype == MH_F32 else 1e-3, 1e-6 if dtype == MH_F32 else 2 ** -8, what="layernorm")


@pytest.mark.parametrize("dtype", [MH_F32, MH_BF16], ids=["f32", "bf16"])
@pytest.mark.parametrize("from_latent", [False, True])
def test_add_pos_time_layernorm(dtype, from_latent):
    B, L, H = 3, 24, 128
    x = rnd(B * L, H, seed=43)
    pos, emb = rnd(L, H, seed=44), rnd(7, H, seed=45)
    rows = torch.tensor([6, 0, 3], dtype=torch.int32)
    g, b = 1 + 0.1 * rnd(H, seed=46), 0.1 * rnd(H, seed=47)
    xin = x if from_latent else q(x, dtype)
    pre = (pos[None] + xin.view(B, L, H)) + emb[rows.long()][:, None]
    ref = torch.nn.functional.layer_norm(pre, (H,), g, b, 1e-12).view(B * L, H)
    xd = x.to(DEV) if from_latent else to_dev(x, dtype)
    out = ops.add_pos_time_layernorm(xd, pos.to(DEV), emb.to(DEV), rows.to(DEV), g.to(DEV), b.to(DEV), B, L, 1e-12, dtype)
    assert_close(out, ref, 5e-6 if dtype == MH_F32 else 1e-3, 1e-6 if dtype == MH_F32 else 2 ** -8, what="add_pos_time_ln")
    out2 = ops.add_pos_time_layernorm(xd, pos.to(DEV), emb[rows.long()].contiguous().to(DEV), None, g.to(DEV), b.to(DEV), B, L, 1e-12, dtype)
    assert torch.equal(out2, out)


@pytest.mark.parametrize("V,E", [(729, 128), (729, 32), (97, 64), (729, 500)])
def test_rounding_and_logits_exact(V, E):
    from oracle import sampling as osa
    tbl = rnd(V, E, seed=50, scale=0.5)
    ids = torch.randint(0, V, (700,), generator=torch.Generator().manual_seed(51))
    x = tbl[ids] + 0.3 * rnd(700, E, seed=52)
    ref_idx = osa.nearest_token(tbl, x)
    for mfma in (True, False):
        got = ops.round_to_embedding(x.to(DEV), tbl.to(DEV), mfma=mfma).cpu().long()
        assert torch.equal(got, ref_idx), "rounding(mfma=%s): %d mismatches" % (mfma, int((got != ref_idx).sum()))
    bias = rnd(V, seed=53, scale=0.1)
    ref_tok = torch.argmax(x @ tbl.T + bias, dim=-1)
    got = ops.logits_argmax(x.to(DEV), tbl.to(DEV), bias.to(DEV)).cpu().long()
    assert torch.equal(got, ref_tok), "logits argmax: %d mismatches" % int((got != ref_tok).sum())


def test_rounding_ties_pick_first_index():
    tbl = rnd(40, 16, seed=54)
    tbl[17] = tbl[3]          # duplicate row: distance ties exactly
    tbl[30] = tbl[3]
    x = tbl[[3, 17, 30, 5]].clone()
    for mfma in (True, False):
        got = ops.round_to_embedding(x.to(DEV), tbl.to(DEV), mfma=mfma).cpu().tolist()
        assert got == [3, 3, 3, 5], (mfma, got)
    bias = torch.zeros(40)
    got = ops.logits_argmax(x.to(DEV), tbl.to(DEV), bias.to(DEV)).cpu()
    ref = torch.argmax(x @ tbl.T, -1)
    assert torch.equal(got.long(), ref)


def test_q_sample_bit_exact():
    from oracle import sampling as osa, schedule as osc
    d = osc.make_diffusion()
    B, L, E = 4, 16, 32
    x0, nz = rnd(B, L, E, seed=60), rnd(B, L, E, seed=61)
    t = torch.tensor([0, 5, 1000, 1999])
    mask = (torch.arange(L)[None] >= torch.tensor([3, 0, 8, 16])[:, None]).long()
    ref = osa.q_sample(d, x0, t, noise=nz, mask=mask)
    a = torch.tensor(d.sqrt_alphas_cumprod, dtype=torch.float)[t]
    s = torch.tensor(d.sqrt_one_minus_alphas_cumprod, dtype=torch.float)[t]
    out = ops.q_sample(x0.to(DEV), nz.to(DEV), a.to(DEV), s.to(DEV), mask.to(DEV)).cpu()
    assert torch.equal(out, ref)
    mask3 = torch.broadcast_to(mask.unsqueeze(-1), x0.shape)
    out = ops.q_sample(x0.to(DEV), nz.to(DEV), a.to(DEV), s.to(DEV), mask3.contiguous().to(DEV)).cpu()
    assert torch.equal(out, ref)
    out = ops.q_sample(x0.to(DEV), nz.to(DEV), a.to(DEV), s.to(DEV), None).cpu()
    assert torch.equal(out, osa.q_sample(d, x0, t, noise=nz))


def test_trunc_normal_statistics_and_determinism():
    n = 1 << 20
    a = ops.trunc_normal((n,), 1.0, seed=105, stream_id=3)
    b = ops.trunc_normal((n,), 1.0, seed=105, stream_id=3)
    c = ops.trunc_normal((n,), 1.0, seed=106, stream_id=3)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert float(a.abs().max()) <= 1.0
    assert abs(float(a.mean())) < 3e-3
    assert abs(float(a.var()) - 0.29112) < 3e-3      # Var[z | |z|<=1] = 1 - 2 phi(1) / (2 Phi(1) - 1)
    u = ops.trunc_normal((n,), 0.0, seed=7)
    assert abs(float(u.mean())) < 4e-3 and abs(float(u.var()) - 1.0) < 6e-3
    assert abs(float((u.abs() > 1).float().mean()) - 0.3173) < 3e-3
    ctr = torch.tensor([5], dtype=torch.int32, device=DEV)
    s5 = ops.trunc_normal((4096,), 1.0, seed=1, step_counter=ctr)
    ctr += 1
    s6 = ops.trunc_normal((4096,), 1.0, seed=1, step_counter=ctr)
    assert not torch.equal(s5, s6)


# ------------------------------------------------------------------ K32-panel layout kernels
def to_panel(t, ld_rows=None, cols_pad=None):
    """CPU reference of the panel image: [cols_pad/32][ld_rows][32]."""
    rows, cols = t.shape
    ld_rows = rows if ld_rows is None else ld_rows
    cols_pad = (cols + 31) // 32 * 32 if cols_pad is None else cols_pad
    full = torch.zeros(ld_rows, cols_pad)
    full[:rows, :cols] = t
    return full.view(ld_rows, cols_pad // 32, 32).permute(1, 0, 2).contiguous()


def from_panel(p, rows, cols):
    kb, ld, _ = p.shape
    return p.permute(1, 0, 2).reshape(ld, kb * 32)[:rows, :cols]


def dev_panel(t, **kw):
    rows, cols = t.shape
    cols_pad = kw.get("cols_pad", (cols + 31) // 32 * 32)
    ld_rows = kw.get("ld_rows", rows)
    out = torch.empty(cols_pad // 32, ld_rows, 32, device=DEV, dtype=torch.bfloat16)
    x = t.to(DEV).contiguous()
    from musediffusion_amd._lib import check, current_stream
    check(lib().mh_pack_panel(x.data_ptr(), cols, out.data_ptr(), ld_rows, rows, cols, cols_pad, current_stream()))
    return out


def test_pack_unpack_panel():
    x = rnd(37, 50, seed=70)
    p = dev_panel(x, ld_rows=40, cols_pad=64)
    assert torch.equal(p.float().cpu(), to_panel(x.bfloat16().float(), 40, 64))
    back = torch.empty(37, 50, device=DEV)
    from musediffusion_amd._lib import check, current_stream
    check(lib().mh_unpack_panel_f32(p.data_ptr(), 40, back.data_ptr(), 50, 37, 50, current_stream()))
    assert torch.equal(back.cpu(), x.bfloat16().float())


@pytest.mark.parametrize("case", [(300, 256, 128, "gelu", False), (1000, 512, 2048, None, True), (130, 128, 64, "tanh", False),
                                  (77, 64, 512, None, True)])
def test_gemm_panel_layouts(case):
    from musediffusion_amd._lib import check, current_stream
    M, N, K, act, use_res = case
    A, W = rnd(M, K, seed=71, scale=0.5), rnd(N, K, seed=72, scale=1.0 / math.sqrt(K))
    b, R = rnd(N, seed=73, scale=0.1), rnd(M, N, seed=74)
    ref = q(A, MH_BF16) @ q(W, MH_BF16).T + b
    ref = {None: lambda v: v, "tanh": torch.tanh, "gelu": torch.nn.functional.gelu}[act](ref)
    if use_res:
        ref = ref + q(R, MH_BF16)
    Ap, Wp, Rp = dev_panel(A), dev_panel(W), dev_panel(R)
    outp = torch.zeros(N // 32, M, 32, device=DEV, dtype=torch.bfloat16)
    bd = b.to(DEV)
    check(lib().mh_gemm_bias_act_ex(Ap.data_ptr(), M, 1, Wp.data_ptr(), N, 1, bd.data_ptr(), Rp.data_ptr() if use_res else None,
                                    M, 1, outp.data_ptr(), M, 1, 0, M, N, K, ops.ACT[act], MH_BF16, current_stream()))
    assert_close(from_panel(outp.float().cpu(), M, N), ref, 2e-3, 2 ** -8, what="gemm panel %s" % (case,))
    # mixed: panel operands, fp32 row-major output (the down-projection's form)
    outf = torch.zeros(M, N, device=DEV)
    check(lib().mh_gemm_bias_act_ex(Ap.data_ptr(), M, 1, Wp.data_ptr(), N, 1, bd.data_ptr(), None, 0, 0, outf.data_ptr(), N, 0, 1,
                                    M, N, K, 0, MH_BF16, current_stream()))
    assert_close(outf, q(A, MH_BF16) @ q(W, MH_BF16).T + b, 2e-3, 1e-4, what="gemm panel->f32")


@pytest.mark.parametrize("variant", [2, 4])
def test_gemm_big_tile_variants(variant, dbg_lib):
    """both big-tile configurations (256x128 / 256x256) against the fp32 matmul of the bf16-rounded operands"""
    lib().mh_gemm_set_variant(variant)
    try:
        for (M, N, K, act, use_res) in [(700, 512, 256, "gelu", False), (513, 264, 96, None, True), (40, 1536, 512, None, False)]:
            A, W = rnd(M, K, seed=171, scale=0.5), rnd(N, K, seed=172, scale=1.0 / math.sqrt(K))
            b, R = rnd(N, seed=173, scale=0.1), rnd(M, N, seed=174)
            ref = q(A, MH_BF16) @ q(W, MH_BF16).T + b
            ref = {None: lambda v: v, "gelu": torch.nn.functional.gelu}[act](ref)
            if use_res:
                ref = ref + q(R, MH_BF16)
            out = ops.gemm_bias_act(A.to(DEV).bfloat16(), W.to(DEV).bfloat16(), b.to(DEV), R.to(DEV).bfloat16() if use_res else None,
                                    act, MH_BF16)
            assert_close(out.float(), ref, 2e-3, 2 ** -8, what="big tile variant %d %s" % (variant, (M, N, K)))
        # QKV scatter through the same tile
        B, L, H, nh = 3, 40, 256, 4
        X, Wq, bq = rnd(B * L, H, seed=175), rnd(3 * H, H, seed=176, scale=1.0 / 16), rnd(3 * H, seed=177, scale=0.1)
        qh, kh, vt = ops.gemm_qkv(X.to(DEV).bfloat16(), Wq.to(DEV).bfloat16(), bq.to(DEV), B, L, nh, MH_BF16)
        ref = (q(X, MH_BF16) @ q(Wq, MH_BF16).T + bq).view(B, L, 3, nh, H // nh)
        assert_close(qh.float(), ref[:, :, 0].permute(0, 2, 1, 3), 2e-3, 2 ** -8, what="qkv q variant %d" % variant)
        assert_close(kh.float(), ref[:, :, 1].permute(0, 2, 1, 3), 2e-3, 2 ** -8, what="qkv k variant %d" % variant)
        assert_close(vt.float(), ref[:, :, 2].permute(0, 2, 3, 1), 2e-3, 2 ** -8, what="qkv vt variant %d" % variant)
    finally:
        lib().mh_gemm_set_variant(2)


@pytest.mark.parametrize("N,rows64", [(128, 0), (256, 0), (512, 0), (512, 1)])
@pytest.mark.parametrize("panel", [0, 1])
def test_gemm_bias_residual_layernorm(N, panel, rows64):
    """dense + residual + LayerNorm in one kernel == the three separate steps (BertSelfOutput / BertOutput); rows64: the 64-row
    full-row tile (twice the blocks; selectable with the debug library's mh_gemm_set_plain_stores bit 16)"""
    from musediffusion_amd import _lib as L_
    if not rows64:
        return _gemm_bias_residual_layernorm(N, panel)
    with L_.debug_library():
        lib().mh_gemm_set_plain_stores(16)
        try:
            _gemm_bias_residual_layernorm(N, panel)
        finally:
            lib().mh_gemm_set_plain_stores(0)


def _gemm_bias_residual_layernorm(N, panel):
    from musediffusion_amd._lib import check, current_stream
    M, K = 333, 2 * N
    A, W = rnd(M, K, seed=181, scale=0.5), rnd(N, K, seed=182, scale=1.0 / math.sqrt(K))
    b, R = rnd(N, seed=183, scale=0.1), rnd(M, N, seed=184)
    g, bt = 1.0 + rnd(N, seed=185, scale=0.1), rnd(N, seed=186, scale=0.1)
    pre = q(A, MH_BF16) @ q(W, MH_BF16).T + b + q(R, MH_BF16)
    ref = torch.nn.functional.layer_norm(pre, (N,), g, bt, 1e-12)
    assert lib().mh_gemm_bias_res_ln_supported(N) == 1 and lib().mh_gemm_bias_res_ln_supported(768) == 0
    gd, btd, bd = g.to(DEV), bt.to(DEV), b.to(DEV)
    if panel:
        Ad, Wd, Rd = dev_panel(A), dev_panel(W), dev_panel(R)
        out = torch.zeros(N // 32, M, 32, device=DEV, dtype=torch.bfloat16)
        check(lib().mh_gemm_bias_res_ln(Ad.data_ptr(), M, 1, Wd.data_ptr(), N, 1, bd.data_ptr(), Rd.data_ptr(), M, 1, gd.data_ptr(),
                                        btd.data_ptr(), 1e-12, out.data_ptr(), M, 1, M, N, K, current_stream()))
        got = from_panel(out.float().cpu(), M, N)
    else:
        Ad, Wd, Rd = A.to(DEV).bfloat16(), W.to(DEV).bfloat16(), R.to(DEV).bfloat16()
        out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        check(lib().mh_gemm_bias_res_ln(Ad.data_ptr(), K, 0, Wd.data_ptr(), K, 0, bd.data_ptr(), Rd.data_ptr(), N, 0, gd.data_ptr(),
                                        btd.data_ptr(), 1e-12, out.data_ptr(), N, 0, M, N, K, current_stream()))
        got = out.float().cpu()
    # the fused path normalises the fp32 accumulator (the unfused one rounds the sum to bf16 first): tolerance is bf16 output rounding
    assert_close(got, ref, 1e-3, 2 ** -7, what="gemm+res+ln N=%d panel=%d" % (N, panel))


@pytest.mark.parametrize("H", [64, 128, 512, 768])
def test_layernorm_panel(H):
    from musediffusion_amd._lib import check, current_stream
    rows = 150
    x = rnd(rows, H, seed=75, scale=2.0) + 0.3
    g, b = 1 + 0.1 * rnd(H, seed=76), 0.1 * rnd(H, seed=77)
    ref = torch.nn.functional.layer_norm(q(x, MH_BF16), (H,), g, b, 1e-12)
    xp = dev_panel(x)
    outp = torch.zeros_like(xp)
    gd, bd = g.to(DEV), b.to(DEV)
    check(lib().mh_layernorm_panel(xp.data_ptr(), rows, gd.data_ptr(), bd.data_ptr(), outp.data_ptr(), rows, rows, H, 1e-12,
                                   current_stream()))
    assert_close(from_panel(outp.float().cpu(), rows, H), ref, 1e-3, 2 ** -8, what="ln panel")
    # add variant from the panel image and from the fp32 latent
    B, L = 5, 30
    pos, emb = rnd(L, H, seed=78), rnd(7, H, seed=79)
    idx = torch.tensor([6, 0, 3, 3, 1], dtype=torch.int32)
    for from_latent in (False, True):
        xin = x if from_latent else q(x, MH_BF16)
        pre = (pos[None] + xin.view(B, L, H)) + emb[idx.long()][:, None]
        ref2 = torch.nn.functional.layer_norm(pre, (H,), g, b, 1e-12).view(rows, H)
        src = x.to(DEV).contiguous() if from_latent else xp
        out2 = torch.zeros_like(xp)
        pd, ed, rd = pos.to(DEV), emb.to(DEV), idx.to(DEV)
        check(lib().mh_add_pos_time_layernorm_panel(src.data_ptr(), H if from_latent else rows, int(from_latent), pd.data_ptr(),
                                                    ed.data_ptr(), rd.data_ptr(), gd.data_ptr(), bd.data_ptr(), out2.data_ptr(),
                                                    rows, B, L, H, 1e-12, current_stream()))
        assert_close(from_panel(out2.float().cpu(), rows, H), ref2, 1e-3, 2 ** -8, what="add ln panel")
