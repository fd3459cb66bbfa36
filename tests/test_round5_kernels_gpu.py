"""Round-5 kernel forms against the forms they replace and against a torch fp32 reference of the same op on the same bf16-rounded operands."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from musediffusion_amd._lib import check, current_stream  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def to_panel(w):
    r, k = w.shape
    return w.bfloat16().reshape(r, k // 32, 32).permute(1, 0, 2).contiguous().to(DEV)


def from_panel(p):
    return p.permute(1, 0, 2).reshape(p.shape[1], -1).float().cpu()


@pytest.mark.parametrize("M,N,K,act", [(1000, 2048, 512, "gelu"), (16384, 2048, 512, "gelu"), (600, 3072, 768, "gelu"), (512, 256, 96, "gelu")])
def test_dense_gelu_with_bias_initialised_accumulators(M, N, K, act, dbg_lib):
    """mh_gemm_set_bias_acc: the 256x128 dense + GELU kernel whose accumulators START from the bias (an LDS-DMA piece per wave in front of the
    tile's first stage; no bias add in the epilogue) against the epilogue-add form and against fp32 arithmetic on the same bf16 operands.
    Partial row tiles (M 1000, 600), two widths, a K that is not a multiple of 64."""
    L = dbg_lib
    X, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3, scale=0.5)
    ref = torch.nn.functional.gelu(X.bfloat16().float() @ W.bfloat16().float().T + b)
    Xp, Wp, bd = to_panel(X), to_panel(W), b.to(DEV)
    outs = {}
    for on in (1, 0):
        L.mh_gemm_set_bias_acc(on)
        out = torch.zeros(N // 32, M, 32, device=DEV, dtype=torch.bfloat16)
        check(L.mh_gemm_bias_act_ex(Xp.data_ptr(), M, 1, Wp.data_ptr(), N, 1, bd.data_ptr(), None, 0, 0, out.data_ptr(), M, 1, 0, M, N, K, 2, 1,
                                    current_stream()), "mh_gemm_bias_act_ex")
        outs[on] = from_panel(out)
    L.mh_gemm_set_bias_acc(0)          # (the library's default)
    for on in (1, 0):
        err = (outs[on] - ref).abs()
        assert float(err.max()) < 3e-2 and float(err.mean()) < 2e-3, (on, float(err.max()), float(err.mean()))
    d = (outs[1] - outs[0]).abs()
    # the two forms round the same sum in another order: at most a bf16 ulp apart, on few elements
    assert float(d.max()) <= 2 ** -6 * max(1.0, float(ref.abs().max())) and float((d > 0).float().mean()) < 0.05, (float(d.max()), float((d > 0).float().mean()))


@pytest.mark.parametrize("B,L_,H,nh", [(3, 512, 512, 8), (2, 528, 768, 12)])
def test_qkv_projection_with_bias_initialised_accumulators(B, L_, H, nh, dbg_lib):
    """The QKV projection + head scatter (mh_gemm_qkv_vtperm: q, k token-major per head, V^T in the streaming kernel's key order) with the
    accumulators started from the bias - the V^T waves run the un-swapped MFMA, whose lanes hold one column each - against the epilogue-add form."""
    L = dbg_lib
    N = B * L_
    dh = H // nh
    X, W, b = rnd(N, H, seed=4), rnd(3 * H, H, seed=5, scale=1 / math.sqrt(H)), rnd(3 * H, seed=6, scale=0.5)
    Xp, Wp, bd = to_panel(X), to_panel(W), b.to(DEV)
    res = {}
    for on in (1, 0):
        L.mh_gemm_set_bias_acc(on)
        q, k, vt = (torch.zeros(N * H + 256, device=DEV, dtype=torch.bfloat16) for _ in range(3))
        check(L.mh_gemm_qkv_vtperm(Xp.data_ptr(), N, 1, Wp.data_ptr(), 3 * H, 1, bd.data_ptr(), q.data_ptr(), k.data_ptr(), vt.data_ptr(), B, L_, H, nh,
                                   current_stream()), "mh_gemm_qkv_vtperm")
        res[on] = (q.float().cpu(), k.float().cpu(), vt.float().cpu())
    L.mh_gemm_set_bias_acc(0)
    ref = X.bfloat16().float() @ W.bfloat16().float().T + b                                  # [N, 3H]
    qr = ref[:, :H].view(B, L_, nh, dh).permute(0, 2, 1, 3).reshape(-1)
    assert float((res[1][0][: N * H] - qr).abs().max()) < 3e-2
    for t1, t0 in zip(res[1], res[0]):
        d = (t1 - t0).abs()
        assert float(d.max()) <= 2 ** -5 and float((d > 0).float().mean()) < 0.05, (float(d.max()), float((d > 0).float().mean()))


@pytest.mark.parametrize("M,N,K", [(1000, 2048, 512), (16384, 2048, 512), (600, 3072, 768), (520, 256, 96)])
def test_stage_dma_as_buffer_loads_is_bit_identical(M, N, K, dbg_lib):
    """mh_gemm_set_buf_dma: the 256x128 panel kernels with their LDS-DMA stages issued as `buffer_load ... lds` (descriptor base + scalar K offset
    + immediate piece offset; rows beyond M / N unclamped) - the same bytes land in the same LDS slots, so dense + GELU and the QKV projection
    must come out bit for bit as with `global_load_lds`, partial tiles included."""
    L = dbg_lib
    X, W, b = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=1 / math.sqrt(K)), rnd(N, seed=13, scale=0.5)
    Xp, Wp, bd = to_panel(X), to_panel(W), b.to(DEV)
    outs = {}
    for on in (1, 0):
        L.mh_gemm_set_buf_dma(on)
        out = torch.zeros(N // 32, M, 32, device=DEV, dtype=torch.bfloat16)
        check(L.mh_gemm_bias_act_ex(Xp.data_ptr(), M, 1, Wp.data_ptr(), N, 1, bd.data_ptr(), None, 0, 0, out.data_ptr(), M, 1, 0, M, N, K, 2, 1,
                                    current_stream()), "mh_gemm_bias_act_ex")
        outs[on] = out.clone()
    L.mh_gemm_set_buf_dma(1)   # (the library's default)
    assert torch.equal(outs[0], outs[1])
    ref = torch.nn.functional.gelu(X.bfloat16().float() @ W.bfloat16().float().T + b)
    assert float((from_panel(outs[1]) - ref).abs().max()) < 3e-2


@pytest.mark.parametrize("B,L_,H,nh", [(3, 512, 512, 8), (2, 528, 768, 12)])
def test_qkv_projection_with_bias_initialised_accumulators(B, L_, H, nh, dbg_lib):
    """The QKV projection + head scatter (mh_gemm_qkv_vtperm: q, k token-major per head, V^T in the streaming kernel's key order) with the
    accumulators started from the bias - the V^T waves run the un-swapped MFMA, whose lanes hold one column each - against the epilogue-add form."""
    L = dbg_lib
    N = B * L_
    dh = H // nh
    X, W, b = rnd(N, H, seed=4), rnd(3 * H, H, seed=5, scale=1 / math.sqrt(H)), rnd(3 * H, seed=6, scale=0.5)
    Xp, Wp, bd = to_panel(X), to_panel(W), b.to(DEV)
    res = {}
    for on in (1, 0):
        L.mh_gemm_set_bias_acc(on)
        q, k, vt = (torch.zeros(N * H + 256, device=DEV, dtype=torch.bfloat16) for _ in range(3))
        check(L.mh_gemm_qkv_vtperm(Xp.data_ptr(), N, 1, Wp.data_ptr(), 3 * H, 1, bd.data_ptr(), q.data_ptr(), k.data_ptr(), vt.data_ptr(), B, L_, H, nh,
                                   current_stream()), "mh_gemm_qkv_vtperm")
        res[on] = (q.float().cpu(), k.float().cpu(), vt.float().cpu())
    L.mh_gemm_set_bias_acc(0)
    ref = X.bfloat16().float() @ W.bfloat16().float().T + b                                  # [N, 3H]
    qr = ref[:, :H].view(B, L_, nh, dh).permute(0, 2, 1, 3).reshape(-1)
    assert float((res[1][0][: N * H] - qr).abs().max()) < 3e-2
    for t1, t0 in zip(res[1], res[0]):
        d = (t1 - t0).abs()
        assert float(d.max()) <= 2 ** -5 and float((d > 0).float().mean()) < 0.05, (float(d.max()), float((d > 0).float().mean()))


@pytest.mark.parametrize("M,N,K", [(1000, 2048, 512), (16384, 2048, 512), (600, 3072, 768), (520, 256, 96)])
def test_stage_dma_as_buffer_loads_is_bit_identical(M, N, K, dbg_lib):
    """mh_gemm_set_buf_dma: the 256x128 panel kernels with their LDS-DMA stages issued as `buffer_load ... lds` (descriptor base + scalar K offset
    + immediate piece offset; rows beyond M / N unclamped) - the same bytes land in the same LDS slots, so dense + GELU and the QKV projection
    must come out bit for bit as with `global_load_lds`, partial tiles included."""
    L = dbg_lib
    X, W, b = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=1 / math.sqrt(K)), rnd(N, seed=13, scale=0.5)
    Xp, Wp, bd = to_panel(X), to_panel(W), b.to(DEV)
    outs = {}
    for on in (1, 0):
        L.mh_gemm_set_buf_dma(on)
        out = torch.zeros(N // 32, M, 32, device=DEV, dtype=torch.bfloat16)
        check(L.mh_gemm_bias_act_ex(Xp.data_ptr(), M, 1, Wp.data_ptr(), N, 1, bd.data_ptr(), None, 0, 0, out.data_ptr(), M, 1, 0, M, N, K, 2, 1,
                                    current_stream()), "mh_gemm_bias_act_ex")
        outs[on] = out.clone()
    L.mh_gemm_set_buf_dma(1)   # (the library's default)
    assert torch.equal(outs[0], outs[1])
    ref = torch.nn.functional.gelu(X.bfloat16().float() @ W.bfloat16().float().T + b)
    assert float((from_panel(outs[1]) - ref).abs().max()) < 3e-2


@pytest.mark.parametrize("M,N,K,act", [(1000, 2048, 512, 2), (32768, 512, 2048, 0), (600, 3072, 768, 0), (520, 256, 96, 1), (4096, 1536, 512, 0)])
def test_row_major_stage_dma_as_buffer_loads_is_bit_identical(M, N, K, act, dbg_lib):
    """the training step's launches (row-major bf16 operands, 256x256 / 256x128 tiles): descriptor per tile, the K step AND the piece in the scalar
    offset, one LDS slot per piece - bit for bit the `global_load_lds` result, partial tiles included (rows beyond M / N are not clamped:
    the descriptor's bound returns zeros there and those outputs are never stored)."""
    L = dbg_lib
    X, W, b = rnd(M, K, seed=21).to(DEV).bfloat16(), rnd(N, K, seed=22, scale=1 / math.sqrt(K)).to(DEV).bfloat16(), rnd(N, seed=23, scale=0.5).to(DEV)
    outs = {}
    for on in (1, 0):
        L.mh_gemm_set_buf_dma(on)
        out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        check(L.mh_gemm_bias_act_ex(X.data_ptr(), K, 0, W.data_ptr(), K, 0, b.data_ptr(), None, 0, 0, out.data_ptr(), N, 0, 0, M, N, K, act, 1,
                                    current_stream()), "mh_gemm_bias_act_ex")
        outs[on] = out.clone()
    L.mh_gemm_set_buf_dma(1)
    assert torch.equal(outs[0], outs[1])
    ref = X.float() @ W.float().T + b
    ref = torch.nn.functional.gelu(ref) if act == 2 else torch.tanh(ref) if act == 1 else ref
    assert float((outs[1].float() - ref).abs().max()) < 6e-2


@pytest.mark.parametrize("B,L_,H,nh", [(3, 512, 512, 8), (2, 528, 768, 12)])
def test_qkv_projection_with_buffer_dma_is_bit_identical(B, L_, H, nh, dbg_lib):
    L = dbg_lib
    N = B * L_
    X, W, b = rnd(N, H, seed=14), rnd(3 * H, H, seed=15, scale=1 / math.sqrt(H)), rnd(3 * H, seed=16, scale=0.5)
    Xp, Wp, bd = to_panel(X), to_panel(W), b.to(DEV)
    res = {}
    for on in (1, 0):
        L.mh_gemm_set_buf_dma(on)
        q, k, vt = (torch.zeros(N * H + 256, device=DEV, dtype=torch.bfloat16) for _ in range(3))
        check(L.mh_gemm_qkv_vtperm(Xp.data_ptr(), N, 1, Wp.data_ptr(), 3 * H, 1, bd.data_ptr(), q.data_ptr(), k.data_ptr(), vt.data_ptr(), B, L_, H, nh,
                                   current_stream()), "mh_gemm_qkv_vtperm")
        res[on] = (q.clone(), k.clone(), vt.clone())
    L.mh_gemm_set_buf_dma(1)
    for t1, t0 in zip(res[1], res[0]):
        assert torch.equal(t1, t0)
