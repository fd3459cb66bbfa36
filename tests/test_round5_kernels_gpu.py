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


def test_stage_dma_of_a_row_window_at_the_end_of_its_allocation(dbg_lib):
    """the buffer descriptor of the stage DMA must end with the rows the operand OWNS in its last K32 panel.  The phased engine
    passes row WINDOWS of larger panel buffers (first row > 0, ld = all rows) whose length is no multiple of the tile: A = rows 1000 .. 1999
    of a 2000-row panel buffer, the last panel's window ending exactly at the end of the torch allocation (a bound computed from the
    window's pointer with the whole buffer's extent overshoots it by first_row x 64 bytes).  Bit-identical with the clamped
    global_load_lds form, and equal to fp32 arithmetic on the same operands; W likewise a window (rows 128 .. 383 of 384)."""
    L = dbg_lib
    rows, first, M, K, N = 2000, 1000, 1000, 512, 256
    X, W, b = rnd(rows, K, seed=31), rnd(384, K, seed=32, scale=1 / math.sqrt(K)), rnd(N, seed=33, scale=0.5)
    Xp, Wp, bd = to_panel(X), to_panel(W), b.to(DEV)
    outs = {}
    for on in (1, 0):
        L.mh_gemm_set_buf_dma(on)
        out = torch.zeros(N // 32, M, 32, device=DEV, dtype=torch.bfloat16)
        check(L.mh_gemm_bias_act_ex(Xp.data_ptr() + first * 64, rows, 1, Wp.data_ptr() + 128 * 64, 384, 1, bd.data_ptr(), None, 0, 0, out.data_ptr(), M, 1, 0,
                                    M, N, K, 0, 1, current_stream()), "mh_gemm_bias_act_ex")
        outs[on] = out.clone()
    L.mh_gemm_set_buf_dma(1)
    assert torch.equal(outs[0], outs[1])
    ref = X[first:first + M].bfloat16().float() @ W[128:128 + N].bfloat16().float().T + b
    assert float((from_panel(outs[1]) - ref).abs().max()) < 3e-2
