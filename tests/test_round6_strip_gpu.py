"""Round 6: the column-strip kernel (csrc/gemm_strip.h) that serves dense + bias + GELU of K32 panels - the sampler's FFN1.  It must write
exactly the bits of gemm_big_kernel's 256 x 128 tile (same products, same order of the K dimension, same epilogue arithmetic), on every
block-to-tile mapping the launcher can choose, and leave the shapes it does not serve where they were."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _panel(x):
    r, c = x.shape
    return x.reshape(r, c // 32, 32).permute(1, 0, 2).contiguous()


def _rows(p, rows, cols):
    return p.reshape(cols // 32, -1, 32)[:, :rows].permute(1, 0, 2).reshape(rows, cols)


@pytest.fixture()
def dbg():
    from musediffusion_amd import _lib
    lib = _lib.use_debug_library()
    yield lib
    lib.mh_gemm_set_strip(1)
    _lib.use_debug_library(False)


def _ffn1(lib, Xp, lda, Wp, ldw, b, out, ldo, M, N, K, act=2):
    from musediffusion_amd import _lib
    _lib.check(lib.mh_gemm_bias_act_ex(Xp.data_ptr(), lda, 1, Wp.data_ptr(), ldw, 1, b.data_ptr(), None, 0, 0, out.data_ptr(), ldo, 1, 0, M, N, K, act, 1,
                                       _lib.current_stream()))


# (M, N): FFN1 of config 2 per batch slice; a full batch; 7 m-tiles in 3 uneven runs; 10 m-tiles x 3 strips (runs not a multiple of 8);
# two m-tiles = one run; 24 m-tiles x 16 strips (runs = 8 < 32: three tiles per block, a middle tile)
@pytest.mark.parametrize("M,N", [(16384, 2048), (32768, 2048), (1792, 2048), (2560, 384), (512, 128), (6144, 2048)])
def test_strip_kernel_writes_the_big_tile_kernels_bits(dbg, M, N):
    K = 512
    torch.manual_seed(M + N)
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda") * 0.5
    Xp, Wp = _panel(X), _panel(W)
    outs = []
    for strip in (1, 0):
        dbg.mh_gemm_set_strip(strip)
        o = torch.full((N // 32, M, 32), float("nan"), device="cuda", dtype=torch.bfloat16)
        _ffn1(dbg, Xp, M, Wp, N, b, o, M, M, N, K)
        torch.cuda.synchronize()
        outs.append(o)
    assert not torch.isnan(outs[0].float()).any()
    assert torch.equal(outs[0], outs[1])
    ref = torch.nn.functional.gelu(X.float() @ W.float().t() + b)
    got = _rows(outs[0], M, N).float()
    assert (got - ref).abs().max().item() <= 2e-2 + 8e-3 * ref.abs().max().item()   # bf16 output rounding


def test_strip_kernel_on_a_row_window_of_larger_panels(dbg):
    """A, W and the output as windows of larger panel buffers (lda > M, first row > 0): the descriptors end with the rows the operand owns, and
    nothing outside the output window is written"""
    M, N, K, LD, R0 = 1024, 256, 512, 1536, 256
    torch.manual_seed(5)
    Xall = torch.randn(LD, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    Xp, Wp = _panel(Xall), _panel(W)               # [K/32][LD][32]
    Xwin = Xp.reshape(-1)[R0 * 32:]                # first row R0 of panel 0
    outs = []
    for strip in (1, 0):
        dbg.mh_gemm_set_strip(strip)
        o = torch.full((N // 32, LD, 32), 7.0, device="cuda", dtype=torch.bfloat16)
        owin = o.reshape(-1)[R0 * 32:]
        _ffn1(dbg, Xwin, LD, Wp, N, b, owin, LD, M, N, K)
        torch.cuda.synchronize()
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    o = outs[0]
    assert (o[:, :R0] == 7.0).all() and (o[:, R0 + M:] == 7.0).all()
    ref = torch.nn.functional.gelu(Xall[R0:R0 + M].float() @ W.float().t() + b)
    got = o[:, R0:R0 + M].permute(1, 0, 2).reshape(M, N).float()
    assert (got - ref).abs().max().item() <= 2e-2 + 8e-3 * ref.abs().max().item()


def test_shapes_the_strip_kernel_does_not_serve_are_unchanged(dbg):
    """ragged M, K != 512, another activation: launch<0> keeps them on the big-tile kernel whatever the switch says"""
    for (M, N, K, act) in ((1000, 2048, 512, 2), (2048, 2048, 256, 2), (2048, 2048, 512, 0)):
        torch.manual_seed(M)
        X = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        Xp, Wp = _panel(X), _panel(W)
        outs = []
        for strip in (1, 0):
            dbg.mh_gemm_set_strip(strip)
            o = torch.zeros(N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
            _ffn1(dbg, Xp, M, Wp, N, b, o, M, M, N, K, act)
            torch.cuda.synchronize()
            outs.append(o)
        assert torch.equal(outs[0], outs[1])
        y = X.float() @ W.float().t() + b
        ref = torch.nn.functional.gelu(y) if act == 2 else y
        assert (_rows(outs[0], M, N).float() - ref).abs().max().item() <= 2e-2 + 8e-3 * ref.abs().max().item()
