"""Kernel-level tests of the round-3 training changes (GPU): each new form against the form it replaces - bit for bit where the
arithmetic is the same, within the stated tolerance where it is not.

* weight-gradient GEMM on the 256 x 256 tile == on the 256 x 128 tile (same slices, same summation order per output element);
* row-major GEMM launches on the automatically chosen 256 x 256 tile == the 256 x 128 tile;
* mh_layernorm_bwd_drop == mh_layernorm_bwd followed by mh_dropout_fwd on its dx (HF BertSelfOutput / BertOutput backward);
* mh_gemm_bias_act_dact: the activation is mh_gemm_bias_act_pre's, the stored derivative is gelu'(pre) to 1.1e-4 + bf16 rounding,
  and mh_gemm_act_grad(MH_ACT_DERIV) multiplies by it (HF BertIntermediate backward, reached from models/diffusion.py:594-699)."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from musediffusion_amd import _lib, ops  # noqa: E402
from musediffusion_amd._lib import MH_BF16, check, current_stream, lib  # noqa: E402

DEV = "cuda"
MH_ACT_GELU, MH_ACT_DERIV = 2, 4


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("K,M,N,colsum", [(8192, 2048, 512, 1), (8192, 512, 2048, 1), (4096, 1536, 512, 0), (4096, 512, 256, 1)])
def test_gemm_dw_wide_and_narrow_tiles_agree(K, M, N, colsum, dbg_lib):
    A = rnd(K, M, seed=901, scale=0.5).to(DEV).bfloat16().contiguous()
    B = rnd(K, N, seed=902, scale=0.5).to(DEV).bfloat16().contiguous()
    outs = []
    try:
        for wide in (0, 1):
            check(lib().mh_gemm_dw_set_wide(wide))
            S = int(lib().mh_gemm_dw_splits(K, M, N))
            n = M * N + (M if colsum else 0)
            part = torch.full((S, n), float("nan"), device=DEV)
            check(lib().mh_gemm_dw_bias(A.data_ptr(), M, B.data_ptr(), N, part.data_ptr(), S, K, M, N, colsum, current_stream()))
            out = torch.empty(n, device=DEV)
            check(lib().mh_sum_slices(part.data_ptr(), S, n, out.data_ptr(), current_stream()))
            outs.append((S, out.cpu()))
    finally:
        lib().mh_gemm_dw_set_wide(1)
    assert not torch.isnan(outs[1][1]).any()
    ref = A.float().T @ B.float()
    assert float((outs[1][1][: M * N].view(M, N).to(DEV) - ref).abs().max()) < 2e-3 * math.sqrt(K / 1024) + 1e-3 * float(ref.abs().max())
    if outs[0][0] == outs[1][0]:      # same token slices: every output element adds the same products in the same order
        assert torch.equal(outs[0][1][: M * N], outs[1][1][: M * N]), "dW differs between the tiles"
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-4)


def test_row_major_gemm_takes_the_wide_tile_with_the_same_result(dbg_lib):
    M, N, K = 16384, 1024, 256            # 64 x 4 = 256 wide tiles: one per CU, so the automatic rule switches
    A = rnd(M, K, seed=903, scale=0.5).to(DEV).bfloat16()
    W = (rnd(N, K, seed=904) / math.sqrt(K)).to(DEV).bfloat16()
    b = rnd(N, seed=905, scale=0.1).to(DEV)
    outs = []
    try:
        for auto in (0, 1):
            check(lib().mh_gemm_set_auto_wide(auto))
            outs.append(ops.gemm_bias_act(A, W, b, None, "gelu", MH_BF16))
    finally:
        lib().mh_gemm_set_auto_wide(1)
    assert torch.equal(outs[0], outs[1])
    ref = torch.nn.functional.gelu(A.float() @ W.float().T + b)
    assert float((outs[1].float() - ref).abs().max()) < 0.03


@pytest.mark.parametrize("H,rows", [(512, 4096), (768, 1000), (128, 264)])
def test_layernorm_backward_with_dropped_copy_matches_two_passes(H, rows):
    x = rnd(rows, H, seed=906).to(DEV).bfloat16()
    dy = rnd(rows, H, seed=907, scale=0.1).to(DEV).bfloat16()
    g = (1 + 0.1 * rnd(H, seed=908)).to(DEV)
    d = _lib.Dropout()
    d.p, d.seed, d.offset, d.mask = 0.1, 0x1234ABCD5678, (3 << 16) | 5, None
    nb = min(1024, (rows + 3) // 4)

    def run(fused):
        dx, dxm = torch.empty_like(x), torch.empty_like(x)
        part = torch.empty(2 * nb * H, device=DEV)
        dgb = torch.empty(2, H, device=DEV)
        if fused:
            check(lib().mh_layernorm_bwd_drop(x.data_ptr(), dy.data_ptr(), g.data_ptr(), dx.data_ptr(), dxm.data_ptr(), C.byref(d), part.data_ptr(), nb,
                                              dgb[0].data_ptr(), dgb[1].data_ptr(), 0, rows, H, 1e-12, MH_BF16, current_stream()))
        else:
            check(lib().mh_layernorm_bwd(x.data_ptr(), dy.data_ptr(), g.data_ptr(), dx.data_ptr(), part.data_ptr(), nb, dgb[0].data_ptr(),
                                         dgb[1].data_ptr(), 0, rows, H, 1e-12, MH_BF16, current_stream()))
            check(lib().mh_dropout_fwd(dx.data_ptr(), H, dxm.data_ptr(), H, rows, H, MH_BF16, C.byref(d), current_stream()))
        return dx, dxm, dgb

    a, b = run(True), run(False)
    for name, u, v in zip(("dx", "dx o keep / (1 - p)", "dgamma | dbeta"), a, b):
        assert torch.equal(u, v), name
    kept = float((a[1] != 0).float().mean())
    assert abs(kept - 0.9) < 0.01


def test_gelu_derivative_stored_by_the_forward():
    M, H, F = 2048, 512, 2048
    x = rnd(M, H, seed=910, scale=0.7).to(DEV).bfloat16()
    W1 = (rnd(F, H, seed=911) / math.sqrt(H) * 2.0).to(DEV).bfloat16()
    b1 = rnd(F, seed=912, scale=0.3).to(DEV)
    st = current_stream()
    pre, f0 = torch.empty(M, F, device=DEV, dtype=torch.bfloat16), torch.empty(M, F, device=DEV, dtype=torch.bfloat16)
    dact, f1 = torch.empty_like(pre), torch.empty_like(pre)
    check(lib().mh_gemm_bias_act_pre(x.data_ptr(), H, W1.data_ptr(), H, b1.data_ptr(), pre.data_ptr(), f0.data_ptr(), F, M, F, H, MH_ACT_GELU, st))
    check(lib().mh_gemm_bias_act_dact(x.data_ptr(), H, W1.data_ptr(), H, b1.data_ptr(), dact.data_ptr(), f1.data_ptr(), F, M, F, H, MH_ACT_GELU, st))
    assert torch.equal(f0, f1), "the activation must not depend on what is stored beside it"
    z = (x.float() @ W1.float().T + b1).double()
    ref = 0.5 * (1 + torch.erf(z / math.sqrt(2))) + z * torch.exp(-z * z / 2) / math.sqrt(2 * math.pi)
    err = (dact.double() - ref).abs()
    tol = 2e-4 + 2.0 ** -8 * ref.abs() + 1e-3 * (z.abs() < 1e-2)     # 1.1e-4 of the approximation + bf16 rounding of the stored value
    assert bool((err <= tol).all()), "gelu'(pre): max err %.3e" % float(err.max())
    assert float(z.abs().max()) > 4.0                                  # both tails are covered
    # the backward: (dY W2) o dact == what the pre-activation form gives, to the derivative's tolerance
    dy = rnd(M, H, seed=913, scale=0.1).to(DEV).bfloat16()
    W2T = (rnd(F, H, seed=914) / math.sqrt(F)).to(DEV).bfloat16()       # [F, H]: dpre = dY W2, the reduction over H
    d0, d1 = torch.empty_like(pre), torch.empty_like(pre)
    check(lib().mh_gemm_act_grad(dy.data_ptr(), H, W2T.data_ptr(), H, pre.data_ptr(), F, d0.data_ptr(), F, M, F, H, MH_ACT_GELU, st))
    check(lib().mh_gemm_act_grad(dy.data_ptr(), H, W2T.data_ptr(), H, dact.data_ptr(), F, d1.data_ptr(), F, M, F, H, MH_ACT_DERIV, st))
    exact = (dy.float() @ W2T.float().T) * dact.float()
    assert float((d1.float() - exact).abs().max()) <= 2.0 ** -8 * float(exact.abs().max()) + 1e-6
    assert float((d1.float() - d0.float()).abs().max()) <= 0.02 * float(d0.float().abs().max())


# ---------------------------------------------------------------------------------------------------------------------------------
# The keep-flag DEFINITION, restated on the host (numpy): Philox4x32-7 + the byte / tie-break rule + the lane-native word layout of
# csrc/common.h (drop_keep_attn, drop_word_index) and the dense sites' 16-bit rule (drop_keep8).  Pins the masks themselves, so that a
# re-expression of the decision code (round 3: one 16-bit compare per element instead of three byte compares) can be checked without an
# old build of the library.

def _philox7(c, k0, k1):
    import numpy as np
    c = [x.astype(np.uint64) for x in c]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(7):
        p0, p1 = M0 * c[0], M1 * c[2]
        n0 = (p1 >> np.uint64(32)) ^ c[1] ^ k0
        n2 = (p0 >> np.uint64(32)) ^ c[3] ^ k1
        c = [n0 & MASK, p1 & MASK, n2 & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c


@pytest.mark.parametrize("BH,L,p", [(3, 96, 0.1), (2, 528, 0.37), (1, 64, 0.0039)])
def test_attention_keep_bits_follow_the_documented_rule(BH, L, p):
    import numpy as np
    seed, offset = 0x1234567887654321, (7 << 16) | 3
    d = _lib.Dropout()
    d.p, d.seed, d.offset, d.mask = p, seed, offset, None
    n = int(lib().mh_dropout_bits_words(BH, L))
    bits = torch.zeros(n, dtype=torch.int32, device=DEV)
    check(lib().mh_dropout_bits(bits.data_ptr(), BH, L, C.byref(d), current_stream()))
    got = bits.cpu().numpy().view(np.uint32)
    nb = (L + 31) // 32
    nkp = (nb + 1) // 2
    p256 = np.float32(p) * np.float32(256.0)
    thr8 = int(p256)
    thr16 = thr8 * 256 + int((p256 - np.float32(thr8)) * np.float32(256.0) + np.float32(0.5))
    lane = np.arange(64)
    lq, h = lane & 31, lane >> 5
    for bh in range(BH):
        for qb in range(nb):
            q = np.minimum(qb * 32 + lq, L - 1)
            for kp in range(nkp):
                word = np.zeros(64, dtype=np.uint64)
                for half in range(2):
                    kb = 2 * kp + half
                    if kb >= nb:
                        continue
                    idx = ((((np.uint64(bh) * np.uint64(L) + q.astype(np.uint64)) * np.uint64(nb) + np.uint64(kb)) << np.uint64(1)) | h.astype(np.uint64))
                    c = _philox7([idx & np.uint64(0xFFFFFFFF), idx >> np.uint64(32), np.full(64, offset & 0xFFFFFFFF, np.uint64),
                                  np.full(64, offset >> 32, np.uint64)], seed & 0xFFFFFFFF, seed >> 32)
                    byte = lambda r: (c[r >> 2] >> np.uint64(8 * (r & 3))) & np.uint64(0xFF)   # noqa: E731
                    for r in range(16):
                        keep = (byte(r) * np.uint64(256) + byte((r + 1) & 15)) >= np.uint64(thr16)
                        word |= keep.astype(np.uint64) << np.uint64(r + 16 * half)
                base = (((bh * nb + qb) * nkp + kp) << 6)
                assert np.array_equal(got[base:base + 64], word.astype(np.uint32)), (bh, qb, kp)


def test_dense_site_keep_flags_follow_the_documented_rule():
    import numpy as np
    rows, cols, p = 37, 128, 0.1
    seed, offset = 0xABCDEF0123456789, (5 << 16) | 2
    d = _lib.Dropout()
    d.p, d.seed, d.offset, d.mask = p, seed, offset, None
    x = torch.ones(rows, cols, device=DEV)
    out = torch.empty_like(x)
    check(lib().mh_dropout_fwd(x.data_ptr(), cols, out.data_ptr(), cols, rows, cols, 0, C.byref(d), current_stream()))     # dtype 0 = fp32
    got = (out.cpu().numpy() != 0).reshape(-1)
    assert np.allclose(out.cpu().numpy()[out.cpu().numpy() != 0], 1.0 / (1.0 - p), rtol=1e-6)
    thr = int(np.float32(p) * np.float32(65536.0) + np.float32(0.5))
    g = np.arange(rows * cols // 8, dtype=np.uint64)
    c = _philox7([g & np.uint64(0xFFFFFFFF), g >> np.uint64(32), np.full(g.shape, offset & 0xFFFFFFFF, np.uint64), np.full(g.shape, offset >> 32, np.uint64)],
                 seed & 0xFFFFFFFF, seed >> 32)
    keep = np.zeros((g.size, 8), dtype=bool)
    for w in range(4):
        keep[:, 2 * w] = (c[w] & np.uint64(0xFFFF)) >= np.uint64(thr)
        keep[:, 2 * w + 1] = (c[w] >> np.uint64(16)) >= np.uint64(thr)
    assert np.array_equal(got, keep.reshape(-1))


def test_training_epilogues_on_the_wide_tile_match_the_narrow_tile(dbg_lib):
    """The tape's full-batch launches take the 256 x 256 tile (auto rule: row-major operands, every CU gets a tile); its epilogue
    variants - stored GELU derivative, multiply-by-derivative, dropout + residual - must give what the 256 x 128 tile gives, bit for bit
    (same K order per output element, same Philox element numbering)."""
    M, H, F = 32768, 512, 2048
    st = current_stream()
    x = rnd(M, H, seed=920, scale=0.7).to(DEV).bfloat16()
    W1 = (rnd(F, H, seed=921) / math.sqrt(H) * 2.0).to(DEV).bfloat16()
    b1 = rnd(F, seed=922, scale=0.3).to(DEV)
    dy = rnd(M, H, seed=923, scale=0.1).to(DEV).bfloat16()
    W2T = (rnd(F, H, seed=924) / math.sqrt(F)).to(DEV).bfloat16()
    W2 = (rnd(H, F, seed=925) / math.sqrt(F)).to(DEV).bfloat16()
    b2 = rnd(H, seed=926, scale=0.1).to(DEV)
    d = _lib.Dropout()
    d.p, d.seed, d.offset, d.mask = 0.1, 0x5555AAAA1234, (9 << 16) | 4, None
    res = {}
    try:
        for auto in (0, 1):
            check(lib().mh_gemm_set_auto_wide(auto))
            dact, f = torch.empty(M, F, device=DEV, dtype=torch.bfloat16), torch.empty(M, F, device=DEV, dtype=torch.bfloat16)
            check(lib().mh_gemm_bias_act_dact(x.data_ptr(), H, W1.data_ptr(), H, b1.data_ptr(), dact.data_ptr(), f.data_ptr(), F, M, F, H, MH_ACT_GELU, st))
            dpre = torch.empty_like(f)
            check(lib().mh_gemm_act_grad(dy.data_ptr(), H, W2T.data_ptr(), H, dact.data_ptr(), F, dpre.data_ptr(), F, M, F, H, MH_ACT_DERIV, st))
            y = torch.empty(M, H, device=DEV, dtype=torch.bfloat16)
            check(lib().mh_gemm_bias_dropout_res(f.data_ptr(), F, W2.data_ptr(), F, b2.data_ptr(), x.data_ptr(), H, y.data_ptr(), H, M, H, F, MH_BF16, C.byref(d), st))
            res[auto] = (dact, f, dpre, y)
    finally:
        lib().mh_gemm_set_auto_wide(1)
    for name, u, v in zip(("gelu'(pre)", "gelu(pre)", "(dY W2) o gelu'", "dropout(f W2 + b) + x"), res[0], res[1]):
        assert torch.equal(u, v), name
    kept = float(((res[1][3].float() - x.float()).abs() > 0).float().mean())
    assert 0.88 < kept < 0.92
