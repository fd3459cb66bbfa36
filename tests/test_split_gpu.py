"""Split-precision kernels (csrc/split.hip: compute_dtype "bf16x3" / "f16x3") against float64 torch references of the same ops on
the same fp32 operands.  The tolerances are the modes' stated accuracy: a value held as hi + lo keeps 16 (bf16) / 22 (f16) mantissa
bits, a product drops only lo x lo, sums are fp32 - errors sit within a few fp32 ulps of the fp32 mode's for f16x3 and ~8x above for
bf16x3, two to three orders below the bf16 mode's.  Reference op groups: models/network.py:141-157, HF BertLayer from network.py:151."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from musediffusion_amd import _lib  # noqa: E402
from musediffusion_amd._lib import MH_BF16X3, MH_F16X3, check, current_stream, lib  # noqa: E402

DEV = "cuda"
MODES = [pytest.param(MH_BF16X3, id="bf16x3"), pytest.param(MH_F16X3, id="f16x3")]
TDT = {MH_BF16X3: torch.bfloat16, MH_F16X3: torch.float16}
# max |error| / rms(reference) of a length-512 product sum (tools/micro/split_mfma.hip measures 1.2e-5 / 2.5e-6)
GEMM_TOL = {MH_BF16X3: 8e-5, MH_F16X3: 1.2e-5}
PART_TOL = {MH_BF16X3: 2.0 ** -16, MH_F16X3: 2.0 ** -21}   # relative error of hi + lo against the fp32 value (|v| >= 2^-3 for f16)


def rnd(*shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def pack(x, dt, kpad=None, ld=None):
    """fp32 [rows, cols] -> split panels (uint8 buffer holding [2][kpad/32][ld][32] 16-bit)"""
    rows, cols = x.shape
    kpad = kpad or (cols + 31) // 32 * 32
    ld = ld or rows
    xd = x.to(DEV).float().contiguous()
    out = torch.zeros(2 * (kpad // 32) * ld * 32, dtype=TDT[dt], device=DEV)
    check(lib().mh_split_pack(xd.data_ptr(), cols, out.data_ptr(), ld, rows, cols, kpad, dt, current_stream()), "mh_split_pack")
    return out


def join(buf, rows, cols, dt, cpad=None, ld=None):
    cpad = cpad or (cols + 31) // 32 * 32
    out = torch.zeros(rows, cols, device=DEV)
    check(lib().mh_split_join(buf.data_ptr(), ld or rows, out.data_ptr(), cols, rows, cols, cpad, dt, current_stream()), "mh_split_join")
    return out.cpu()


@pytest.mark.parametrize("dt", MODES)
def test_pack_join_round_trip(dt):
    x = rnd(200, 72, seed=1) * torch.tensor([1.0, 30.0, 0.2]).repeat(24)[None]
    x = torch.where(x.abs() < 0.125, torch.full_like(x, 0.5), x)     # (f16 lo parts of smaller values are subnormal: absolute 2^-25 then)
    buf = pack(x, dt, kpad=96, ld=256)
    back = join(buf, 200, 72, dt, cpad=96, ld=256)
    rel = ((back - x).abs() / x.abs()).max()
    assert float(rel) <= PART_TOL[dt], float(rel)
    parts = buf.view(2, 3, 256, 32)
    assert float(parts[:, 2, :200, 8:].abs().max()) == 0.0       # K padding (columns 72 .. 95) is zero in both parts
    hi = parts[0, :, :200].permute(1, 0, 2).reshape(200, 96)[:, :72].float().cpu()
    assert torch.equal(hi, x.to(TDT[dt]).float())                 # hi is the value rounded to the 16-bit type
    if dt == MH_F16X3:   # small values keep an absolute precision of 2^-25 (subnormal lo parts are not flushed)
        s = rnd(64, 32, seed=2, scale=1e-3)
        assert float((join(pack(s, dt), 64, 32, dt) - s).abs().max()) <= 2.0 ** -24


@pytest.mark.parametrize("dt", MODES)
@pytest.mark.parametrize("M,N,K,act,res,mode", [
    (512, 256, 128, 0, False, 2), (300, 64, 64, 1, False, 0), (1000, 512, 512, 2, False, 0), (256, 128, 2048, 0, True, 2),
    (700, 500, 512, 0, False, 2), (130, 128, 96, 0, False, 1), (4096, 2048, 512, 2, False, 0)])
def test_split_gemm(dt, M, N, K, act, res, mode):
    """act(A W^T + b) [+ residual] in the three output forms, ragged M / N (row and column guards, the fp32 output's partial last group
    at N = 500), K from 2 to 64 K-steps per term"""
    A, W, b = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=1 / math.sqrt(K)), rnd(N, seed=5, scale=0.3)
    R = rnd(M, N, seed=6) if res else None
    ref = A.double() @ W.double().T + b.double()
    if act == 1:
        ref = torch.tanh(ref)
    if act == 2:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + R.double()
    Ap, Wp, bd = pack(A, dt), pack(W, dt), b.to(DEV)
    Rp = pack(R, dt) if res else None
    if mode == 2:
        out = torch.zeros(M, N, device=DEV)
        ldo, part = N, 0
    elif mode == 0:
        out = torch.zeros(2 * (N // 32) * M * 32, dtype=TDT[dt], device=DEV)
        ldo, part = M, 0
    else:
        out = torch.zeros(2, M, N, dtype=TDT[dt], device=DEV)
        ldo, part = N, M * N
    check(lib().mh_split_gemm(Ap.data_ptr(), M, Wp.data_ptr(), N, bd.data_ptr(), 0, Rp.data_ptr() if res else None, M, out.data_ptr(), ldo, mode, part,
                              M, N, K, act, dt, current_stream()), "mh_split_gemm")
    got = out.cpu().double() if mode == 2 else (join(out, M, N, dt).double() if mode == 0 else out[0].double().cpu() + out[1].double().cpu())
    err = float((got - ref).abs().max() / ref.pow(2).mean().sqrt())
    fp32 = A @ W.T + b
    if act == 1: fp32 = torch.tanh(fp32)
    if act == 2: fp32 = torch.nn.functional.gelu(fp32)
    if res: fp32 = fp32 + R
    print("split gemm %dx%dx%d act %d: max |err| / rms %.2e (torch fp32 on the host: %.2e)" % (M, N, K, act, err, float((fp32.double() - ref).abs().max() / ref.pow(2).mean().sqrt())))
    assert err <= GEMM_TOL[dt] * (2 if K > 1024 else 1), err


@pytest.mark.parametrize("dt", MODES)
def test_split_gemm_bias_per_row_is_the_transposed_projection(dt):
    """V^T = W_v X^T + b_v (rows): what the attention kernel reads as its V^T operand, written split row-major"""
    H, Ntok, K = 128, 600, 128
    Wv, X, b = rnd(H, K, seed=7, scale=1 / math.sqrt(K)), rnd(Ntok, K, seed=8), rnd(H, seed=9, scale=0.3)
    ref = (X.double() @ Wv.double().T + b.double()).T
    Wp, Xp = pack(Wv, dt), pack(X, dt)
    ldo = 608
    out = torch.zeros(2, H, ldo, dtype=TDT[dt], device=DEV)
    check(lib().mh_split_gemm(Wp.data_ptr(), H, Xp.data_ptr(), Ntok, b.to(DEV).data_ptr(), 1, None, 0, out.data_ptr(), ldo, 1, H * ldo, H, Ntok, K, 0, dt,
                              current_stream()), "mh_split_gemm")
    got = (out[0].double() + out[1].double()).cpu()[:, :Ntok]
    assert float((got - ref).abs().max() / ref.pow(2).mean().sqrt()) <= GEMM_TOL[dt]
    assert float(out[:, :, Ntok:].abs().max()) == 0.0


@pytest.mark.parametrize("dt", MODES)
@pytest.mark.parametrize("H,B,L,add", [(512, 3, 40, True), (64, 2, 16, False), (768, 2, 24, True), (2048, 1, 8, False)])
def test_split_layernorm(dt, H, B, L, add):
    N = B * L
    x = rnd(N, H, seed=10, scale=2.0)
    pos, emb = rnd(L, H, seed=11), rnd(B + 1, H, seed=12)
    rows = torch.tensor([(b + 1) % (B + 1) for b in range(B)], dtype=torch.int32)
    g, bt = 1 + rnd(H, seed=13, scale=0.2), rnd(H, seed=14, scale=0.2)
    pre = ((pos[None] + x.view(B, L, H)) + emb[rows.long()][:, None]).view(N, H) if add else x
    ref = torch.nn.functional.layer_norm(pre.double(), (H,), g.double(), bt.double(), 1e-12)
    d = lambda t: t.to(DEV).contiguous()
    out = torch.zeros(2 * (H // 32) * N * 32, dtype=TDT[dt], device=DEV)
    xd, pd, ed, rd, gd, bd = d(x), d(pos), d(emb), d(rows), d(g), d(bt)
    check(lib().mh_split_layernorm(xd.data_ptr(), H, pd.data_ptr() if add else None, ed.data_ptr() if add else None, rd.data_ptr() if add else None,
                                   gd.data_ptr(), bd.data_ptr(), out.data_ptr(), N, N, L, H, 1e-12, dt, current_stream()), "mh_split_layernorm")
    err = float((join(out, N, H, dt).double() - ref).abs().max())
    assert err <= (6e-5 if dt == MH_BF16X3 else 4e-6), err


@pytest.mark.parametrize("dt", MODES)
@pytest.mark.parametrize("B,L,nh,dh", [(2, 16, 4, 16), (3, 24, 2, 32), (2, 136, 2, 64), (2, 512, 8, 64), (1, 1024, 2, 64)])
def test_split_attention(dt, B, L, nh, dh):
    """softmax(q k^T / sqrt(dh)) v per (batch, head), no mask (HF BertSelfAttention as called from network.py:151): head dims 16 / 32 / 64,
    key tiles with a masked tail (L = 16, 24, 136), the operand images the projections write (packed q | k rows, transposed v)"""
    H, N = nh * dh, B * L
    q, k, v = rnd(N, H, seed=15), rnd(N, H, seed=16), rnd(N, H, seed=17)
    qh = lambda t: t.view(B, L, nh, dh).permute(0, 2, 1, 3).double()
    sc = torch.softmax(qh(q) @ qh(k).transpose(-1, -2) / math.sqrt(dh), dim=-1)
    ref = (sc @ qh(v)).permute(0, 2, 1, 3).reshape(N, H)
    T = TDT[dt]

    def parts(t):   # [rows, cols] fp32 -> [2][rows][cols] hi / lo on the device
        hi = t.to(T)
        return torch.stack([hi, (t - hi.float()).to(T)]).to(DEV).contiguous()
    qk = parts(torch.cat([q, k], dim=1))
    vt = parts(v.T.contiguous())
    ctx = torch.zeros(2 * (H // 32) * N * 32, dtype=T, device=DEV)
    check(lib().mh_split_attention(qk.data_ptr(), 2 * H, H, N * 2 * H, vt.data_ptr(), N, H * N, ctx.data_ptr(), N, B, L, nh, dh, 1.0 / math.sqrt(dh), dt,
                                   current_stream()), "mh_split_attention")
    err = float((join(ctx, N, H, dt).double() - ref).abs().max())
    print("split attention L=%d dh=%d: max |err| %.2e" % (L, dh, err))
    assert err <= (5e-5 if dt == MH_BF16X3 else 6e-6), err


@pytest.mark.parametrize("dt", MODES)
@pytest.mark.parametrize("M,K", [(256, 512), (1000, 2048), (4096, 512)])
def test_split_gemm_res_ln_fused(dt, M, K):
    """LayerNorm(A W^T + b + residual) in one kernel (hidden size 512: a block owns complete rows) against float64, and against the two
    launches it replaces (same tolerance class: the pre-LayerNorm rows never leave fp32 in either form)"""
    N = 512
    assert lib().mh_split_gemm_res_ln_supported(N) == 1 and lib().mh_split_gemm_res_ln_supported(768) == 0
    A, W, b, R = rnd(M, K, seed=20), rnd(N, K, seed=21, scale=1 / math.sqrt(K)), rnd(N, seed=22, scale=0.3), rnd(M, N, seed=23)
    g, bt = 1 + rnd(N, seed=24, scale=0.2), rnd(N, seed=25, scale=0.2)
    pre = A.double() @ W.double().T + b.double() + R.double()
    ref = torch.nn.functional.layer_norm(pre, (N,), g.double(), bt.double(), 1e-12)
    Ap, Wp, Rp = pack(A, dt), pack(W, dt), pack(R, dt)
    d = lambda t: t.to(DEV).contiguous()
    bd, gd, btd = d(b), d(g), d(bt)
    out = torch.zeros(2 * (N // 32) * M * 32, dtype=TDT[dt], device=DEV)
    check(lib().mh_split_gemm_res_ln(Ap.data_ptr(), M, Wp.data_ptr(), N, bd.data_ptr(), Rp.data_ptr(), M, gd.data_ptr(), btd.data_ptr(), 1e-12, out.data_ptr(), M,
                                     M, N, K, dt, current_stream()), "mh_split_gemm_res_ln")
    got = join(out, M, N, dt).double()
    err = float((got - ref).abs().max())
    # the two launches it replaces
    rows = torch.zeros(M, N, device=DEV)
    check(lib().mh_split_gemm(Ap.data_ptr(), M, Wp.data_ptr(), N, bd.data_ptr(), 0, Rp.data_ptr(), M, rows.data_ptr(), N, 2, 0, M, N, K, 0, dt, current_stream()), "mh_split_gemm")
    out2 = torch.zeros_like(out)
    check(lib().mh_split_layernorm(rows.data_ptr(), N, None, None, None, gd.data_ptr(), btd.data_ptr(), out2.data_ptr(), M, M, 1, N, 1e-12, dt, current_stream()), "mh_split_layernorm")
    err2 = float((join(out2, M, N, dt).double() - ref).abs().max())
    print("fused dense + LN M=%d K=%d: max |err| %.2e (separate launches %.2e)" % (M, K, err, err2))
    tol = (2e-4 if dt == MH_BF16X3 else 3e-5) * (2 if K > 1024 else 1)
    assert err <= tol and err2 <= tol, (err, err2)


@pytest.mark.parametrize("dt", MODES)
def test_split_kernels_repeat_launches_are_bit_identical_under_load(dt):
    """Race screen of the LDS-DMA rings (persistent blocks walking several tiles, the next tile's stages issued before the epilogue, W hi
    fragments kept across two stages) and of the attention's double buffer: 60 launches of each kernel on the same operands while a second
    stream keeps the chip busy - every launch must reproduce the first bit for bit."""
    M, N, K, H = 8192, 2048, 512, 512
    A, W, b = pack(rnd(M, K, seed=30), dt), pack(rnd(N, K, seed=31, scale=K ** -0.5), dt), rnd(N, seed=32).to(DEV)
    W2, R = pack(rnd(H, N, seed=33, scale=N ** -0.5), dt), pack(rnd(M, H, seed=34), dt)
    g, bt = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    T = TDT[dt]
    out1 = torch.zeros(2 * (N // 32) * M * 32, dtype=T, device=DEV)
    out2 = torch.zeros(2 * (H // 32) * M * 32, dtype=T, device=DEV)
    B, L, nh, dh = 8, 512, 8, 64
    Ntok = B * L
    qk = (torch.randn(2, Ntok, 2 * H, generator=torch.Generator().manual_seed(35)) * 0.5).to(T).to(DEV)
    vt = (torch.randn(2, H, Ntok, generator=torch.Generator().manual_seed(36)) * 0.5).to(T).to(DEV)
    ctx = torch.zeros(2 * (H // 32) * Ntok * 32, dtype=T, device=DEV)
    na, nb = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16), torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
    side = torch.cuda.Stream()

    def run():
        check(lib().mh_split_gemm(A.data_ptr(), M, W.data_ptr(), N, b.data_ptr(), 0, None, 0, out1.data_ptr(), M, 0, 0, M, N, K, 2, dt, current_stream()), "gemm")
        check(lib().mh_split_gemm_res_ln(out1.data_ptr(), M, W2.data_ptr(), H, b.data_ptr(), R.data_ptr(), M, g.data_ptr(), bt.data_ptr(), 1e-12, out2.data_ptr(), M,
                                         M, H, N, dt, current_stream()), "gemm_ln")
        check(lib().mh_split_attention(qk.data_ptr(), 2 * H, H, Ntok * 2 * H, vt.data_ptr(), Ntok, H * Ntok, ctx.data_ptr(), Ntok, B, L, nh, dh, 0.125, dt,
                                       current_stream()), "attention")
    run()
    torch.cuda.synchronize()
    ref = (out1.clone(), out2.clone(), ctx.clone())
    assert all(bool(torch.isfinite(t.float()).all()) for t in ref)
    bad = 0
    for rep in range(60):
        with torch.cuda.stream(side):
            na @ nb
        out1.zero_(); out2.zero_(); ctx.zero_()
        run()
        torch.cuda.synchronize()
        bad += int(not (torch.equal(out1, ref[0]) and torch.equal(out2, ref[1]) and torch.equal(ctx, ref[2])))
    assert bad == 0, "%d of 60 repeated launches differ from the first" % bad
