"""Training path: `training_losses` (MuseDiffusion/models/diffusion.py:594-699) with gradients, as
called from TrainLoop._forward_backward_logic (utils/train_util.py:188-232).

The reference relies on torch autograd over ~500 ATen kernels per micro-batch.  Here every tensor
operation — forward and backward — is a libmusehip kernel; torch.autograd is only the tape that
connects them, so `.backward()`, DDP's gradient all-reduce hooks (RCCL), optimizers, EMA and
checkpointing keep working on the model's ordinary fp32 `nn.Parameter`s.

Backward matrix products are brought to the forward's "A W^T" GEMM form:
    dX = dY W          -> gemm(dY, W^T)            (weight transposed once per call)
    dW = dY^T X        -> gemm(dY^T, X^T)          (activations transposed; reduction over tokens)
and attention backward recomputes P = softmax(Q K^T) per (batch, head) with batched GEMMs.
Activations are kept in the compute dtype (fp32 = the reference's precision, bf16 optional), all
parameter gradients are produced in fp32.
"""
import math

import torch
from torch.autograd import Function

from . import _lib, ops
from ._lib import MH_F32, check, current_stream, lib, ptr

NPART = 256  # rows of the two-stage column-sum scratch
TN_DW = True            # bf16: weight gradients straight from the k-major activations (mh_gemm_dw), no transposed copies
FUSED_FFN = True        # bf16: FFN as one tape node with the GELU backward fused into a GEMM epilogue (A/B switch for tests)
FUSED_ATTENTION = True  # bf16: streaming forward + fused backward kernels when the shape allows (A/B switch for tests)
LN_BWD_DROP = True           # the LayerNorm backward also writes its input gradient x the dense node's dropout mask (mh_layernorm_bwd_drop)
GELU_DERIV_FWD = True        # the FFN node keeps gelu'(pre) instead of pre (mh_gemm_bias_act_dact; A/B: tools/ab_train.py GELU_DERIV_FWD)
ACT_DERIV = 4                # MH_ACT_DERIV
WEIGHT_PREP = True      # bf16: the encoder's weight casts / transposes of a forward + backward in one launch (_WeightPrep)
FUSED_DENSE_LN = True   # bf16, d_model 512: BertSelfOutput / BertOutput dense -> dropout -> + input -> LayerNorm as one kernel
# round 5: the fused q | k | v projection as tape-only stacking of its three parameters (no fp32 torch.cat per layer and step, one bias pack per
# forward) and the attention-output dense taking the projection node's pass-through of X as its residual (the residual branch's gradient is
# added in the projection's input-gradient GEMM, not by an autograd add kernel).  False = round 4's tape (A/B: tools/ab_train.py TAPE_STACK)
TAPE_STACK = True
# round 6: the encoder layers as one tape node each over libmusehip's per-layer calls (csrc/train_layer.hip): every GEMM operand in the
# K32-panel layout of the sampler's tiles.  Where the shape is served (d_model 512, ffn % 256 == 0, seq_len % 64 == 0 and >= 512, head dim
# 32 / 64); other shapes and fp32 keep the node-per-op tape above.  False = that tape everywhere (A/B and the equivalence test)
PANEL_LAYERS = True
FOLD_SIDE_STREAM = True  # the panel layers' backward folds its gradient partials on the model's side stream (A/B: tools/ab_train.py)
PANEL_LAYER_CALLS = 0   # forward calls of the panel-layer node so far (tests assert which tape ran)
# rows added to every panel buffer of the panel layers: with exactly B L = 2^k rows the panels of a tensor lie 2^(k + 6) bytes apart and
# the 16 panels a weight-gradient block streams, or a GEMM tile's K steps, fall on the same HBM channels
import os as _os
PANEL_LD_PAD = int(_os.environ.get("MUSE_PANEL_LD_PAD", "64"))


class _FusedLN:
    """Hand-over between a dense node and the LayerNorm node behind it: when the dense node can run the one-kernel form
    (mh_gemm_bias_dropout_res_ln) it leaves the normalised rows here and the LayerNorm node's forward takes them instead of launching
    its own kernel.  Both nodes keep their own backward (the dense node returns the UN-normalised rows, which is what the LayerNorm
    backward reads), so the tape is the same as with two kernels."""

    def __init__(self, ln):
        self.g, self.b, self.eps, self.y = ln.weight, ln.bias, float(ln.eps), None
        self.drop = None     # the dense node's dropout site when it ran the one-kernel form: the LayerNorm backward then also writes ...
        self.dym = None      # ... dx o keep / (1 - p), which the dense node's backward takes instead of a separate mh_dropout_fwd pass

    def usable(self, x, N, Kp, residual, dt, act):
        return (FUSED_DENSE_LN and dt == ops.MH_BF16 and act is None and residual is not None and N == 512 and Kp % 32 == 0
                and x.shape[1] % 8 == 0 and residual.shape[1] == N and self.g.dtype == torch.float32)

    def run(self, x, Wc, b, residual, pre, drop):
        import ctypes as C
        M, Kp = x.shape
        N = Wc.shape[0]
        self.y = torch.empty_like(pre)
        d = drop.c() if drop is not None else None
        self.drop = drop if (LN_BWD_DROP and drop is not None) else None
        check(lib().mh_gemm_bias_dropout_res_ln(ptr(x), Kp, ptr(Wc), Wc.shape[1], ptr(b), ptr(residual), residual.shape[1],
                                                ptr(self.g.detach()), ptr(self.b.detach()), self.eps, ptr(pre), ptr(self.y), N, M, N, Kp,
                                                C.byref(d) if d is not None else None, current_stream()), "mh_gemm_bias_dropout_res_ln")
        return pre


def _td(dt):
    return ops.TORCH_DTYPE[dt]


def _zeros(rows, cols, dt, dev, valid_cols=None):
    """[rows, cols] buffer whose padding columns (beyond valid_cols) must read as zero; no fill when there are none."""
    if valid_cols is not None and valid_cols == cols:
        return torch.empty(rows, cols, device=dev, dtype=_td(dt))
    return torch.zeros(rows, cols, device=dev, dtype=_td(dt))


def _gemm(A, W, bias, dt, N, K, out=None, out_f32=False, residual=None):
    return ops.gemm_bias_act(A, W, bias, residual, None, dt, out_f32=out_f32, N=N, K=K, out=out)


class _Drop:
    """One dropout site of one forward call: rate, Philox key / counter offset (the backward re-creates the mask from them),
    and - for parity tests only - an explicit keep mask (`mask`: uint8 [rows, cols] for dense sites; `bits`: the int32 keep-bit
    tensor of an attention site, layout of mh_dropout_bits)."""
    __slots__ = ("p", "seed", "offset", "mask", "bits", "ready")

    def __init__(self, p, seed, offset, mask=None, bits=None, ready=None):
        self.p, self.seed, self.offset, self.mask, self.bits = float(p), int(seed), int(offset), mask, bits
        self.ready = ready      # event after which `bits` (generated on a side stream) may be read

    def c(self):
        d = _lib.Dropout()
        d.p, d.seed, d.offset = self.p, self.seed & (2 ** 64 - 1), self.offset & (2 ** 64 - 1)
        d.mask = ptr(self.mask)
        return d

    def apply(self, x, dt):
        """x o keep / (1 - p) for a dense [rows, cols] tensor (the same op forward and backward)."""
        import ctypes as C
        x = x.contiguous()
        out = torch.empty_like(x)
        d = self.c()
        check(lib().mh_dropout_fwd(ptr(x), x.shape[1], ptr(out), out.shape[1], x.shape[0], x.shape[1], dt, C.byref(d), current_stream()),
              "mh_dropout_fwd")
        return out


def _dropped_grad(ctx_ln, drop, dy, dt):
    """dY o keep / (1 - p) for a dense node's backward: taken from the LayerNorm backward when it wrote it next to this very dY
    (_FusedLN.dym), else one mh_dropout_fwd pass."""
    if ctx_ln is not None and ctx_ln.dym is not None:
        src, dym = ctx_ln.dym
        ctx_ln.dym = None
        if src is dy or (src.data_ptr() == dy.data_ptr() and src.shape == dy.shape and src.dtype == dy.dtype and src._version == dy._version):
            return dym
    return drop.apply(dy, dt)


def _active(drop):
    """A site drops something: an explicit mask, or a rate the 16-bit threshold of the kernels can express (csrc/common.h drop_thr:
    round(65536 p) > 0 - below 2^-17 the kernels' own test `thr == 0` reads "off", and every host decision must agree with it)."""
    return drop is not None and (drop.mask is not None or drop.bits is not None or int(drop.p * 65536.0 + 0.5) > 0)


def _gemm_drop(A, W, bias, dt, N, K, out, residual, drop):
    """out = dropout(A W^T + bias) + residual (HF BertSelfOutput / BertOutput in train mode)."""
    import ctypes as C
    d = drop.c()
    check(lib().mh_gemm_bias_dropout_res(ptr(A), A.shape[1], ptr(W), W.shape[1], ptr(bias), ptr(residual),
                                         residual.shape[1] if residual is not None else 0, ptr(out), out.shape[1], A.shape[0], N, K, dt,
                                         C.byref(d), current_stream()), "mh_gemm_bias_dropout_res")
    return out


def _gemm_dw(dT, xT, N, Kp, Mp, dt):
    """dW [N, Kp] fp32 = dT [N, Mp] . xT [Kp, Mp]^T: the reduction runs over the Mp tokens while the output is tiny,
    so the K range is cut into slices that run as the batch of one batched GEMM (enough tiles to fill the chip)
    and the fp32 partials are summed in a fixed order."""
    tiles = ((N + 255) // 256) * ((Kp + 127) // 128)
    S = 1
    while S < 64 and tiles * S < 512 and Mp % (2 * S * 64) == 0 and Mp // (2 * S) >= 512:
        S *= 2
    dW = torch.empty(N, Kp, device=dT.device, dtype=torch.float32)
    if S == 1:
        _gemm(dT, xT, None, dt, Kp, Mp, out=dW, out_f32=True)
        return dW
    Ks = Mp // S
    part = torch.empty(S, N, Kp, device=dT.device, dtype=torch.float32)
    check(lib().mh_gemm_batched(ptr(dT), Mp, Ks, ptr(xT), Mp, Ks, None, ptr(part), Kp, N * Kp, 1, S, N, Kp, Ks, dt, current_stream()),
          "mh_gemm_batched")
    check(lib().mh_sum_slices(ptr(part), S, N * Kp, ptr(dW), current_stream()), "mh_sum_slices")
    return dW


def _dw(dy, x, N, Kp, M, dt, bias=False):
    """dW [N, Kp] fp32 = dy[:, :N]^T x[:, :Kp] over the M token rows (bias: and db [N] = column sums of dy, returned as a pair).
    bf16: one kernel reads both operands as the forward left them (k-major), transposes inside LDS and takes the column sums off
    the matrix pipe; otherwise explicit transposes + the split-K GEMM and a column-sum pass."""
    if TN_DW and dt == ops.MH_BF16 and N % 8 == 0 and Kp % 8 == 0 and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 and M % 32 == 0:
        S = int(lib().mh_gemm_dw_splits(M, N, Kp))
        n = N * Kp + (N if bias else 0)
        part = torch.empty(S, n, device=dy.device, dtype=torch.float32)
        check(lib().mh_gemm_dw_bias(ptr(dy), dy.shape[1], ptr(x), x.shape[1], ptr(part), S, M, N, Kp, int(bias), current_stream()),
              "mh_gemm_dw_bias")
        if S == 1:
            out = part[0]
        else:
            out = torch.empty(n, device=dy.device, dtype=torch.float32)
            check(lib().mh_sum_slices(ptr(part), S, n, ptr(out), current_stream()), "mh_sum_slices")
        dW = out[:N * Kp].view(N, Kp)
        return (dW, out[N * Kp:]) if bias else dW
    Mp = ops.pad64(M)
    dW = _gemm_dw(_transpose(dy, M, N, dt, ld_out=Mp), _transpose(x, M, Kp, dt, ld_out=Mp), N, Kp, Mp, dt)
    return (dW, _col_sum(dy, M, N, dt)) if bias else dW


def _transpose(x, rows, cols, dt, ld_out=None, batch=1, stride_in=0, stride_out=0, ld_in=None):
    """[batch][rows, cols] (row pitch ld_in, default x.shape[-1]) -> [batch][cols, ld_out] zero-padded."""
    ld_out = ops.pad64(rows) if ld_out is None else ld_out
    ld_in = x.shape[-1] if ld_in is None else ld_in
    out = (torch.empty if ld_out == rows else torch.zeros)(batch * cols, ld_out, device=x.device, dtype=x.dtype)
    check(lib().mh_transpose(ptr(x), ld_in, stride_in, ptr(out), ld_out, stride_out or cols * ld_out, rows, cols, batch, dt,
                             current_stream()), "mh_transpose")
    return out


def _col_sum(x, rows, cols, dt, batch=1, stride_in=0):
    # partial rows: one per 4 input rows at most (a sum over the BATCH has 32 rows: 256 partial rows made the launch 262 144 blocks of which
    # 8 192 had work, and its scratch 537 MB)
    npart = max(1, min(NPART, (rows + 3) // 4))
    part = torch.empty(batch * npart * cols, device=x.device, dtype=torch.float32)
    out = torch.empty(batch, cols, device=x.device, dtype=torch.float32)
    check(lib().mh_col_sum(ptr(x), x.shape[-1], rows, cols, batch, stride_in, ptr(part), npart, ptr(out), 0, dt,
                           current_stream()), "mh_col_sum")
    return out if batch > 1 else out[0]


# ---------------------------------------------------------------------------------------------- Functions
class _Cast(Function):
    """fp32 [M, C] <-> compute dtype [M, pad64(C)] (zero padded columns)."""

    @staticmethod
    def forward(ctx, x, dt, to_compute):
        ctx.dt, ctx.to_compute, ctx.cols = dt, to_compute, x.shape[1]
        if to_compute:
            return ops.cast_pad(x, ops.pad64(x.shape[1]), dt)
        raise AssertionError

    @staticmethod
    def backward(ctx, g):
        return ops.cast_to_f32(g.contiguous(), ctx.cols, ctx.dt), None, None


class _ToF32(Function):
    """compute dtype [M, ld] -> fp32 [M, C]."""

    @staticmethod
    def forward(ctx, x, cols, dt):
        ctx.dt, ctx.ld = dt, x.shape[1]
        return ops.cast_to_f32(x, cols, dt)

    @staticmethod
    def backward(ctx, g):
        return ops.cast_pad(g.contiguous(), ctx.ld, ctx.dt), None, None


class _Linear(Function):
    """y = act(x W^T + b) (+ residual).  x [M, Kp] compute dtype (padding columns zero), W [N, K] fp32, b [N] fp32.
    Output [M, pad64(N)] with zero padding columns."""

    @staticmethod
    def forward(ctx, x, W, b, act, residual, dt, drop=None, prep=None, ln=None, passthrough=False):
        # passthrough: also return x itself as a second output.  A consumer that takes THAT as its residual sends the residual branch's
        # gradient back through this node, where the input-gradient GEMM adds it in its epilogue (fp32 accumulator + residual, one
        # rounding) - instead of autograd summing the two branches with a separate bf16 add kernel per layer and step
        M, Kp = x.shape
        N, K = W.shape
        Np = ops.pad64(N)
        # prep: (W in the compute dtype [N, Kp], its transpose [Kp, Np]) made by _WeightPrep in one launch for the whole encoder
        Wc = prep[0] if prep is not None else ops.cast_pad(W.detach(), Kp, dt)                         # [N, Kp]
        ctx.WT = prep[1] if prep is not None else None
        pre = _zeros(M, Np, dt, x.device, N)
        fused_act = act is not None and dt == ops.MH_BF16 and N == Np and N % 8 == 0 and Kp % 32 == 0
        ctx.drop = drop if _active(drop) else None
        if ln is not None and b is not None and ln.usable(x, N, Kp, residual, dt, act):
            y = ln.run(x, Wc, b.detach(), residual, pre, ctx.drop)       # ... and the LayerNorm behind it (ln.y) in the same kernel
        elif ctx.drop is not None:   # dense -> dropout -> + residual in one kernel; the backward re-creates the mask
            assert act is None and N == Np
            y = _gemm_drop(x, Wc, b.detach() if b is not None else None, dt, N, Kp, pre, residual, ctx.drop)
        elif fused_act:      # one kernel writes the pre-activation (kept for the backward) and the activation
            assert residual is None
            y = torch.empty_like(pre)
            check(lib().mh_gemm_bias_act_pre(ptr(x), Kp, ptr(Wc), Kp, ptr(b.detach()) if b is not None else None, ptr(pre), ptr(y), Np,
                                             M, N, Kp, ops.ACT[act], current_stream()), "mh_gemm_bias_act_pre")
        else:
            _gemm(x, Wc, b.detach() if b is not None else None, dt, N, Kp, out=pre, residual=residual if act is None else None)
            if act is None:
                y = pre
            else:
                y = torch.empty_like(pre)
                check(lib().mh_act_fwd(ptr(pre), ptr(y), pre.numel(), ops.ACT[act], dt, current_stream()), "mh_act_fwd")
                assert residual is None
        ctx.save_for_backward(x, Wc, pre if act is not None else None)
        ctx.ln = ln
        ctx.meta = (act, dt, N, K, Kp, Np, residual is not None, b is not None)
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dx_branch=None):
        x, Wc, pre = ctx.saved_tensors
        act, dt, N, K, Kp, Np, has_res, has_b = ctx.meta
        M = x.shape[0]
        dy = dy.contiguous()
        if act is not None:
            dpre = torch.empty_like(dy)
            check(lib().mh_act_bwd(ptr(dy), ptr(pre), ptr(dpre), dy.numel(), ops.ACT[act], dt, current_stream()), "mh_act_bwd")
        elif ctx.drop is not None:
            dpre = _dropped_grad(ctx.ln, ctx.drop, dy, dt)   # the dense branch sees dY o keep / (1 - p); the residual branch sees dY
        else:
            dpre = dy
        # dX = dpre W : reduction over the N outputs
        WT = ctx.WT if ctx.WT is not None else _transpose(Wc, N, Kp, dt, ld_out=Np)                      # [Kp, Np]
        dx = _zeros(M, Kp, dt, x.device, Kp)
        _gemm(dpre, WT, None, dt, Kp, Np, out=dx, residual=None if dx_branch is None else dx_branch.contiguous())
        # dW = dpre^T X : reduction over the M rows
        dW, db = _dw(dpre, x, N, Kp, M, dt, bias=True) if has_b else (_dw(dpre, x, N, Kp, M, dt), None)
        return dx, dW[:, :K].contiguous(), db, None, (dy if has_res else None), None, None, None, None, None


class _FFN(Function):
    """y = gelu(x W1^T + b1) W2^T + b2 + x  (HF BertIntermediate + BertOutput.dense + residual) as ONE tape node, so the
    backward can fold the GELU derivative into the input-gradient GEMM of the second dense: d(pre) = (dY W2) o gelu'(pre)
    comes out of one kernel and d(activation) is never written (bf16, 64-aligned widths; otherwise two _Linear nodes)."""

    @staticmethod
    def supported(x, W1, W2, dt):
        F, H = W1.shape
        return dt == ops.MH_BF16 and x.shape[1] == H and H % 64 == 0 and F % 64 == 0 and tuple(W2.shape) == (H, F)

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, dt, drop=None, prep=None, ln=None):
        M, H = x.shape
        F = W1.shape[0]
        if prep is not None:      # (W1c, W1T, W2c, W2T) from _WeightPrep
            W1c, W2c = prep[0], prep[2]
            ctx.WT = (prep[1], prep[3])
        else:
            W1c, W2c = ops.cast_pad(W1.detach(), H, dt), ops.cast_pad(W2.detach(), F, dt)
            ctx.WT = None
        pre = torch.empty(M, F, device=x.device, dtype=x.dtype)
        f = torch.empty_like(pre)
        # GELU_DERIV_FWD: `pre` receives gelu'(pre) - from the exp / rcp pair that gelu(pre) needs anyway - and the backward multiplies
        ctx.deriv = bool(GELU_DERIV_FWD)
        fwd = lib().mh_gemm_bias_act_dact if ctx.deriv else lib().mh_gemm_bias_act_pre
        check(fwd(ptr(x), H, ptr(W1c), H, ptr(b1.detach()), ptr(pre), ptr(f), F, M, F, H, ops.ACT["gelu"], current_stream()),
              "mh_gemm_bias_act_dact" if ctx.deriv else "mh_gemm_bias_act_pre")
        y = torch.empty(M, H, device=x.device, dtype=x.dtype)
        ctx.drop = drop if _active(drop) else None
        if ln is not None and ln.usable(f, H, F, x, dt, None):
            ln.run(f, W2c, b2.detach(), x, y, ctx.drop)
        elif ctx.drop is not None:
            _gemm_drop(f, W2c, b2.detach(), dt, H, F, y, x, ctx.drop)
        else:
            _gemm(f, W2c, b2.detach(), dt, H, F, out=y, residual=x)
        ctx.save_for_backward(x, W1c, W2c, pre, f)
        ctx.ln = ln
        ctx.dt = dt
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W1c, W2c, pre, f = ctx.saved_tensors
        dt = ctx.dt
        M, H = x.shape
        F = W1c.shape[0]
        dy = dy.contiguous()
        dym = _dropped_grad(ctx.ln, ctx.drop, dy, dt) if ctx.drop is not None else dy      # gradient of the dropped dense output
        dW2, db2 = _dw(dym, f, H, F, M, dt, bias=True)
        # d(pre) = (dy W2) o gelu'(pre): W2 is [H, F]; the GEMM wants the reduction dim contiguous -> W2^T [F, H]
        dpre = torch.empty(M, F, device=x.device, dtype=x.dtype)
        W2T = ctx.WT[1] if ctx.WT is not None else _transpose(W2c, H, F, dt, ld_out=H)
        check(lib().mh_gemm_act_grad(ptr(dym), H, ptr(W2T), H, ptr(pre), F, ptr(dpre), F, M, F, H, ACT_DERIV if ctx.deriv else ops.ACT["gelu"],
                                     current_stream()), "mh_gemm_act_grad")
        dW1, db1 = _dw(dpre, x, F, H, M, dt, bias=True)
        dx = torch.empty(M, H, device=x.device, dtype=x.dtype)
        W1T = ctx.WT[0] if ctx.WT is not None else _transpose(W1c, F, H, dt, ld_out=F)                         # [H, F]
        _gemm(dpre, W1T, None, dt, H, F, out=dx, residual=dy)             # + dy: the residual branch
        return dx, dW1, db1, dW2, db2, None, None, None, None


class _Dropout(Function):
    """y = x o keep / (1 - p) on a dense activation (nn.Dropout after the embedding LayerNorm, network.py:149)."""

    @staticmethod
    def forward(ctx, x, dt, drop):
        ctx.drop, ctx.dt = drop, dt
        return drop.apply(x, dt)

    @staticmethod
    def backward(ctx, g):
        return ctx.drop.apply(g, ctx.dt), None, None


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, g, b, eps, dt, fused=None):
        if fused is not None and fused.y is not None:      # the dense node in front already normalised these rows (_FusedLN)
            y, fused.y = fused.y, None
        else:
            y = ops.layernorm(x, g.detach(), b.detach(), eps, dt)
        ctx.save_for_backward(x, g.detach())
        ctx.meta = (eps, dt)
        ctx.fused = fused if (fused is not None and fused.drop is not None) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        eps, dt = ctx.meta
        M, H = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        nb = min(1024, (M + 3) // 4)
        part = torch.empty(2 * nb * H, device=x.device, dtype=torch.float32)
        dgb = torch.empty(2, H, device=x.device, dtype=torch.float32)     # side by side: one launch folds both (mh_layernorm_bwd)
        dg, db = dgb[0], dgb[1]
        if ctx.fused is not None:      # ... and dx o keep / (1 - p) of the dense node in front, handed over through the _FusedLN
            import ctypes as C
            dxm = torch.empty_like(x)
            d = ctx.fused.drop.c()
            check(lib().mh_layernorm_bwd_drop(ptr(x), ptr(dy), ptr(g), ptr(dx), ptr(dxm), C.byref(d), ptr(part), nb, ptr(dg), ptr(db), 0, M, H,
                                              eps, dt, current_stream()), "mh_layernorm_bwd_drop")
            ctx.fused.dym = (dx, dxm)
        else:
            check(lib().mh_layernorm_bwd(ptr(x), ptr(dy), ptr(g), ptr(dx), ptr(part), nb, ptr(dg), ptr(db), 0, M, H, eps, dt,
                                         current_stream()), "mh_layernorm_bwd")
        return dx, dg, db, None, None, None


class _AddPosTime(Function):
    """pre[b,l,:] = pos[l,:] + x[b,l,:] + emb_t[b,:]   (network.py:146-148)."""

    @staticmethod
    def forward(ctx, x, pos, emb_t, B, L, dt):
        H = pos.shape[1]
        out = torch.empty(B * L, H, device=x.device, dtype=_td(dt))
        check(lib().mh_add_pos_time(ptr(x), x.shape[1], ptr(pos.detach()[:L].contiguous()), ptr(emb_t.contiguous()), ptr(out),
                                    B, L, H, dt, current_stream()), "mh_add_pos_time")
        ctx.meta = (B, L, H, dt, x.shape[1], pos.shape[0])
        return out

    @staticmethod
    def backward(ctx, g):
        B, L, H, dt, ldx, Lmax = ctx.meta
        g = g.contiguous()
        dx = g if ldx == H else torch.nn.functional.pad(g, (0, ldx - H))
        dpos_l = _col_sum(g.view(B, L * H), B, L * H, dt).view(L, H)                 # sum over the batch
        dpos = torch.zeros(Lmax, H, device=g.device, dtype=torch.float32)
        dpos[:L] = dpos_l
        demb = _col_sum(g, L, H, dt, batch=B, stride_in=L * H).reshape(B, H)          # sum over positions, per sequence
        return dx, dpos, demb, None, None, None


class _Attention(Function):
    """ctx = softmax(q k^T / sqrt(dh)) v over [M, 3H] = [q | k | v] token-major projections (HF BertSelfAttention)."""

    @staticmethod
    def _keep_bits(drop, BH, L, dev):
        """The attention site's keep-bit tensor: injected (tests) or generated by the standalone Philox kernel - bit for bit
        what the fused forward writes itself."""
        import ctypes as C
        if drop.bits is not None:
            return drop.bits
        bits = torch.empty(int(lib().mh_dropout_bits_words(BH, L)), device=dev, dtype=torch.int32)
        d = drop.c()
        check(lib().mh_dropout_bits(ptr(bits), BH, L, C.byref(d), current_stream()), "mh_dropout_bits")
        return bits

    @staticmethod
    def _probs(q, k, BH, L, Lp, dh, scale, dt):
        """P = softmax(q k^T * scale) materialised as [BH L, Lp] (padding columns zero)."""
        P = (torch.empty if Lp == L else torch.zeros)(BH * L, Lp, device=q.device, dtype=q.dtype)
        check(lib().mh_gemm_batched(ptr(q), dh, L * dh, ptr(k), dh, L * dh, None, ptr(P), Lp, L * Lp, 0, BH, L, L, dh, dt, current_stream()),
              "mh_gemm_batched")
        check(lib().mh_softmax_rows(ptr(P), BH * L, L, Lp, scale, dt, current_stream()), "mh_softmax_rows")
        return P

    @staticmethod
    def forward(ctx, qkv, B, L, nh, dt, drop=None):
        import ctypes as C
        drop = drop if _active(drop) else None
        ctx.drop, ctx.bits = drop, None
        if drop is not None and drop.ready is not None:      # keep bits drawn ahead on a side stream
            torch.cuda.current_stream().wait_event(drop.ready)
        M, ld = qkv.shape
        H = ld // 3 if ld % 3 == 0 else None
        assert H is not None
        dh = H // nh
        td = _td(dt)
        L_ = lib()
        st = current_stream()
        es = qkv.element_size()
        scale = 1.0 / math.sqrt(dh)
        fused = FUSED_ATTENTION and dt == ops.MH_BF16 and bool(L_.mh_attention_stream_bwd_supported(L, dh)) and ld == 3 * H
        vt = torch.empty(B * nh * dh * L + 256, device=qkv.device, dtype=td)     # slack: 16-B tail over-read of the last row
        slack_in_kernel = fused and L % 64 == 0 and dh in (32, 64, 128)              # (mode 4 zeroes it in the same launch)
        if not slack_in_kernel:
            vt[-256:].zero_()
        check(L_.mh_head_permute(qkv.data_ptr() + 2 * H * es, ptr(vt), ld, B, L, nh, dh, (4 if slack_in_kernel else 3) if fused else 2, dt, st),
              "mh_head_permute")
        if fused:
            # streaming forward straight off the token-major projections (rows of one head are dh-wide column blocks, pitch 3H);
            # the log-sum-exp lets the backward kernels re-create P tile by tile (no [L, L] tensor in HBM)
            out = torch.empty(B * L, H, device=qkv.device, dtype=td)
            lse = torch.empty(B * nh * L, device=qkv.device, dtype=torch.float32)
            if drop is not None:   # probability dropout inside the kernel: it reads the keep bits a pre-pass / a test wrote, or draws and writes them
                bits_in = int(drop.bits is not None)
                ctx.bits = drop.bits if bits_in else torch.empty(int(L_.mh_dropout_bits_words(B * nh, L)), device=qkv.device, dtype=torch.int32)
                d = drop.c()
                check(L_.mh_attention_stream_fwd_drop(qkv.data_ptr(), qkv.data_ptr() + H * es, ptr(vt), ptr(out), H, 0, B, L, nh, dh, scale,
                                                      ptr(lse), L * ld, dh, ld, C.byref(d), ptr(ctx.bits), bits_in, st),
                      "mh_attention_stream_fwd_drop")
            else:
                check(L_.mh_attention_stream_fwd_ex(qkv.data_ptr(), qkv.data_ptr() + H * es, ptr(vt), ptr(out), H, 0, B, L, nh, dh, scale,
                                                    ptr(lse), L * ld, dh, ld, st), "mh_attention_stream_fwd_ex")
            ctx.save_for_backward(qkv, out, lse)
        else:
            q = torch.empty(B, nh, L, dh, device=qkv.device, dtype=td)
            k = torch.empty_like(q)
            check(L_.mh_head_permute(qkv.data_ptr(), ptr(q), ld, B, L, nh, dh, 0, dt, st), "mh_head_permute")
            check(L_.mh_head_permute(qkv.data_ptr() + H * es, ptr(k), ld, B, L, nh, dh, 0, dt, st), "mh_head_permute")
            if drop is not None:
                # materialised: P -> P o keep / (1 - p) -> . V as batched GEMMs (fp32 parity mode / shapes without a streaming kernel)
                BH, Lp = B * nh, ops.pad64(L)
                ctx.bits = _Attention._keep_bits(drop, BH, L, qkv.device)
                P = _Attention._probs(q, k, BH, L, Lp, dh, scale, dt)
                check(L_.mh_dropout_bits_apply(ptr(P), Lp, ptr(ctx.bits), BH, L, drop.p, dt, st), "mh_dropout_bits_apply")
                vtp = torch.zeros(BH * dh, Lp, device=qkv.device, dtype=td)                       # V^T with the reduction dim padded
                vtp.view(BH, dh, Lp)[:, :, :L] = vt[: BH * dh * L].view(BH, dh, L)
                o4 = torch.empty(B, nh, L, dh, device=qkv.device, dtype=td)
                check(L_.mh_gemm_batched(ptr(P), Lp, L * Lp, ptr(vtp), Lp, dh * Lp, None, ptr(o4), dh, L * dh, 0, BH, L, dh, Lp, dt, st),
                      "mh_gemm_batched")
                out = torch.empty(B * L, H, device=qkv.device, dtype=td)
                check(L_.mh_head_permute(ptr(o4), ptr(out), H, B, L, nh, dh, 1, dt, st), "mh_head_permute")
            else:
                out = ops.attention(q, k, vt, scale, dt)
            ctx.save_for_backward(q, k, vt)
        ctx.meta = (B, L, nh, dh, H, dt, scale, fused)
        return out

    @staticmethod
    def _backward_fused(ctx, dctx):
        qkv, out, lse = ctx.saved_tensors
        B, L, nh, dh, H, dt, scale, _ = ctx.meta
        dev, td = qkv.device, qkv.dtype
        L_, st, es = lib(), current_stream(), qkv.element_size()
        dctx = dctx.contiguous()
        # (no transposed copies of q / k / dO: the kernels read K^T, Q^T, dO^T out of the row-layout LDS stages with transposing reads)
        D = torch.empty(B * nh * L, device=dev, dtype=torch.float32)      # scratch: rowsum(dO o O) (x (1 - p) under dropout), produced by the dQ kernel
        dqkv = torch.empty(B * L, 3 * H, device=dev, dtype=td)
        check(L_.mh_attention_stream_bwd_drop(qkv.data_ptr(), qkv.data_ptr() + H * es, qkv.data_ptr() + 2 * H * es, None, None,
                                              ptr(dctx), None, ptr(out), ptr(lse), ptr(D), dqkv.data_ptr(), dqkv.data_ptr() + H * es,
                                              dqkv.data_ptr() + 2 * H * es, 3 * H, B, L, nh, dh, scale,
                                              L * 3 * H, dh, 3 * H, L * H, dh, H, ptr(ctx.bits), ctx.drop.p if ctx.drop is not None else 0.0, st),
              "mh_attention_stream_bwd_drop")
        return dqkv, None, None, None, None, None

    @staticmethod
    def backward(ctx, dctx):
        if ctx.meta[-1]:
            return _Attention._backward_fused(ctx, dctx)
        q, k, vt = ctx.saved_tensors
        B, L, nh, dh, H, dt, scale, _ = ctx.meta
        dev, td = q.device, q.dtype
        L_ = lib()
        st = current_stream()
        BH, Lp = B * nh, ops.pad64(L)
        dctx = dctx.contiguous()
        dO = torch.empty(B, nh, L, dh, device=dev, dtype=td)
        dOT = torch.zeros(BH * dh, Lp, device=dev, dtype=td)
        check(L_.mh_head_permute(ptr(dctx), ptr(dO), H, B, L, nh, dh, 0, dt, st), "mh_head_permute")
        tmp = torch.empty(BH * dh * L, device=dev, dtype=td)
        check(L_.mh_head_permute(ptr(dctx), ptr(tmp), H, B, L, nh, dh, 2, dt, st), "mh_head_permute")
        dOT.view(BH, dh, Lp)[:, :, :L] = tmp.view(BH, dh, L)                                    # pad the reduction dim
        # P = softmax(q k^T * scale)   [BH, L, Lp]
        P = _Attention._probs(q, k, BH, L, Lp, dh, scale, dt)
        drop = ctx.drop
        Pd = P
        if drop is not None:   # the forward multiplied V by P o keep / (1 - p)
            Pd = P.clone()
            check(L_.mh_dropout_bits_apply(ptr(Pd), Lp, ptr(ctx.bits), BH, L, drop.p, dt, st), "mh_dropout_bits_apply")
        # dV = P^T dO : A = P^T [L(keys), Lp(q)], W = dO^T [dh, Lp(q)]
        PT = _transpose(Pd, L, L, dt, ld_out=Lp, batch=BH, stride_in=L * Lp, stride_out=L * Lp)
        dV = torch.empty(B, nh, L, dh, device=dev, dtype=td)
        check(L_.mh_gemm_batched(ptr(PT), Lp, L * Lp, ptr(dOT), Lp, dh * Lp, None, ptr(dV), dh, L * dh, 0, BH, L, dh, Lp, dt, st),
              "mh_gemm_batched")
        # dP = dO V^T : W = V [L(keys), dh]
        V = _transpose(vt, dh, L, dt, ld_out=dh, batch=BH, stride_in=dh * L, stride_out=L * dh, ld_in=L)
        dP = (torch.empty if Lp == L else torch.zeros)(BH * L, Lp, device=dev, dtype=td)
        check(L_.mh_gemm_batched(ptr(dO), dh, L * dh, ptr(V), dh, L * dh, None, ptr(dP), Lp, L * Lp, 0, BH, L, L, dh, dt, st),
              "mh_gemm_batched")
        if drop is not None:   # gradient w.r.t. the un-dropped probabilities
            check(L_.mh_dropout_bits_apply(ptr(dP), Lp, ptr(ctx.bits), BH, L, drop.p, dt, st), "mh_dropout_bits_apply")
        check(L_.mh_softmax_bwd_rows(ptr(P), ptr(dP), BH * L, L, Lp, scale, dt, st), "mh_softmax_bwd_rows")   # dP := dS
        # dQ = dS K : W = K^T [dh, Lp(keys)]
        KT = _transpose(k.view(BH * L, dh), L, dh, dt, ld_out=Lp, batch=BH, stride_in=L * dh, stride_out=dh * Lp)
        dQ = torch.empty(B, nh, L, dh, device=dev, dtype=td)
        check(L_.mh_gemm_batched(ptr(dP), Lp, L * Lp, ptr(KT), Lp, dh * Lp, None, ptr(dQ), dh, L * dh, 0, BH, L, dh, Lp, dt, st),
              "mh_gemm_batched")
        # dK = dS^T Q : A = dS^T [L(keys), Lp(q)], W = Q^T [dh, Lp(q)]
        dST = _transpose(dP, L, L, dt, ld_out=Lp, batch=BH, stride_in=L * Lp, stride_out=L * Lp)
        QT = _transpose(q.view(BH * L, dh), L, dh, dt, ld_out=Lp, batch=BH, stride_in=L * dh, stride_out=dh * Lp)
        dK = torch.empty(B, nh, L, dh, device=dev, dtype=td)
        check(L_.mh_gemm_batched(ptr(dST), Lp, L * Lp, ptr(QT), Lp, dh * Lp, None, ptr(dK), dh, L * dh, 0, BH, L, dh, Lp, dt, st),
              "mh_gemm_batched")
        dqkv = torch.empty(B * L, 3 * H, device=dev, dtype=td)
        es = dqkv.element_size()
        for i, t in enumerate((dQ, dK, dV)):
            check(L_.mh_head_permute(ptr(t), dqkv.data_ptr() + i * H * es, 3 * H, B, L, nh, dh, 1, dt, st), "mh_head_permute")
        return dqkv, None, None, None, None, None


class _Embed(Function):
    @staticmethod
    def forward(ctx, W, ids):
        ids32 = ids.to(torch.int32).contiguous()
        ctx.save_for_backward(ids32)
        ctx.shape = W.shape
        return ops.embed_gather(W.detach(), ids32)

    @staticmethod
    def backward(ctx, g):
        (ids32,) = ctx.saved_tensors
        V, E = ctx.shape
        dW = torch.zeros(V, E, device=g.device, dtype=torch.float32)
        g = g.contiguous()
        ws = torch.empty(int(lib().mh_scatter_add_rows_workspace_bytes(E, V)), device=g.device, dtype=torch.uint8)
        check(lib().mh_scatter_add_rows(ptr(g), ptr(ids32), ptr(dW), ids32.numel(), E, V, ptr(ws), ws.numel(), current_stream()),
              "mh_scatter_add_rows")
        return dW, None


class _QSample(Function):
    """where(mask == 0, x0, a[b] x0 + s[b] noise)   (diffusion.py:229-255); gradient flows to x0 only."""

    @staticmethod
    def forward(ctx, x0, noise, a, s, mask):
        m32 = None if mask is None else mask.to(torch.int32).contiguous()
        ctx.save_for_backward(a, m32)
        return ops.q_sample(x0, noise, a, s, m32)

    @staticmethod
    def backward(ctx, g):
        a, m32 = ctx.saved_tensors
        g = g.contiguous()
        B = g.shape[0]
        dx = torch.empty_like(g)
        check(lib().mh_scale_rows(ptr(g), ptr(a), ptr(m32), ptr(dx), 0, B, g.numel() // B, g.shape[-1], current_stream()),
              "mh_scale_rows")
        return dx, None, None, None, None


class _AxPy(Function):
    """a[b] x + s[b] y with gradients to BOTH operands: `_predict_xstart_from_eps` (diffusion.py:228-236) inside training_losses when
    the model output is read as the noise (predict_xstart=False, `_x0_helper`, diffusion.py:586-590)."""

    @staticmethod
    def forward(ctx, x, y, a, s):
        ctx.save_for_backward(a, s)
        return ops.q_sample(x, y, a, s, None)

    @staticmethod
    def backward(ctx, g):
        a, s = ctx.saved_tensors
        g = g.contiguous()
        B = g.shape[0]
        outs = []
        for k, coef in enumerate((a, s)):
            if not ctx.needs_input_grad[k]:
                outs.append(None)
                continue
            d = torch.empty_like(g)
            check(lib().mh_scale_rows(ptr(g), ptr(coef), None, ptr(d), 0, B, g.numel() // B, g.shape[-1], current_stream()), "mh_scale_rows")
            outs.append(d)
        return outs[0], outs[1], None, None


class _SqDiffMean(Function):
    """mean_flat((scale * a - b)^2) per batch row (diffusion.py:15-19 applied at :627-639)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        B = a.shape[0]
        a = a.contiguous()
        b = None if b is None else b.contiguous()
        out = torch.empty(B, device=a.device, dtype=torch.float32)
        check(lib().mh_sqdiff_mean(ptr(a), ptr(b), float(scale), ptr(out), B, a.numel() // B, current_stream()), "mh_sqdiff_mean")
        ctx.save_for_backward(a, b)
        ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        B = a.shape[0]
        g = g.contiguous()
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(a) if (b is not None and ctx.needs_input_grad[1]) else None
        if da is None and db is None:
            return None, None, None
        check(lib().mh_sqdiff_bwd(ptr(a), ptr(b), ctx.scale, ptr(g), ptr(da), ptr(db), 0, B, a.numel() // B, current_stream()),
              "mh_sqdiff_bwd")
        return da, db, None


class _TokenCE(Function):
    """Per-token cross-entropy of fp32 logits [N, V] against ids [N] (diffusion.py:556-566)."""

    @staticmethod
    def forward(ctx, logits, ids, V):
        n = logits.shape[0]
        ids32 = ids.reshape(-1).to(torch.int32).contiguous()
        loss = torch.empty(n, device=logits.device, dtype=torch.float32)
        lse = torch.empty_like(loss)
        check(lib().mh_cross_entropy_fwd(ptr(logits), logits.shape[1], ptr(ids32), ptr(loss), ptr(lse), n, V, current_stream()),
              "mh_cross_entropy_fwd")
        ctx.save_for_backward(logits, ids32, lse)
        ctx.V = V
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, ids32, lse = ctx.saved_tensors
        n, ld = logits.shape
        dl = torch.empty_like(logits)
        check(lib().mh_cross_entropy_bwd(ptr(logits), ld, ptr(ids32), ptr(lse), ptr(g.contiguous()), ptr(dl), ld, n, ctx.V, ld,
                                         MH_F32, current_stream()), "mh_cross_entropy_bwd")
        return dl, None, None


# ---------------------------------------------------------------------------------------------- composites
class _StackRows(Function):
    """The [3H, K] weight of the fused query | key | value projection as a TAPE node: with _WeightPrep's packed working copy its values
    are never read (_Linear takes the shape from it and the operand from `prep`), so nothing is copied - round 4 ran a 3 MB fp32
    torch.cat per layer and step plus its CatBackward here.  Without the working copy (fp32 mode, shapes _WeightPrep does not serve)
    the rows are concatenated as before.  The backward hands each parameter its rows of the stacked gradient (views, no copy)."""

    @staticmethod
    def forward(ctx, copy, *parts):
        ctx.rows = [int(p.shape[0]) for p in parts]
        if copy:
            return torch.cat([p.detach() for p in parts], dim=0)
        return parts[0].new_empty((sum(ctx.rows),) + tuple(parts[0].shape[1:]))

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(g.split(ctx.rows, dim=0))


class _PackVectors(Function):
    """Every layer's query | key | value bias as ONE concatenation per forward (36 small tensors, one launch; round 4: a torch.cat per
    layer): returns one [3H] view per layer; the backward splits each layer's gradient back into its three parameters (views)."""

    @staticmethod
    def forward(ctx, group, *vecs):
        ctx.sizes = [int(v.numel()) for v in vecs]
        ctx.group = int(group)
        packed = torch.cat([v.detach().reshape(-1) for v in vecs])
        per = [sum(ctx.sizes[i:i + group]) for i in range(0, len(vecs), group)]
        return tuple(packed.split(per))

    @staticmethod
    def backward(ctx, *gs):
        out = []
        for li, g in enumerate(gs):
            sz = ctx.sizes[li * ctx.group:(li + 1) * ctx.group]
            out += [None] * len(sz) if g is None else list(g.split(sz))
        return (None,) + tuple(out)


def _linear(x, lin, act, dt, residual=None, drop=None, prep=None, ln=None):
    return _Linear.apply(x, lin.weight, lin.bias, act, residual, dt, drop, prep, ln)


def _side_stream(model):
    """The model's second stream: one that really runs beside the caller's (HIP shares hardware queues between streams:
    ops.streams_overlap).  Forward: the attention keep bits are drawn on it next to the first GEMMs; backward: the panel layers fold
    their gradient partials on it under the GEMMs that follow (csrc/train_layer.hip)."""
    side = getattr(model, "_bits_stream", None)
    if side is None:
        main = torch.cuda.current_stream()
        for _ in range(6):
            side = torch.cuda.Stream()
            if ops.streams_overlap(main, side):
                break
        object.__setattr__(model, "_bits_stream", side)
    return side


class _DropSites:
    """The dropout sites of ONE training forward (reference: nn.Dropout at network.py:149; HF BertSelfAttention / BertSelfOutput /
    BertOutput dropouts inside network.py:151).  Every site gets its own Philox counter offset (call number, site index); rates
    are zero in eval mode.  `model.dropout_masks` (tests) maps site names ("emb", "l0.attn", "l0.ao", "l0.ffn", ...) to explicit
    keep masks (uint8 [N, H]; int32 keep-bit tensors for the ".attn" sites)."""

    def __init__(self, model):
        train = model.training
        self.p_emb = float(model.dropout.p) if train else 0.0
        self.p_hid = float(model.bert_hidden_dropout) if train else 0.0
        self.p_att = float(model.bert_attention_dropout) if train else 0.0
        self.masks = getattr(model, "dropout_masks", None) or {}
        self.seed = model.dropout_seed()
        self.call = model.next_dropout_call() if (self.p_emb or self.p_hid or self.p_att) else 0
        self.nsite = 0

    def pregenerate_attention_bits(self, model, B, L, dt, dev):
        """The attention masks do not depend on data (counter-based Philox): all layers' keep-bit tensors are drawn at the start of
        the forward by the standalone generator on a side stream - a VALU-only kernel next to the first GEMMs - and the streaming
        forward only READS them (the in-kernel generator needs the 8-wave geometry or spills: 260 us per layer against 128 us for
        the reader at seq_len 1024 x batch 32; tools/attn_drop_bench.py).  Only where the fused attention path will run."""
        import ctypes as C
        self.pre = {}
        nh = model.num_heads
        dh = model.hidden_size // nh
        if self.p_att <= 0.0 or not (FUSED_ATTENTION and dt == ops.MH_BF16 and bool(lib().mh_attention_stream_bwd_supported(L, dh))):
            return
        side = _side_stream(model)
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        nwords = int(lib().mh_dropout_bits_words(B * nh, L))
        for li in range(len(model.input_transformers.layer)):
            name = "l%d.attn" % li
            if name in self.masks:
                continue
            bits = torch.empty(nwords, device=dev, dtype=torch.int32)
            bits.record_stream(side)
            d = _Drop(self.p_att, self.seed, (self.call << 16) | (2 + 3 * li)).c()      # the offset site() gives this site
            with torch.cuda.stream(side):
                check(lib().mh_dropout_bits(ptr(bits), B * nh, L, C.byref(d), side.cuda_stream), "mh_dropout_bits")
                ev = torch.cuda.Event()
                ev.record(side)
            self.pre[name] = (bits, ev)

    def site(self, name, p):
        self.nsite += 1
        if p <= 0.0:
            return None
        inj = self.masks.get(name)
        if name.endswith(".attn"):
            if inj is None and name in getattr(self, "pre", {}):
                bits, ev = self.pre[name]
                assert (self.call << 16) | self.nsite == (self.call << 16) | (2 + 3 * int(name[1:].split(".")[0]))
                return _Drop(p, self.seed, (self.call << 16) | self.nsite, bits=bits, ready=ev)
            return _Drop(p, self.seed, (self.call << 16) | self.nsite, bits=inj)
        return _Drop(p, self.seed, (self.call << 16) | self.nsite, mask=None if inj is None else inj.to(torch.uint8).contiguous())


class _WeightPrep:
    """bf16 working copies (row-major and transposed) of the encoder's dense weights, refreshed by ONE launch per forward
    (mh_weight_prep) instead of a cast per dense layer in the forward and a transpose per layer in the backward.  The buffers and
    the device table live with the model; the table is rebuilt when a parameter moved (model.to(), a fresh state)."""

    def __init__(self, model, dev, panel=False):
        H, layers = model.hidden_size, list(model.input_transformers.layer)
        F = layers[0].intermediate.dense.weight.shape[0]
        bf = torch.bfloat16
        self.panel = bool(panel)     # both copies as K32 panels (the per-layer calls of csrc/train_layer.hip) instead of row-major
        self.qkv = [torch.empty(3 * H, H, device=dev, dtype=bf) for _ in layers]
        self.qkv_t = [torch.empty(H, 3 * H, device=dev, dtype=bf) for _ in layers]
        self.ao = [torch.empty(H, H, device=dev, dtype=bf) for _ in layers]
        self.ao_t = [torch.empty(H, H, device=dev, dtype=bf) for _ in layers]
        self.w1 = [torch.empty(F, H, device=dev, dtype=bf) for _ in layers]
        self.w1_t = [torch.empty(H, F, device=dev, dtype=bf) for _ in layers]
        self.w2 = [torch.empty(H, F, device=dev, dtype=bf) for _ in layers]
        self.w2_t = [torch.empty(F, H, device=dev, dtype=bf) for _ in layers]
        items, tiles = [], 0

        def add(W, dst, dst_t, ld_dst, ld_t):
            nonlocal tiles
            it = _lib.WPrepItem(W.data_ptr(), dst, dst_t, W.shape[0], W.shape[1], ld_dst, ld_t, tiles, 3 if self.panel else 0)
            tiles += (W.shape[0] // 64) * (W.shape[1] // 64)
            items.append(it)
        for li, layer in enumerate(layers):
            sa = getattr(layer.attention, "self")
            for j, lin in enumerate((sa.query, sa.key, sa.value)):      # three parameters, one [3H, H] operand
                if self.panel:   # W [3H][H] as panels over H: rows j H ..; W^T [H][3H] as panels over 3H: panels j H / 32 ..  (ld = rows of the panel buffer)
                    add(lin.weight, self.qkv[li].data_ptr() + j * H * 32 * 2, self.qkv_t[li].data_ptr() + j * H * H * 2, 3 * H, H)
                else:
                    add(lin.weight, self.qkv[li].data_ptr() + j * H * H * 2, self.qkv_t[li].data_ptr() + j * H * 2, H, 3 * H)
            if self.panel:
                add(layer.attention.output.dense.weight, self.ao[li].data_ptr(), self.ao_t[li].data_ptr(), H, H)
                add(layer.intermediate.dense.weight, self.w1[li].data_ptr(), self.w1_t[li].data_ptr(), F, H)
                add(layer.output.dense.weight, self.w2[li].data_ptr(), self.w2_t[li].data_ptr(), H, F)
                continue
            add(layer.attention.output.dense.weight, self.ao[li].data_ptr(), self.ao_t[li].data_ptr(), H, H)
            add(layer.intermediate.dense.weight, self.w1[li].data_ptr(), self.w1_t[li].data_ptr(), H, F)
            add(layer.output.dense.weight, self.w2[li].data_ptr(), self.w2_t[li].data_ptr(), F, H)
        import ctypes as C
        raw = (_lib.WPrepItem * len(items))(*items)
        self.table = torch.frombuffer(bytearray(C.string_at(C.addressof(raw), C.sizeof(raw))), dtype=torch.uint8).to(dev)
        self.n, self.tiles = len(items), tiles
        self.key = _WeightPrep.key_of(model)

    @staticmethod
    def _params(model):
        """The encoder's Parameter objects, listed once per model (walking `layer.parameters()` costs 1 ms per call at 12 layers, and the
        step asks three times): every call checks that each listed Parameter is still the object its module holds (a replaced Parameter
        rebuilds the list); storage moves (model.to(), a fresh state) are caught by the pointer key below."""
        cached = getattr(model, "_encoder_params", None)
        layers = model.input_transformers.layer
        if cached is None or cached[0] is not layers or any(m._parameters.get(k) is not q for (m, k, q) in cached[2]):
            triples = [(m, k, q) for m in layers.modules() for k, q in m._parameters.items() if q is not None]
            cached = (layers, [q for _, _, q in triples], triples)
            object.__setattr__(model, "_encoder_params", cached)
        return cached[1]

    @staticmethod
    def key_of(model):
        return tuple(p.data_ptr() for p in _WeightPrep._params(model))

    @staticmethod
    def supported(model, dt):
        layers = model.input_transformers.layer
        H = model.hidden_size
        if not (dt == ops.MH_BF16 and len(layers) > 0 and H % 64 == 0 and layers[0].intermediate.dense.weight.shape[0] % 64 == 0):
            return False
        return all(p.dtype == torch.float32 and p.is_contiguous() for p in _WeightPrep._params(model))

    @staticmethod
    def refresh(model, dt, dev, panel=False):
        """-> the model's _WeightPrep with this forward's copies queued on the current stream, or None (shapes / dtype not served)"""
        if not WEIGHT_PREP or not _WeightPrep.supported(model, dt):
            return None
        wp = getattr(model, "_weight_prep", None)
        if wp is None or wp.key != _WeightPrep.key_of(model) or wp.table.device != dev or wp.panel != bool(panel):
            wp = _WeightPrep(model, dev, panel)
            object.__setattr__(model, "_weight_prep", wp)
        check(lib().mh_weight_prep(ptr(wp.table), wp.n, wp.tiles, current_stream()), "mh_weight_prep")
        return wp


_LAYER_SCRATCH = {}


def _layer_scratch(dev, nbytes):
    """One scratch buffer per device for the backward of the panel layers (gradients in flight, split-K partials): the layers run one
    after the other on one stream, so they share it."""
    buf = _LAYER_SCRATCH.get(dev)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        _LAYER_SCRATCH[dev] = buf
    return buf


class _PackPanel(Function):
    """bf16 rows [N, H] -> K32 panels [H / 32][N][32] at the entry of the panel layers.  The panel tensor keeps the SHAPE [N, H] (autograd
    checks gradient shapes); its memory is in panel order and only _EncoderLayer reads it.  Gradients of panel tensors travel as rows."""

    @staticmethod
    def forward(ctx, x):
        N, H = x.shape
        ld = N + PANEL_LD_PAD
        out = torch.empty(ld, H, device=x.device, dtype=x.dtype)[:N]     # [H / 32][ld][32] in memory; presented as the first N "rows"
        check(lib().mh_repack_panel(ptr(x), H, ptr(out), ld, N, H, 1, current_stream()), "mh_repack_panel")
        return out

    @staticmethod
    def backward(ctx, g):
        return g


class _EncoderLayer(Function):
    """One HF BertLayer (network.py:151) of the training forward / backward as ONE tape node over mh_train_layer_fwd / _bwd: x, ctx, x1,
    gelu, gelu' and every gradient a GEMM reads live as K32 panels.  Inputs: x (panels, shape [N, H]), the stacked q | k | v weight
    (tape only) and packed bias, the other ten parameters; output y (panels, or rows for the last layer)."""

    @staticmethod
    def forward(ctx, x, cfg, wts, drops, y_panel, Wqkv, bqkv, Wao, bao, g1, be1, W1, b1, W2, b2, g2, be2):
        import ctypes as C
        B, L, H, F, nh, eps, side = cfg
        N, dev, bf = B * L, x.device, torch.bfloat16
        ld = N + PANEL_LD_PAD
        t = _lib.TrainLayer()
        t.B, t.L, t.H, t.F, t.nh, t.ln_eps, t.ld = B, L, H, F, nh, eps, ld
        for name, w in zip(("wqkv", "wqkv_t", "wao", "wao_t", "w1", "w1_t", "w2", "w2_t"), wts):
            setattr(t, name, w.data_ptr())
        vecs = [v.detach() for v in (bqkv, bao, b1, b2, g1, be1, g2, be2)]
        for name, v in zip(("bqkv", "bao", "b1", "b2", "ln1_g", "ln1_b", "ln2_g", "ln2_b"), vecs):
            assert v.dtype == torch.float32 and v.is_contiguous()
            setattr(t, name, v.data_ptr())
        d_attn, d_ao, d_ffn = drops
        keep = []
        for name, d in (("drop_attn", d_attn), ("drop_ao", d_ao), ("drop_ffn", d_ffn)):
            if d is not None:
                cd = d.c()
                setattr(t, name, cd)
                keep.append(d.mask)
        bits = None
        if d_attn is not None:
            if d_attn.ready is not None:      # keep bits drawn ahead on a side stream
                torch.cuda.current_stream().wait_event(d_attn.ready)
            t.bits_in = int(d_attn.bits is not None)
            bits = d_attn.bits if t.bits_in else torch.empty(int(lib().mh_dropout_bits_words(B * nh, L)), device=dev, dtype=torch.int32)
            t.keep_bits = bits.data_ptr()
        qkv = torch.empty(N, 3 * H, device=dev, dtype=bf)
        vt = torch.empty(N * H + 256, device=dev, dtype=bf)
        lse = torch.empty(B * nh * L, device=dev, dtype=torch.float32)
        pre1, pre2 = (torch.empty(N, H, device=dev, dtype=bf) for _ in range(2))
        ctxv, x1 = (torch.empty(ld, H, device=dev, dtype=bf) for _ in range(2))                  # panels [H / 32][ld][32]
        y = torch.empty(ld, H, device=dev, dtype=bf)[:N] if y_panel else torch.empty(N, H, device=dev, dtype=bf)
        gact, dact = (torch.empty(ld, F, device=dev, dtype=bf) for _ in range(2))
        t.x, t.y_panel = x.data_ptr(), int(y_panel)
        for name, buf in zip(("qkv", "vt", "ctx", "lse", "pre1", "x1", "g", "dact", "pre2", "y"), (qkv, vt, ctxv, lse, pre1, x1, gact, dact, pre2, y)):
            setattr(t, name, buf.data_ptr())
        check(lib().mh_train_layer_fwd(C.byref(t), current_stream()), "mh_train_layer_fwd")
        global PANEL_LAYER_CALLS
        PANEL_LAYER_CALLS += 1
        ctx.desc = t
        ctx.keep = (wts, vecs, keep, bits, x, qkv, vt, ctxv, lse, pre1, x1, gact, dact, pre2)     # what the descriptor points at
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        B, L, H, F, nh, eps, side = ctx.cfg
        N, dev = B * L, dy.device
        t = ctx.desc
        t.side_stream = side
        dy = dy.contiguous()
        dx = torch.empty(N, H, device=dev, dtype=torch.bfloat16)
        grads = torch.empty(int(lib().mh_train_layer_grad_floats(H, F)), device=dev, dtype=torch.float32)
        nbytes = int(lib().mh_train_layer_scratch_bytes(B, L, H, F, nh, t.ld))
        scratch = _layer_scratch(dev, nbytes)
        t.dy, t.dx, t.scratch, t.scratch_bytes, t.grads = dy.data_ptr(), dx.data_ptr(), scratch.data_ptr(), nbytes, grads.data_ptr()
        check(lib().mh_train_layer_bwd(C.byref(t), current_stream()), "mh_train_layer_bwd")
        sizes = [3 * H * H, 3 * H, H * H, H, F * H, F, H * F, H, H, H, H, H]
        gWqkv, gbqkv, gWao, gbao, gW1, gb1, gW2, gb2, gg1, gbe1, gg2, gbe2 = grads.split(sizes)
        return (dx, None, None, None, None, gWqkv.view(3 * H, H), gbqkv, gWao.view(H, H), gbao, gg1, gbe1, gW1.view(F, H), gb1,
                gW2.view(H, F), gb2, gg2, gbe2)


def _panel_layers_ok(model, B, L, dt):
    layers = list(model.input_transformers.layer)
    if not (PANEL_LAYERS and TAPE_STACK and WEIGHT_PREP and FUSED_ATTENTION and dt == ops.MH_BF16 and layers):
        return False
    H = model.hidden_size
    F = layers[0].intermediate.dense.weight.shape[0]
    return bool(lib().mh_train_layer_supported(B, L, H, F, model.num_heads)) and _WeightPrep.supported(model, dt)


def denoiser_forward_with_grad(model, x, timesteps):
    """TransformerNetModel.forward (network.py:131-158) with a gradient tape made of libmusehip kernels."""
    _lib.require_device(x)
    dt = ops.dtype_code(model.compute_dtype)
    if dt in ops.SPLIT_DTYPES:
        raise NotImplementedError("compute_dtype %r is a sampling mode (the forward without a tape): train in 'bf16' or 'fp32'" % (model.compute_dtype,))
    B, L, E = x.shape
    H = model.hidden_size
    dev = x.device
    # time MLP (network.py:139): sinusoid (no parameters) -> Linear -> SiLU -> Linear, kept in fp32 compute dtype
    t = timesteps.to(dev).float()
    temb = ops.timestep_embedding(t, model.hidden_t_dim, dt, ops.pad64(model.hidden_t_dim))
    h = _linear(temb, model.time_embed[0], "silu", dt)
    emb_t = _ToF32.apply(_linear(h, model.time_embed[2], None, dt), H, dt)                       # [B, H] fp32
    xin = _Cast.apply(x.reshape(B * L, E).float(), dt, True)                                      # [N, pad64(E)]
    if model.input_dims != H:
        h = _linear(xin, model.input_up_proj[0], "tanh", dt)
        h = _linear(h, model.input_up_proj[2], None, dt)
    else:
        h = xin
    sites = _DropSites(model)
    sites.pregenerate_attention_bits(model, B, L, dt, dev)
    panel = _panel_layers_ok(model, B, L, dt)
    wp = _WeightPrep.refresh(model, dt, dev, panel)
    pre = _AddPosTime.apply(h, model.position_embeddings.weight, emb_t, B, L, dt)
    X = _LayerNorm.apply(pre, model.LayerNorm.weight, model.LayerNorm.bias, model.LayerNorm.eps, dt)
    d_emb = sites.site("emb", sites.p_emb)
    if d_emb is not None:
        X = _Dropout.apply(X, dt, d_emb)                                                          # network.py:149
    selfs = [getattr(layer.attention, "self") for layer in model.input_transformers.layer]
    bqkv_all = _PackVectors.apply(3, *[lin.bias for sa in selfs for lin in (sa.query, sa.key, sa.value)]) if (selfs and TAPE_STACK) else ()
    nlayers = len(selfs)
    if panel:
        X = _PackPanel.apply(X)
        F = model.input_transformers.layer[0].intermediate.dense.weight.shape[0]
        cfg = (B, L, H, F, model.num_heads, float(model.input_transformers.layer[0].output.LayerNorm.eps),
               _side_stream(model).cuda_stream if FOLD_SIDE_STREAM else None)
    for li, layer in enumerate(model.input_transformers.layer if panel else ()):
        sa = selfs[li]
        Wqkv = _StackRows.apply(False, sa.query.weight, sa.key.weight, sa.value.weight)           # (tape only: the operand is wp's packed copy)
        drops = tuple(d if _active(d) else None for d in (sites.site("l%d.attn" % li, sites.p_att), sites.site("l%d.ao" % li, sites.p_hid),
                                                            sites.site("l%d.ffn" % li, sites.p_hid)))
        ao, lo = layer.attention.output, layer.output
        assert float(ao.LayerNorm.eps) == cfg[5] and float(lo.LayerNorm.eps) == cfg[5]
        X = _EncoderLayer.apply(X, cfg, (wp.qkv[li], wp.qkv_t[li], wp.ao[li], wp.ao_t[li], wp.w1[li], wp.w1_t[li], wp.w2[li], wp.w2_t[li]), drops,
                                li + 1 < nlayers, Wqkv, bqkv_all[li], ao.dense.weight, ao.dense.bias, ao.LayerNorm.weight, ao.LayerNorm.bias,
                                layer.intermediate.dense.weight, layer.intermediate.dense.bias, lo.dense.weight, lo.dense.bias,
                                lo.LayerNorm.weight, lo.LayerNorm.bias)
    for li, layer in enumerate(() if panel else model.input_transformers.layer):
        sa = selfs[li]
        if TAPE_STACK:
            Wqkv = _StackRows.apply(wp is None, sa.query.weight, sa.key.weight, sa.value.weight)      # (values unused when wp holds the copy)
            # (Xr is X: the attention-output dense takes it as its residual, so that branch's gradient returns through the projection's node)
            qkv, Xr = _Linear.apply(X, Wqkv, bqkv_all[li], None, None, dt, None, None if wp is None else (wp.qkv[li], wp.qkv_t[li]), None, True)   # [N, 3H]
        else:
            Wqkv = torch.cat([sa.query.weight, sa.key.weight, sa.value.weight], dim=0)
            bqkv = torch.cat([sa.query.bias, sa.key.bias, sa.value.bias], dim=0)
            qkv = _Linear.apply(X, Wqkv, bqkv, None, None, dt, None, None if wp is None else (wp.qkv[li], wp.qkv_t[li]))   # [N, 3H]
            Xr = X
        ctxv = _Attention.apply(qkv, B, L, model.num_heads, dt, sites.site("l%d.attn" % li, sites.p_att))
        ln1 = _FusedLN(layer.attention.output.LayerNorm)
        y1 = _linear(ctxv, layer.attention.output.dense, None, dt, residual=Xr, drop=sites.site("l%d.ao" % li, sites.p_hid),
                     prep=None if wp is None else (wp.ao[li], wp.ao_t[li]), ln=ln1)
        X1 = _LayerNorm.apply(y1, layer.attention.output.LayerNorm.weight, layer.attention.output.LayerNorm.bias,
                              layer.attention.output.LayerNorm.eps, dt, ln1)
        d1, d2 = layer.intermediate.dense, layer.output.dense
        d_ffn = sites.site("l%d.ffn" % li, sites.p_hid)
        ln2 = _FusedLN(layer.output.LayerNorm)
        if FUSED_FFN and (B * L) % 64 == 0 and _FFN.supported(X1, d1.weight, d2.weight, dt):
            y2 = _FFN.apply(X1, d1.weight, d1.bias, d2.weight, d2.bias, dt, d_ffn,
                            None if wp is None else (wp.w1[li], wp.w1_t[li], wp.w2[li], wp.w2_t[li]), ln2)
        else:
            f = _linear(X1, d1, "gelu", dt, prep=None if wp is None else (wp.w1[li], wp.w1_t[li]))
            y2 = _linear(f, d2, None, dt, residual=X1, drop=d_ffn, prep=None if wp is None else (wp.w2[li], wp.w2_t[li]), ln=ln2)
        X = _LayerNorm.apply(y2, layer.output.LayerNorm.weight, layer.output.LayerNorm.bias, layer.output.LayerNorm.eps, dt, ln2)
    if model.output_dims != H:
        h = _linear(X, model.output_down_proj[0], "tanh", dt)
        h = _linear(h, model.output_down_proj[2], None, dt)
    else:
        h = X
    out = _ToF32.apply(h, E, dt)
    return out.view(B, L, E).type(x.dtype)


def _token_nll(net, x, ids, mask=None):
    """_token_discrete_loss (diffusion.py:556-575): CE of get_logits(x) against ids, averaged over positions
    (mask-weighted when a mask is given).  The logits GEMM runs in fp32 on the exact-fp32 MFMA."""
    B, L, E = x.shape
    V = net.lm_head.weight.shape[0]
    xin = _Cast.apply(x.reshape(B * L, E).float(), MH_F32, True)
    logits = _Linear.apply(xin, net.lm_head.weight, net.lm_head.bias, None, None, MH_F32)         # [N, pad64(V)] fp32
    nll = _TokenCE.apply(logits, ids, V).view(B, L)
    if mask is not None:
        m = mask.to(nll.device, torch.float32)
        return (nll * m).sum(dim=-1) / m.sum(dim=-1)
    return nll.mean(dim=-1)


def training_losses(diffusion, model, t, model_kwargs, noise=None, with_corruption=False):
    """training_losses_seq2seq / ..._with_corruption (diffusion.py:594-699).  `model` may be a DDP / _WrappedModel
    shell; returns {'mse', 'nll', 'loss'} as [B] fp32 tensors attached to the kernel-level gradient tape."""
    from .models.diffusion import _device_table, unwrap_model
    assert "input_ids" in model_kwargs
    dev = t.device
    net = unwrap_model(model)
    ids = model_kwargs["input_ids"].to(dev)
    mask = model_kwargs["input_mask"].to(dev)
    x_start_mean = _Embed.apply(net.word_embedding.weight, ids)                                   # diffusion.py:608
    B = x_start_mean.shape[0]
    ones = torch.ones(B, device=dev)
    std0 = _device_table(diffusion.sqrt_one_minus_alphas_cumprod, dev)[0].expand(B).contiguous()  # diffusion.py:610-612

    def jitter(mean):                                                                              # _get_x_start, :542-554
        return _QSample.apply(mean, torch.randn_like(mean), ones, std0, None)

    x_start = jitter(x_start_mean)
    if with_corruption:
        tgt_ids = model_kwargs["correct_ids"].to(dev)
        tgt_mean = _Embed.apply(net.word_embedding.weight, tgt_ids)
        tgt_start = jitter(tgt_mean)
    else:
        tgt_ids, tgt_mean, tgt_start = ids, x_start_mean, x_start
    if noise is None:
        noise = torch.randn_like(x_start)
    a = _device_table(diffusion.sqrt_alphas_cumprod, dev)[t]
    s = _device_table(diffusion.sqrt_one_minus_alphas_cumprod, dev)[t]
    x_t = _QSample.apply(x_start, noise, a, s, mask)                                               # diffusion.py:618
    model_output = model(x_t, diffusion._scale_timesteps(t), **model_kwargs)                        # diffusion.py:624
    assert model_output.shape == x_start.shape
    if diffusion.predict_xstart:                                                                   # _x0_helper, :577-592
        pred_x0 = model_output
    else:   # the output is the noise: x0 = sqrt(1 / ab_t) x_t - sqrt(1 / ab_t - 1) eps (:228-236); t_loss below still compares the
        #     raw output with the target latent, as the reference does (:627 / :679)
        r1 = _device_table(diffusion.sqrt_recip_alphas_cumprod, dev)[t]
        r2 = _device_table(diffusion.sqrt_recipm1_alphas_cumprod, dev)[t]
        pred_x0 = _AxPy.apply(x_t, model_output, r1, -r2)
    t_loss = _SqDiffMean.apply(tgt_start, model_output, 1.0)                                       # :627 / :679
    t0_loss = _SqDiffMean.apply(tgt_mean, pred_x0, 1.0)                                            # :630 / :682
    terms = {"mse": torch.where(t == 0, t0_loss, t_loss)}
    sqrt_ab_T = float(diffusion.sqrt_alphas_cumprod[diffusion.num_timesteps - 1].astype("float32"))
    tT_loss = _SqDiffMean.apply(x_start, None, sqrt_ab_T)                                          # :634-639
    decoder_nll = _token_nll(net, x_start, ids)                                                    # :641
    terms["nll"] = _token_nll(net, pred_x0, tgt_ids, mask=mask)                                    # :642 / :694
    terms["loss"] = terms["mse"] + decoder_nll + tT_loss                                           # :645
    return terms
