"""Weight arena + denoiser engine: the host object behind TransformerNetModel.forward.

The reference keeps ~200 separate fp32 parameter tensors and walks them through
torch ops; here they are packed ONCE into one contiguous device arena laid out for the kernels
(nn.Linear weights [out,in] cast to the compute dtype with K zero-padded to 64, Q/K/V stacked into one
[3H,H] matrix, fp32 vectors as they are).  The arena is a single `torch.uint8` tensor, which also
makes it the unit of the multi-GPU weight broadcast (one RCCL broadcast instead of the reference's
per-tensor `sync_params`, utils/dist_util.py:141-152).
"""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import MH_BF16, Denoiser, LayerWeights, check, current_stream, lib, ptr


def _align(n, a=256):
    return (n + a - 1) // a * a


class DenoiserEngine:
    """Packed weights + scratch for `mh_denoiser_forward` (models/network.py:131-158)."""

    def __init__(self, cfg, dtype, device, panel=True):
        # cfg: dict(E, H, F, nh, nL, Tt, L_max)
        self.cfg = dict(cfg)
        self.dtype = ops.dtype_code(dtype)
        self.device = torch.device(device)
        self.split = self.dtype in ops.SPLIT_DTYPES      # split precision: weights as hi + lo 16-bit panels, 4 bytes per element like fp32
        self.es = 2 if self.dtype == MH_BF16 else 4
        c = self.cfg
        c["E_pad"], c["Tt_pad"], c["T4_pad"] = ops.pad64(c["E"]), ops.pad64(c["Tt"]), ops.pad64(4 * c["Tt"])
        c["has_proj"] = int(c["E"] != c["H"])
        # K32-panel layout for every bf16 weight / activation when the shapes allow it
        dh = c["H"] // c["nh"]
        # (E % 4: the released checkpoints' E = 500 stays on this path - E is zero-padded to 64 inside the arena / input panels)
        c["panel"] = int(self.dtype == MH_BF16 and panel and c["E"] % 4 == 0 and dh % 32 == 0 and
                         c["H"] // 32 in (2, 4, 8, 12, 16, 24))
        # deferred-LayerNorm operands (folded weights + two vectors per consumer: +58 % matrix bytes at config 2) are planned and packed
        # only where the forward uses them: widths without a full-row LayerNorm epilogue (d_model 768), or when mode 2 (A/B: deferred
        # everywhere) is already switched on at construction
        c["fold_ln"] = int(c["panel"] and (not lib().mh_gemm_bias_res_ln_supported(c["H"]) or lib().mh_denoiser_get_defer_ln() == 2))
        self._plan = self._make_plan()
        self.arena = torch.zeros(self._plan["total"], dtype=torch.uint8, device=self.device)
        self._ws = None
        self._layers = (LayerWeights * max(1, c["nL"]))()
        self._desc = Denoiser()
        self._fill_descriptor()

    # ------------------------------------------------------------------ arena layout
    def _make_plan(self):
        c, es = self.cfg, self.es
        H, F, E, Tt = c["H"], c["F"], c["E"], c["Tt"]
        plan, off = {}, 0

        def mat(name, rows, kpad):          # compute-dtype matrix [rows, kpad]
            nonlocal off
            plan[name] = ("mat", off, rows, kpad)
            off = _align(off + rows * kpad * es)

        def vec(name, n):                   # fp32 vector / table
            nonlocal off
            plan[name] = ("vec", off, n, 0)
            off = _align(off + n * 4)

        mat("w_t0", 4 * Tt, c["Tt_pad"]); vec("b_t0", 4 * Tt)
        mat("w_t2", H, c["T4_pad"]); vec("b_t2", H)
        if c["has_proj"]:
            mat("w_up0", H, c["E_pad"]); vec("b_up0", H)
            mat("w_up2", H, H); vec("b_up2", H)
            mat("w_dn0", H, H); vec("b_dn0", H)
            mat("w_dn2", E, H); vec("b_dn2", E)
        vec("pos", c["L_max"] * H); vec("ln0_g", H); vec("ln0_b", H)
        for l in range(c["nL"]):
            p = "l%d." % l
            mat(p + "w_qkv", 3 * H, H); vec(p + "b_qkv", 3 * H)
            mat(p + "w_ao", H, H); vec(p + "b_ao", H)
            vec(p + "ln1_g", H); vec(p + "ln1_b", H)
            mat(p + "w_ff1", F, H); vec(p + "b_ff1", F)
            mat(p + "w_ff2", H, F); vec(p + "b_ff2", H)
            vec(p + "ln2_g", H); vec(p + "ln2_b", H)
            if c["fold_ln"]:   # deferred LayerNorm (csrc/gemm.hip DeferArgs): weights folded with the gain of the LayerNorm feeding them
                if l > 0:
                    mat(p + "w_qkv_f", 3 * H, H); vec(p + "c1_qkv", 3 * H); vec(p + "c2_qkv", 3 * H)
                mat(p + "w_ff1_f", F, H); vec(p + "c1_ff1", F); vec(p + "c2_ff1", F)
        plan["total"] = off
        return plan

    def _addr(self, name):
        return self.arena.data_ptr() + self._plan[name][1]

    def _fill_descriptor(self):
        c, d = self.cfg, self._desc
        d.dtype = self.dtype
        for k in ("E", "H", "F", "nh", "nL", "Tt", "Tt_pad", "T4_pad", "E_pad", "L_max", "has_proj", "panel"):
            setattr(d, k, int(c[k]))
        d.ln_eps = float(c.get("ln_eps", 1e-12))
        for k in ("w_t0", "b_t0", "w_t2", "b_t2", "pos", "ln0_g", "ln0_b"):
            setattr(d, k, self._addr(k))
        for k in ("w_up0", "b_up0", "w_up2", "b_up2", "w_dn0", "b_dn0", "w_dn2", "b_dn2"):
            setattr(d, k, self._addr(k) if c["has_proj"] else None)
        for l in range(c["nL"]):
            for k, _ in LayerWeights._fields_:
                name = "l%d.%s" % (l, k)
                setattr(self._layers[l], k, self._addr(name) if name in self._plan else None)
        d.layers = C.cast(self._layers, C.POINTER(LayerWeights))

    # ------------------------------------------------------------------ packing
    def _put_mat(self, name, w):
        _, off, rows, kpad = self._plan[name]
        w = w.detach().to(self.device, torch.float32).contiguous()
        r, k = w.shape
        assert r == rows and k <= kpad, (name, w.shape, rows, kpad)
        dst = self.arena.data_ptr() + off
        if self.split:       # (the time MLP stays fp32 row-major: it runs once per table build)
            if name.startswith("w_t"):
                check(lib().mh_cast_pad(ptr(w), k, dst, kpad, r, k, r, _lib.MH_F32, current_stream()), "mh_cast_pad")
            else:
                check(lib().mh_split_pack(ptr(w), k, dst, rows, r, k, kpad, self.dtype, current_stream()), "mh_split_pack")
        elif self.cfg["panel"] and not name.startswith("w_t"):   # time MLP runs once per table build: row-major kernel
            check(lib().mh_pack_panel(ptr(w), k, dst, rows, r, k, kpad, current_stream()), "mh_pack_panel")
        else:
            check(lib().mh_cast_pad(ptr(w), k, dst, kpad, r, k, r, self.dtype, current_stream()), "mh_cast_pad")

    def _put_vec(self, name, v, elem0=0):
        _, off, n, _ = self._plan[name]
        v = v.detach().to(self.device, torch.float32).reshape(-1)
        view = self.arena[off + elem0 * 4: off + (elem0 + v.numel()) * 4].view(torch.float32)
        view.copy_(v)

    def load_state_dict(self, sd):
        """Pack a reference-format state_dict (key names of SURVEY.md §3.4) into the arena."""
        c = self.cfg
        H = c["H"]
        self._put_mat("w_t0", sd["time_embed.0.weight"]); self._put_vec("b_t0", sd["time_embed.0.bias"])
        self._put_mat("w_t2", sd["time_embed.2.weight"]); self._put_vec("b_t2", sd["time_embed.2.bias"])
        if c["has_proj"]:
            self._put_mat("w_up0", sd["input_up_proj.0.weight"]); self._put_vec("b_up0", sd["input_up_proj.0.bias"])
            self._put_mat("w_up2", sd["input_up_proj.2.weight"]); self._put_vec("b_up2", sd["input_up_proj.2.bias"])
            self._put_mat("w_dn0", sd["output_down_proj.0.weight"]); self._put_vec("b_dn0", sd["output_down_proj.0.bias"])
            self._put_mat("w_dn2", sd["output_down_proj.2.weight"]); self._put_vec("b_dn2", sd["output_down_proj.2.bias"])
        self._put_vec("pos", sd["position_embeddings.weight"][: c["L_max"]])
        self._put_vec("ln0_g", sd["LayerNorm.weight"]); self._put_vec("ln0_b", sd["LayerNorm.bias"])
        for l in range(c["nL"]):
            s, p = "input_transformers.layer.%d." % l, "l%d." % l
            qkv = [sd[s + "attention.self.%s.weight" % nm].detach().to(self.device, torch.float32) for nm in ("query", "key", "value")]
            self._put_mat(p + "w_qkv", torch.cat(qkv, dim=0))          # one [3H, H] matrix: a single projection GEMM
            for i, nm in enumerate(("query", "key", "value")):
                self._put_vec(p + "b_qkv", sd[s + "attention.self.%s.bias" % nm], elem0=i * H)
            self._put_mat(p + "w_ao", sd[s + "attention.output.dense.weight"])
            self._put_vec(p + "b_ao", sd[s + "attention.output.dense.bias"])
            self._put_vec(p + "ln1_g", sd[s + "attention.output.LayerNorm.weight"])
            self._put_vec(p + "ln1_b", sd[s + "attention.output.LayerNorm.bias"])
            self._put_mat(p + "w_ff1", sd[s + "intermediate.dense.weight"]); self._put_vec(p + "b_ff1", sd[s + "intermediate.dense.bias"])
            self._put_mat(p + "w_ff2", sd[s + "output.dense.weight"]); self._put_vec(p + "b_ff2", sd[s + "output.dense.bias"])
            self._put_vec(p + "ln2_g", sd[s + "output.LayerNorm.weight"])
            self._put_vec(p + "ln2_b", sd[s + "output.LayerNorm.bias"])
            if c["fold_ln"]:
                f32 = lambda t: t.detach().to(self.device, torch.float32)
                if l > 0:
                    ps = "input_transformers.layer.%d." % (l - 1)
                    bq = torch.cat([f32(sd[s + "attention.self.%s.bias" % nm]) for nm in ("query", "key", "value")])
                    self._put_folded(p, "qkv", torch.cat(qkv, dim=0), bq, f32(sd[ps + "output.LayerNorm.weight"]), f32(sd[ps + "output.LayerNorm.bias"]))
                self._put_folded(p, "ff1", f32(sd[s + "intermediate.dense.weight"]), f32(sd[s + "intermediate.dense.bias"]),
                                 f32(sd[s + "attention.output.LayerNorm.weight"]), f32(sd[s + "attention.output.LayerNorm.bias"]))
        return self

    def _put_folded(self, p, nm, W, b, gamma, beta):
        """Deferred LayerNorm operands of a dense layer fed by LN(y) = (y - mean) rstd gamma + beta:  W' = gamma o W (rounded to the
        compute dtype FIRST, so that c1 = row sums of exactly the matrix the MFMA multiplies), c1 = sum_k W'[:, k],
        c2 = sum_k beta_k W[:, k] + b.  One-time weight preprocessing, like the casts."""
        Wf = (W * gamma[None, :]).to(ops.TORCH_DTYPE[self.dtype]).to(torch.float32)
        self._put_mat(p + "w_%s_f" % nm, Wf)
        self._put_vec(p + "c1_" + nm, Wf.sum(dim=1))
        self._put_vec(p + "c2_" + nm, (W * beta[None, :]).sum(dim=1) + b)

    # ------------------------------------------------------------------ execution
    def _workspace(self, B, L):
        need = int(lib().mh_denoiser_workspace_bytes(C.byref(self._desc), B, L))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def new_workspace(self, B, L):
        """A private scratch buffer (concurrent forwards on different streams must not share one)."""
        return torch.empty(int(lib().mh_denoiser_workspace_bytes(C.byref(self._desc), B, L)), dtype=torch.uint8, device=self.device)

    def reserve(self, B, L):
        """Allocate scratch up front (call before hipGraph capture: nothing may allocate inside)."""
        self._workspace(B, L)
        return self

    def time_embed(self, t, out=None):
        """emb_t = time_embed(timestep_embedding(t)) -> [B,H] fp32   (models/network.py:139)."""
        t = t.to(self.device, torch.float32).contiguous()
        B = t.numel()
        out = torch.empty(B, self.cfg["H"], device=self.device, dtype=torch.float32) if out is None else out
        ws = self._workspace(max(B, 1), 8)
        check(lib().mh_time_embed(C.byref(self._desc), ptr(t), ptr(out), B, ptr(ws), ws.numel(), current_stream()),
              "mh_time_embed")
        return out

    # ------------------------------------------------------------------ the forward in phases (bf16 panel models with projections)
    def phases_supported(self):
        return bool(lib().mh_denoiser_phases_supported(C.byref(self._desc)))

    def new_rows(self, n_rows):
        """A K32-panel activation buffer [H / 32][n_rows][32] bf16 for head / layers / tail to hand rows over in."""
        return torch.empty(self.cfg["H"] // 32, n_rows, 32, dtype=torch.bfloat16, device=self.device)

    @staticmethod
    def _window(rows, first):
        """(pointer, ld) of the row window of a panel buffer that starts at row `first`"""
        return rows.data_ptr() + first * 32 * rows.element_size(), rows.shape[1]

    def head(self, x, emb_t, emb_row, rows, first, ws):
        """network.py:141-149 for the batch x [B, L, E]: its B L normalised hidden rows -> rows[:, first : first + B L, :]"""
        B, L, _ = x.shape
        p, ld = self._window(rows, first)
        check(lib().mh_denoiser_head(C.byref(self._desc), ptr(x), ptr(emb_t), ptr(emb_row), p, ld, B, L, ptr(ws), ws.numel(), current_stream()),
              "mh_denoiser_head")

    def layers(self, rows_in, first_in, rows_out, first_out, B, L, ws):
        """network.py:151 (the encoder) on the B L rows starting at first_in of rows_in -> rows_out from first_out"""
        pi, ldi = self._window(rows_in, first_in)
        po, ldo = self._window(rows_out, first_out)
        check(lib().mh_denoiser_layers(C.byref(self._desc), pi, ldi, po, ldo, B, L, ptr(ws), ws.numel(), current_stream()), "mh_denoiser_layers")

    def tail(self, rows, first, out, ws):
        """network.py:153-157: down-projection of the B L rows from `first` -> out [B, L, E] fp32"""
        B, L, _ = out.shape
        p, ld = self._window(rows, first)
        check(lib().mh_denoiser_tail(C.byref(self._desc), p, ld, ptr(out), B, L, ptr(ws), ws.numel(), current_stream()), "mh_denoiser_tail")

    def forward(self, x, emb_t, emb_row=None, out=None, ws=None, sqnorm=None, round_to=None):
        """x [B,L,E] fp32 -> [B,L,E] fp32 (models/network.py:131-158).  emb_t [*,H] fp32,
        emb_row [B] int32 selecting the emb_t row of each batch element (None: row b).
        sqnorm [B L] fp32 (optional, only when mh_denoiser_gives_sqnorm): receives |out row|^2 per token."""
        _lib.require_device(x, emb_t, emb_row)
        B, L, E = x.shape
        if E != self.cfg["E"]:
            raise ValueError("latent width %d != model input_dims %d" % (E, self.cfg["E"]))
        x = x.to(torch.float32).contiguous()
        out = torch.empty_like(x) if out is None else out
        ws = self._workspace(B, L) if ws is None else ws
        if round_to is not None:       # (split table buffer, V, idx_out [B L] int32[, mh_step_update]): the last kernel also rounds its rows [and steps them]
            tsplit, V, idx = round_to[:3]
            upd = round_to[3] if len(round_to) > 3 else None
            check(lib().mh_denoiser_forward_round(C.byref(self._desc), ptr(x), ptr(emb_t), ptr(emb_row), ptr(out), ptr(tsplit), int(V), ptr(idx),
                                                  C.byref(upd) if upd is not None else None, B, L, ptr(ws), ws.numel(), current_stream()),
                  "mh_denoiser_forward_round")
            return out
        if sqnorm is not None:
            check(lib().mh_denoiser_forward_sqnorm(C.byref(self._desc), ptr(x), ptr(emb_t), ptr(emb_row), ptr(out), ptr(sqnorm), B, L, ptr(ws),
                                                   ws.numel(), current_stream()), "mh_denoiser_forward_sqnorm")
            return out
        check(lib().mh_denoiser_forward(C.byref(self._desc), ptr(x), ptr(emb_t), ptr(emb_row), ptr(out), B, L, ptr(ws),
                                        ws.numel(), current_stream()), "mh_denoiser_forward")
        return out

    def arena_bytes(self):
        return self.arena.numel()
