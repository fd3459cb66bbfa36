"""Tensor-level wrappers over the C ABI (one Python function per kernel entry point).

PyTorch is used for device allocation and stream handles only; every function launches exactly
one libmusehip kernel on the current stream.  Inputs must be contiguous CUDA (ROCm) tensors.
"""
import ctypes as C

import torch

from ._lib import MH_BF16, MH_BF16X3, MH_F16X3, MH_F32, check, current_stream, lib, ptr, require_device

TORCH_DTYPE = {MH_F32: torch.float32, MH_BF16: torch.bfloat16}
ACT = {None: 0, "none": 0, "tanh": 1, "gelu": 2, "silu": 3}


SPLIT_DTYPES = (MH_BF16X3, MH_F16X3)   # split precision (csrc/split.hip): the denoiser forward only


def dtype_code(dtype):
    if isinstance(dtype, int) and dtype in (MH_F32, MH_BF16, MH_BF16X3, MH_F16X3):
        return dtype
    if dtype in (torch.float32, "fp32", "f32", "float32"):
        return MH_F32
    if dtype in (torch.bfloat16, "bf16", "bfloat16"):
        return MH_BF16
    if dtype == "bf16x3":
        return MH_BF16X3
    if dtype == "f16x3":
        return MH_F16X3
    raise ValueError("compute dtype must be fp32, bf16, bf16x3 or f16x3, got %r" % (dtype,))


def to_device_async(x, device):
    """A host tensor to the device without stalling the host behind the GPU's queue: `tensor.to(device)` from pageable memory waits
    for the stream to drain (one bubble per copy, 15 of them per training step); a pinned staging copy + non_blocking does not - the
    caching host allocator keeps the staging block until the copy has run."""
    device = torch.device(device)
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x)
    if x.device == device or device.type != "cuda" or x.is_cuda:
        return x.to(device)
    return x.pin_memory().to(device, non_blocking=True)


def pad64(n):
    return (int(n) + 63) // 64 * 64


def _c(t, dtype=None):
    require_device(t)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


def cast_pad(x, ld_out, dtype, rows_out=None):
    x = _c(x, torch.float32)
    rows, cols = x.shape
    rows_out = rows if rows_out is None else rows_out
    out = torch.empty(rows_out, ld_out, device=x.device, dtype=TORCH_DTYPE[dtype])
    check(lib().mh_cast_pad(ptr(x), cols, ptr(out), ld_out, rows, cols, rows_out, dtype, current_stream()), "mh_cast_pad")
    return out


def cast_to_f32(x, cols, dtype):
    x = _c(x)
    rows, ld = x.shape
    out = torch.empty(rows, cols, device=x.device, dtype=torch.float32)
    check(lib().mh_cast_to_f32(ptr(x), ld, ptr(out), cols, rows, cols, dtype, current_stream()), "mh_cast_to_f32")
    return out


def row_sqnorm(table):
    table = _c(table, torch.float32)
    V, E = table.shape
    out = torch.empty(V, device=table.device, dtype=torch.float32)
    check(lib().mh_row_sqnorm(ptr(table), ptr(out), V, E, current_stream()), "mh_row_sqnorm")
    return out


def embed_gather(table, ids):
    table = _c(table, torch.float32)
    ids32 = _c(ids, torch.int32)
    V, E = table.shape
    out = torch.empty(*ids.shape, E, device=table.device, dtype=torch.float32)
    check(lib().mh_embed_gather(ptr(table), ptr(ids32), ptr(out), ids32.numel(), E, V, current_stream()), "mh_embed_gather")
    return out


def timestep_embedding(t, dim, dtype=MH_F32, ld_out=None, max_period=10000.0):
    t = _c(t, torch.float32)
    B = t.numel()
    ld_out = dim if ld_out is None else ld_out
    out = torch.empty(B, ld_out, device=t.device, dtype=TORCH_DTYPE[dtype])
    check(lib().mh_timestep_embedding(ptr(t), ptr(out), B, dim, ld_out, float(max_period), dtype, current_stream()),
          "mh_timestep_embedding")
    return out


def gemm_bias_act(A, W, bias=None, residual=None, act=None, dtype=MH_F32, out_f32=False, N=None, K=None, out=None):
    """out[M,N] = act(A[:, :K] W[:N, :K]^T + bias) (+ residual).  A [M, lda], W [rows>=N, ldw]."""
    A, W = _c(A), _c(W)
    M, lda = A.shape
    N = W.shape[0] if N is None else N
    K = W.shape[1] if K is None else K
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32 if out_f32 else TORCH_DTYPE[dtype])
    ldr = residual.shape[1] if residual is not None else 0
    check(lib().mh_gemm_bias_act(ptr(A), lda, ptr(W), W.shape[1], ptr(bias), ptr(residual), ldr, ptr(out),
                                 out.shape[1], int(bool(out_f32)), M, N, K, ACT[act], dtype, current_stream()),
          "mh_gemm_bias_act")
    return out


def gemm_qkv(A, Wqkv, bqkv, B, L, nh, dtype):
    A, Wqkv = _c(A), _c(Wqkv)
    H = Wqkv.shape[0] // 3
    dh = H // nh
    td = TORCH_DTYPE[dtype]
    q = torch.empty(B, nh, L, dh, device=A.device, dtype=td)
    k = torch.empty_like(q)
    vt = torch.empty(B * nh * dh * L + 128, device=A.device, dtype=td)  # slack for the 16-B tail over-read
    check(lib().mh_gemm_qkv(ptr(A), A.shape[1], ptr(Wqkv), Wqkv.shape[1], ptr(bqkv), ptr(q), ptr(k), ptr(vt), B, L, H,
                            nh, dtype, current_stream()), "mh_gemm_qkv")
    return q, k, vt[: B * nh * dh * L].view(B, nh, dh, L)


def attention(q, k, vt, scale, dtype):
    B, nh, L, dh = q.shape
    ctx = torch.empty(B * L, nh * dh, device=q.device, dtype=TORCH_DTYPE[dtype])
    check(lib().mh_attention_fwd(ptr(q), ptr(k), ptr(vt), ptr(ctx), nh * dh, B, L, nh, dh, float(scale), dtype,
                                 current_stream()), "mh_attention_fwd")
    return ctx


def layernorm(x, gamma, beta, eps, dtype):
    x = _c(x)
    rows, H = x.shape
    out = torch.empty_like(x)
    check(lib().mh_layernorm(ptr(x), ptr(gamma), ptr(beta), ptr(out), rows, H, float(eps), dtype, current_stream()),
          "mh_layernorm")
    return out


def add_pos_time_layernorm(x, pos, emb_t, emb_row, gamma, beta, B, L, eps, dtype):
    x = _c(x)
    H = pos.shape[1]
    # fp32 latent fed straight in (E == H); in the f32 mode the element type already is float
    x_is_f32 = int(x.dtype == torch.float32 and dtype == MH_BF16)
    out = torch.empty(B * L, H, device=x.device, dtype=TORCH_DTYPE[dtype])
    check(lib().mh_add_pos_time_layernorm(ptr(x), x.shape[-1], x_is_f32, ptr(pos), ptr(emb_t), ptr(emb_row), ptr(gamma),
                                          ptr(beta), ptr(out), B, L, H, float(eps), dtype, current_stream()),
          "mh_add_pos_time_layernorm")
    return out


def pad_table16(table):
    """[V, E] fp32 -> [V, E_pad16] (zero columns); the table itself when E % 16 == 0."""
    table = _c(table, torch.float32)
    E = table.shape[1]
    Ep = (E + 15) // 16 * 16
    return table if Ep == E else cast_pad(table, Ep, MH_F32)


def round_workspace(n_tokens, E, V, device):
    return torch.empty(int(lib().mh_round_workspace_bytes(n_tokens, E, V)), dtype=torch.uint8, device=device)


def round_to_embedding(x, table, table_norm=None, table_pad=None, workspace=None, out=None, mfma=True):
    """idx[n] = nearest embedding row of x[n] (models/rounding.py:21-28), int32.  mfma=True runs the
    exact-fp32 MFMA GEMM with the fused arg-best epilogue, mfma=False the plain fp32 VALU kernel."""
    x, table = _c(x, torch.float32), _c(table, torch.float32)
    V, E = table.shape
    n = x.numel() // E
    if table_norm is None:
        table_norm = row_sqnorm(table)
    idx = torch.empty(n, device=x.device, dtype=torch.int32) if out is None else out
    if mfma:
        table_pad = pad_table16(table) if table_pad is None else table_pad
        ws = round_workspace(n, E, V, x.device) if workspace is None else workspace
        check(lib().mh_round_to_embedding_mfma(ptr(x), ptr(table_pad), ptr(table_norm), ptr(idx), n, E, V, ptr(ws),
                                               ws.numel(), current_stream()), "mh_round_to_embedding_mfma")
    else:
        check(lib().mh_round_to_embedding(ptr(x), ptr(table), ptr(table_norm), ptr(idx), n, E, V, current_stream()),
              "mh_round_to_embedding")
    return idx


def logits_argmax(x, table, bias):
    x, table, bias = _c(x, torch.float32), _c(table, torch.float32), _c(bias, torch.float32)
    V, E = table.shape
    n = x.numel() // E
    idx = torch.empty(n, device=x.device, dtype=torch.int32)
    check(lib().mh_logits_argmax(ptr(x), ptr(table), ptr(bias), ptr(idx), n, E, V, current_stream()), "mh_logits_argmax")
    return idx


def _mask_args(mask, x):
    """mask -> (int32 tensor, per_elem flag).  Accepts [B,L] (per token) or x-shaped (per element)."""
    if mask is None:
        return None, 0
    if mask.dim() == x.dim() and mask.shape == x.shape and mask.stride(-1) == 0:
        mask = mask[..., 0]  # broadcast view of a per-token mask (run/sample.py:186): keep it per token
    m = _c(mask, torch.int32)
    if m.numel() == x.numel():
        return m, 1
    if m.numel() * x.shape[-1] == x.numel():
        return m, 0
    raise ValueError("mask shape %s does not match latent %s" % (tuple(mask.shape), tuple(x.shape)))


def q_sample(x0, noise, a, s, mask=None):
    x0, noise = _c(x0, torch.float32), _c(noise, torch.float32)
    a, s = _c(a, torch.float32), _c(s, torch.float32)
    B = x0.shape[0]
    m, per_elem = _mask_args(mask, x0)
    out = torch.empty_like(x0)
    check(lib().mh_q_sample(ptr(x0), ptr(noise), ptr(a), ptr(s), ptr(m), per_elem, ptr(out), B, x0.numel() // B,
                            x0.shape[-1], current_stream()), "mh_q_sample")
    return out


def make_coef(rows, device):
    """rows: list of 7-tuples (coef1, coef2, sigma, recip, recipm1, sqrt_abp, dir) -> device [n,8] fp32."""
    t = torch.zeros(len(rows), 8, dtype=torch.float32)
    t[:, :7] = torch.tensor(rows, dtype=torch.float32)
    return t.to(device)


def step_epilogue(kind, model_out, x_t, noise, coef, coef_per_batch, clip, round_idx=None, table=None, mask=None,
                  x_start=None, out=None, want_x0=True, want_mean=False):
    """Fused p_sample ('p') / ddim ('ddim') tail.  Returns (sample, pred_xstart, mean)."""
    x_t = _c(x_t, torch.float32)
    B, E = x_t.shape[0], x_t.shape[-1]
    m, per_elem = _mask_args(mask, x_t)
    out = torch.empty_like(x_t) if out is None else out
    x0 = torch.empty_like(x_t) if want_x0 else None
    mean = torch.empty_like(x_t) if (want_mean and kind == "p") else None
    args = [ptr(model_out), ptr(x_t), ptr(noise), ptr(round_idx), ptr(table), ptr(coef), int(coef_per_batch), int(bool(clip)),
            ptr(m), per_elem, ptr(x_start), ptr(out), ptr(x0)]
    if kind == "p":
        check(lib().mh_p_sample_epilogue(*args, ptr(mean), B, x_t.numel() // B, E, current_stream()), "mh_p_sample_epilogue")
    else:
        check(lib().mh_ddim_epilogue(*args, B, x_t.numel() // B, E, current_stream()), "mh_ddim_epilogue")
    return out, x0, mean


def trunc_normal(shape, bound, seed, stream_id=0, step_counter=None, device="cuda", out=None):
    out = torch.empty(shape, device=device, dtype=torch.float32) if out is None else out
    check(lib().mh_trunc_normal(ptr(out), out.numel(), float(bound or 0.0), int(seed) & (2 ** 64 - 1), int(stream_id),
                                ptr(step_counter), current_stream()), "mh_trunc_normal")
    return out


def streams_overlap(a, b, microseconds=120):
    """True when work on streams a and b runs CONCURRENTLY.  HIP multiplexes a process's streams onto a few hardware queues and two
    streams on one queue run back to back (torch's default stream against every fourth pool stream: 4.44 instead of 3.55 ms per step for
    two batch-slice chains, tools/debug/chain_streams.py): a one-wave delay kernel on each, bracketed by events.  Only a and b are
    waited for - other streams of the process (a training loop that also samples, a second sampling loop) keep running."""
    for st in (a, b):        # (a stream's first launch pays a one-time setup of a few hundred microseconds: not part of the measurement)
        check(lib().mh_stream_delay(1, st.cuda_stream), "mh_stream_delay")
    a.synchronize()
    b.synchronize()
    e0, e1, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
    e0.record(a)
    b.wait_event(e0)
    check(lib().mh_stream_delay(int(microseconds), a.cuda_stream), "mh_stream_delay")
    check(lib().mh_stream_delay(int(microseconds), b.cuda_stream), "mh_stream_delay")
    eb.record(b)
    a.wait_event(eb)
    e1.record(a)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 < 1.6 * microseconds


_STREAM_SETS = {}     # (device index, n) -> streams a probe has seen overlap: one probe per process and device, not one per loop


def concurrent_streams(n, device=None, tries=8, reprobe=False):
    """n torch streams that pairwise run concurrently (see streams_overlap); after `tries` replacements the last candidates are returned
    as they are - the results do not depend on it, only the overlap does.  The verified set is kept per device: later callers get the
    same streams without another probe (`reprobe=True` measures again)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(n))
    if not reprobe and key in _STREAM_SETS:
        return list(_STREAM_SETS[key])
    streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
    for _ in range(tries):
        clash = next(((i, j) for i in range(n) for j in range(i + 1, n) if not streams_overlap(streams[i], streams[j])), None)
        if clash is None:
            break
        streams[clash[1]] = torch.cuda.Stream(device=dev)
    _STREAM_SETS[key] = list(streams)
    return streams


class Graph:
    """A captured sequence of libmusehip launches (hipGraph) that can be replayed."""

    def __init__(self):
        self.handle = C.c_void_p()

    def capture(self, fn, stream):
        """Run fn() with `stream` capturing; everything fn launches on it becomes the graph."""
        check(lib().mh_graph_begin_capture(stream.cuda_stream), "mh_graph_begin_capture")
        try:
            with torch.cuda.stream(stream):
                fn()
        except BaseException:
            # leave capture mode, but let fn()'s own exception through (an end-capture error here would mask it)
            lib().mh_graph_end_capture(stream.cuda_stream, C.byref(self.handle))
            if self.handle:
                lib().mh_graph_destroy(self.handle)
                self.handle = C.c_void_p()
            raise
        check(lib().mh_graph_end_capture(stream.cuda_stream, C.byref(self.handle)), "mh_graph_end_capture")
        return self

    def launch(self, stream):
        check(lib().mh_graph_launch(self.handle, stream.cuda_stream), "mh_graph_launch")

    def __del__(self):
        try:
            if self.handle:
                lib().mh_graph_destroy(self.handle)
        except Exception:
            pass
