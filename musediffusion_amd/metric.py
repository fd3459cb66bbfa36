"""MSIM / 1NNC of MuseDiffusion/metric.py:4-117 for whole batches on the device.

The reference walks every sequence token by token in Python (`get_vectors`) and then multiplies three small similarity
matrices; here one kernel launch extracts the [32 rhythm | 12 melody | 12 harmony] features of all sequences and the
three similarity matrices come from the library's exact-fp32 MFMA GEMM."""
import torch

from ._lib import check, current_stream, lib, ptr, require_device


def get_vectors(tokens, lengths=None, note_len=128, return_status=False):
    """[B, L] int note tokens (from anywhere before the first BAR) -> fp32 [B, 56]; metric.py:4-71 for every row at once."""
    require_device(tokens, lengths)
    tokens = tokens.to(torch.int32).contiguous()
    B, L = tokens.shape
    lengths = None if lengths is None else lengths.to(torch.int32).contiguous()
    out = torch.empty(B, 56, device=tokens.device, dtype=torch.float32)
    status = torch.empty(B, device=tokens.device, dtype=torch.int32)
    check(lib().mh_msim_vectors(ptr(tokens), ptr(lengths), ptr(out), ptr(status), B, L, float(note_len), current_stream()), "mh_msim_vectors")
    return (out, status) if return_status else out


def _gram(a, b):
    """a b^T for [Ba, C] x [Bb, C] fp32 on the library's exact-fp32 MFMA GEMM (C zero-padded to the kernel's K granule of 16)."""
    from . import ops
    C = a.shape[1]
    Cp = (C + 15) // 16 * 16
    pad = lambda t: ops.cast_pad(t.contiguous(), Cp, ops.MH_F32)
    return ops.gemm_bias_act(pad(a), pad(b), None, None, None, ops.MH_F32, out_f32=True, N=b.shape[0], K=Cp)


def _similarity(vec_a, vec_b):
    """metric.py:96-104: rhythm x melody x harmony similarity matrices, multiplied element-wise"""
    r = _gram(vec_a[:, :32], vec_b[:, :32])
    m = _gram(vec_a[:, 32:44], vec_b[:, 32:44])
    h = _gram(vec_a[:, 44:], vec_b[:, 44:])
    return r * m * h


def MSIM(tokens1, tokens2, lengths1=None, lengths2=None):
    """metric.py:74-83 for paired batches: msim[b] = <r1,r2> <m1,m2> <h1,h2>"""
    a, b = get_vectors(tokens1, lengths1), get_vectors(tokens2, lengths2)
    return (a[:, :32] * b[:, :32]).sum(-1) * (a[:, 32:44] * b[:, 32:44]).sum(-1) * (a[:, 44:] * b[:, 44:]).sum(-1)


def ONNC(tokens, lengths=None, return_MSIM=False, return_mostsim=False):
    """metric.py:86-117: first half of the batch = ground truth, second half = generated; 1NNC from the MSIM matrix."""
    vec = get_vectors(tokens, lengths)
    sim = _similarity(vec, vec)
    sim.fill_diagonal_(0)
    most = torch.argmax(sim, dim=1)
    half = vec.shape[0] // 2
    onnc = ((most[:half] < half).sum() + (most[half:] >= half).sum()) / vec.shape[0]
    extra = ([sim] if return_MSIM else []) + ([most] if return_mostsim else [])
    return onnc if not extra else [onnc] + extra


PITCH_RANGE = {631: (3, 38), 632: (39, 50), 633: (51, 62), 634: (63, 74), 635: (75, 86), 636: (87, 98), 637: (99, 130)}


def _counts(metas, tokens, lengths):
    require_device(metas, tokens, lengths)
    tokens = tokens.to(torch.int32).contiguous()
    metas = metas.to(torch.int32).contiguous()
    B, L = tokens.shape
    lengths = None if lengths is None else lengths.to(torch.int32).contiguous()
    out = torch.empty(B, 4, device=tokens.device, dtype=torch.int32)
    check(lib().mh_controllability_counts(ptr(tokens), ptr(lengths), ptr(metas), metas.shape[1], ptr(out), B, L, current_stream()),
          "mh_controllability_counts")
    return out


def Controllability_Pitch(metas, tokens, lengths=None):
    """metric.py:131-148 on device batches -> (total rows, rows whose mean pitch leaves the range its meta asks for)"""
    c = _counts(metas, tokens, lengths)
    rng = metas[:, 3].long()
    lo = torch.tensor([PITCH_RANGE.get(k, (0, 0))[0] for k in range(631, 638)], device=tokens.device, dtype=torch.float32)
    hi = torch.tensor([PITCH_RANGE.get(k, (0, 0))[1] for k in range(631, 638)], device=tokens.device, dtype=torch.float32)
    idx = (rng - 631).clamp(0, 6)
    mean = c[:, 0].float() / c[:, 1].clamp(min=1).float()
    wrong = (rng != 630) & ~((lo[idx] <= mean) & (mean <= hi[idx]))
    return metas.shape[0], int(wrong.sum())


def Controllability_Velocity(metas, tokens, lengths=None):
    """metric.py:151-168 -> (velocity tokens of the rows with a bounded maximum, those outside their meta's [min, max])"""
    c = _counts(metas, tokens, lengths)
    use = (metas[:, 8].long() - 524) != 130
    return int(c[use, 2].sum()), int(c[use, 3].sum())
