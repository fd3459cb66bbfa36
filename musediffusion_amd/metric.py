"""MSIM / 1NNC of MuseDiffusion/metric.py:4-117 for whole batches on the device.

The reference walks every sequence token by token in Python (`get_vectors`) and then multiplies three small similarity
matrices; here one kernel launch extracts the [32 rhythm | 12 melody | 12 harmony] features of all sequences and the
matrix part stays the reference's three matmuls (on the device)."""
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


def _similarity(vec_a, vec_b):
    r = vec_a[:, :32] @ vec_b[:, :32].T
    m = vec_a[:, 32:44] @ vec_b[:, 32:44].T
    h = vec_a[:, 44:] @ vec_b[:, 44:].T
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
