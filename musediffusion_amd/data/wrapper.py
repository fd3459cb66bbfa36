"""collate_batches of MuseDiffusion/data/wrapper.py:90-127 on the device: ragged fields -> padded [B, seq_len] tensors."""
import torch

from .._lib import check, current_stream, lib, ptr, require_device


def to_ragged(rows, device):
    """list of 1-D int sequences -> (values int32 [sum len], offsets int64 [B + 1]) on `device` (one H2D copy each)"""
    lens = torch.tensor([len(r) for r in rows], dtype=torch.int64)
    offsets = torch.zeros(len(rows) + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(lens, 0)
    values = torch.cat([torch.as_tensor(r, dtype=torch.int32).reshape(-1) for r in rows]) if rows else torch.zeros(0, dtype=torch.int32)
    return values.to(device), offsets.to(device)


def _pad(values, offsets, L, pad, want_length=False):
    require_device(values, offsets)
    B = offsets.numel() - 1
    out = torch.empty(B, L, device=values.device, dtype=torch.int32)
    length = torch.empty(B, device=values.device, dtype=torch.int32) if want_length else None
    check(lib().mh_ragged_to_padded(ptr(values.to(torch.int32).contiguous()), ptr(offsets.to(torch.int64).contiguous()), ptr(out),
                                    ptr(length), B, L, int(pad), current_stream()), "mh_ragged_to_padded")
    return (out, length) if want_length else out


def collate_batches(fields, offsets, seq_len=None):
    """fields: dict name -> ragged int32 values sharing `offsets` (keys as in the reference: 'input_ids', 'input_mask',
    optional 'correct_ids', 'label').  Returns the reference's dict: ids / correct_ids / label zero padded, input_mask
    padded with ONES, plus 'length' (wrapper.py:100-127).  seq_len None = the longest row (one host sync to read it)."""
    L = int(seq_len) if seq_len else int((offsets[1:] - offsets[:-1]).max())
    out = {}
    for name, vals in fields.items():
        if name == "input_ids":
            out[name], out["length"] = _pad(vals, offsets, L, 0, want_length=True)
        else:
            out[name] = _pad(vals, offsets, L, 1 if name == "input_mask" else 0)
    return out
