"""Device versions of the batch layout / token checks of MuseDiffusion/utils/decode_util.py.
The MIDI encoders / decoders themselves (the vendored `commu` package) are out of scope: `meta_to_batch` takes the
already encoded meta + chord tokens."""
import torch

from .._lib import check, current_stream, lib, ptr, require_device


def meta_to_batch(encoded_meta, batch_size, seq_len, device="cuda"):
    """decode_util.py:221-230: input_ids[:, :len(meta)] = meta, input_mask = 1 except [:, :len(meta) + 1] = 0 (int32)"""
    meta = torch.as_tensor(encoded_meta, dtype=torch.int32).to(device).contiguous()
    ids = torch.empty(batch_size, seq_len, device=device, dtype=torch.int32)
    mask = torch.empty_like(ids)
    check(lib().mh_meta_to_batch(ptr(meta), meta.numel(), ptr(ids), ptr(mask), batch_size, seq_len, current_stream()), "mh_meta_to_batch")
    return {"input_ids": ids, "input_mask": mask}


def validate_tokens(tokens, lengths=None):
    """[B, L] int tokens (note sequences, meta already split off) -> int32 [B, 3]: (index of the first EOS or -1 = the
    reference's "NO EOS TOKEN", validate_once passes, validate_rigidly passes; -2 where the reference's strict validator
    indexes past the end of a truncated note) - decode_util.py:73-84, :142-183, without leaving the device."""
    require_device(tokens, lengths)
    tokens = tokens.to(torch.int32).contiguous()
    B, L = tokens.shape
    lengths = None if lengths is None else lengths.to(torch.int32).contiguous()
    res = torch.empty(B, 3, device=tokens.device, dtype=torch.int32)
    check(lib().mh_validate_tokens(ptr(tokens), ptr(lengths), ptr(res), B, L, current_stream()), "mh_validate_tokens")
    return res
