"""Synthetic ComMU-shaped token batches (no dataset, no network).

The reference feeds the hot path with `{'input_ids', 'input_mask'[, 'correct_ids', 'length']}`
batches (MuseDiffusion/data/wrapper.py:90-127, utils/decode_util.py:221-230).  These
generators reproduce the *layout and value ranges* of those batches from the vocabulary
offsets in commu/preprocessor/encoder/event_tokens.py:308-329 (vocab 729):

  meta (11 tokens): BPM 560-600, KEY 601-625, TS 626-629, PITCH_RANGE 630-637,
      NUM_MEASURES 638-640, INST 641-649, GENRE 650-652, MIN_VEL/MAX_VEL 653-718,
      TRACK_ROLE 719-725, RHYTHM 726-728
  chords: per bar `432, chord in [195,303]` (+ optional `432+16i, chord`)   (decode_util.py:25-39)
  notes:  BAR=2 then (POSITION 432-559, VELOCITY 131-194, PITCH 3-130, DURATION 304-431)*
  EOS=1, padding 0.  input_mask is 0 over meta+chords+EOS and 1 elsewhere (incl. padding)
  (data/preprocess.py:54-55).

CPU int tensors only; deterministic under the given seed.
"""
import torch

VOCAB_SIZE = 729
_META_RANGES = ((560, 600), (601, 625), (626, 629), (630, 637), (638, 640), (641, 649),
                (650, 652), (653, 717), (653, 718), (719, 725), (726, 728))


def _randint(g, lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


def _meta_prefix(g, bars=8):
    toks = [_randint(g, lo, hi) for lo, hi in _META_RANGES]
    for _ in range(bars):
        toks += [432, _randint(g, 195, 303)]
        if _randint(g, 0, 3) == 0:  # a chord change inside the bar
            toks += [432 + 16 * _randint(g, 1, 7), _randint(g, 195, 303)]
    return toks


def generation_batch(batch_size, seq_len, seed=1):
    """`meta_to_batch` layout (decode_util.py:221-230): one meta prefix repeated over the batch,
    zeros after it, mask 0 over prefix + 1 slot.  dtype int32 like the reference."""
    g = torch.Generator().manual_seed(seed)
    meta = _meta_prefix(g)[: max(1, seq_len - 2)]
    ids = torch.zeros(batch_size, seq_len, dtype=torch.int)
    ids[:, : len(meta)] = torch.tensor(meta, dtype=torch.int)
    mask = torch.ones(batch_size, seq_len, dtype=torch.int)
    mask[:, : len(meta) + 1] = 0
    return {"input_ids": ids, "input_mask": mask}


def training_batch(batch_size, seq_len, seed=1, corruption_p=0.3, fill=0.8):
    """Training / modification layout (data/preprocess.py:54-55, data/wrapper.py:100-127):
    [meta+chords, EOS, notes..., 0 padding]; `correct_ids` is the clean sequence and
    `input_ids` has ~corruption_p of the note tokens zeroed (the 'mt' corruption,
    data/corruption.py:100-114).  dtype int64 like the reference's collate."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(batch_size, seq_len, dtype=torch.long)
    mask = torch.ones(batch_size, seq_len, dtype=torch.long)
    length = torch.zeros(batch_size, dtype=torch.long)
    for b in range(batch_size):
        seq = _meta_prefix(g)[: max(1, seq_len // 2)]
        src_len = len(seq) + 1
        seq.append(1)  # EOS separates condition from target
        budget = int((seq_len - len(seq)) * fill * (0.5 + 0.5 * float(torch.rand(1, generator=g))))
        notes = []
        while len(notes) + 5 <= budget:
            if len(notes) % 33 == 0:
                notes.append(2)  # BAR
            notes += [_randint(g, 432, 559), _randint(g, 131, 194), _randint(g, 3, 130), _randint(g, 304, 431)]
        seq += notes
        seq = seq[:seq_len]
        ids[b, : len(seq)] = torch.tensor(seq)
        mask[b, :src_len] = 0
        length[b] = len(seq)
    correct = ids.clone()
    drop = (torch.rand(ids.shape, generator=g) < corruption_p) & (mask == 1) & (ids != 0)
    corrupted = torch.where(drop, torch.zeros_like(ids), ids)
    return {"input_ids": corrupted, "input_mask": mask, "correct_ids": correct, "length": length}
