"""ORACLE (test infrastructure only - never imported by the product): numpy restatement of the reference's batch
producers and token validators, the rows "next" to the hot path.

  corruptions        MuseDiffusion/data/corruption.py:100-195 (masking_token, masking_note, randomize_note, random_rotating)
  collate            MuseDiffusion/data/wrapper.py:90-127     (collate_batches)
  meta_to_batch      MuseDiffusion/utils/decode_util.py:221-230
  validators         MuseDiffusion/utils/decode_util.py:73-84 (remove_padding), :142-155 (validate_once), :157-183 (validate_rigidly)

The reference draws its randomness from a module-level random.Random in data-dependent order; here every draw is an
INPUT (arrays indexed the way the reference consumes them), which is what lets the device kernels be compared bit
for bit.  Pinned by tests/golden/batch.npz (tools/make_golden_batch.py runs the reference itself).
"""
import numpy as np

EOS, BAR = 1, 2
PITCH, VELOCITY, CHORD, DURATION, POSITION, BPM = 3, 131, 195, 304, 432, 560   # commu TOKEN_OFFSET (event_tokens.py:308-329)


def masking_token(seq, u, p):
    """corruption.py:100-114: tokens from index 12 up to (not including) the first EOS at or after 12 become 0 when the
    k-th draw u[k] < p, k counting positions from 12."""
    out = seq.copy()
    for i in range(len(seq) - 12):
        if seq[i + 12] == EOS:
            break
        if u[i] < p:
            out[i + 12] = 0
    return out


def _eligible_velocity(seq):
    """velocity tokens (131..194) in index order, skipping those with idx + 3 > len(seq) (corruption.py:125-131)"""
    idx = np.nonzero((seq >= 131) & (seq <= 194))[0]
    return [int(i) for i in idx if i + 3 <= len(seq)]


def masking_note(seq, u, p):
    """corruption.py:117-133: the k-th eligible velocity token zeroes seq[idx-1 : idx+3] when u[k] < p"""
    out = seq.copy()
    for k, idx in enumerate(_eligible_velocity(seq)):
        if u[k] < p:
            out[idx - 1: idx + 3] = 0
    return out


def randomize_note(seq, u, new, p):
    """corruption.py:136-162: the k-th eligible velocity token gets new[k] = (velocity, pitch, duration) when u[k] < p"""
    out = seq.copy()
    for k, idx in enumerate(_eligible_velocity(seq)):
        if u[k] < p:
            out[idx], out[idx + 1], out[idx + 2] = new[k]
    return out


def random_rotating(seq, pairs):
    """corruption.py:165-195: for every (first, second) pair swap two bars.  Bar starts and the EOS position are taken ONCE
    from the input sequence and reused for every swap (the reference does not refresh them after a rotation)."""
    rot = seq.copy()
    bar_idx = np.nonzero(seq == BAR)[0]
    eos_idx = np.nonzero(seq == EOS)[0][-1]
    for first, second in pairs:
        b1s, b2s = bar_idx[first], bar_idx[second]
        b1e = bar_idx[first + 1]
        b2e = bar_idx[second + 1] if second < len(bar_idx) - 1 else eos_idx
        rot = np.concatenate([rot[:b1s], rot[b2s:b2e], rot[b1e:b2s], rot[b1s:b1e], rot[b2e:]])
    return rot


def collate(rows, masks, correct, seq_len=None):
    """wrapper.py:90-127 -> dict of [B, seq_len] int arrays: ids / correct_ids zero padded, mask padded with ONES, length"""
    L = seq_len or max(len(r) for r in rows)
    B = len(rows)
    ids, cor, msk = np.zeros((B, L), np.int32), np.zeros((B, L), np.int32), np.ones((B, L), np.int32)
    for b in range(B):
        n = len(rows[b])
        ids[b, :n], cor[b, :n], msk[b, :n] = rows[b], correct[b], masks[b]
    return {"input_ids": ids, "correct_ids": cor, "input_mask": msk, "length": np.array([len(r) for r in rows], np.int32)}


def meta_to_batch(meta, B, L):
    """decode_util.py:221-230: ids[:, :len] = meta, mask = 1 except [:, :len + 1] = 0"""
    ids, msk = np.zeros((B, L), np.int32), np.ones((B, L), np.int32)
    ids[:, :len(meta)] = meta
    msk[:, :len(meta) + 1] = 0
    return ids, msk


def validate(tokens, length):
    """(eos index or -1, validate_once ok, validate_rigidly ok) of one note sequence, as the reference's
    remove_padding -> validate_once / validate_rigidly chain decides them.  validate_rigidly's index error on a note
    truncated at the very end is reported as -2 (the reference raises IndexError there, decode_util.py:171-175)."""
    seq = tokens[:length]
    eos = np.nonzero(seq == EOS)[0]
    if len(eos) == 0:
        return -1, 0, 0
    seq = seq[: eos[0] + 1]
    n = len(seq)
    once = 0
    for i in range(n):
        if i + 2 > n - 1:
            break
        if (VELOCITY <= seq[i] < CHORD and POSITION <= seq[i - 1] < BPM and PITCH <= seq[i + 1] < VELOCITY
                and DURATION <= seq[i + 2] < POSITION):
            once = 1
            break
    rigid, i = 0, 0
    while True:
        if i >= n:
            break
        if seq[i] == EOS:
            rigid = 1
            break
        if seq[i] == BAR:
            i += 1
            continue
        if not (POSITION <= seq[i] < BPM):
            break
        if i + 1 >= n:
            rigid = -2
            break
        if VELOCITY <= seq[i + 1] < CHORD:
            if i + 3 >= n:
                rigid = -2
                break
            if PITCH <= seq[i + 2] < VELOCITY and DURATION <= seq[i + 3] < POSITION:
                i += 4
                continue
            break
        if CHORD <= seq[i + 1] < DURATION:
            i += 2
            continue
        break
    return int(eos[0]), once, rigid


def msim_vectors(midi, note_len=128):
    """metric.py:4-71 get_vectors: [32 rhythm | 12 melody | 12 harmony] fp32, each L2-normalised.  `midi` = note tokens from
    anywhere before the first BAR up to the EOS / padding.  The amplitude expression is evaluated in double and rounded to
    fp32 on store, as Python floats assigned into a float32 tensor are."""
    f32 = np.float32
    i = 0
    while midi[i] != BAR:
        i += 1
    i += 1
    rhythm = np.full(32, 1e-8, f32)
    tmp = np.full(32, 1e-8, f32)
    melody = np.full(12, 1e-8, f32)
    harmony = np.zeros(12, f32)
    cur_hi, prev_hi, prev_startp, startp = -1, -1, -1, None

    def norm(v):
        return f32(np.sqrt(np.sum(v.astype(np.float32) ** 2, dtype=np.float32)))
    while True:
        if midi[i] <= 2:
            tmp = tmp / norm(tmp)
            rhythm = rhythm + tmp
            tmp = np.full(32, 1e-8, f32)
            i += 1
            if midi[i - 1] == BAR:
                prev_startp = -1
                continue
            if prev_startp != startp and prev_hi >= 0:
                melody[(cur_hi - prev_hi) % 12] += 1
            break
        assert POSITION <= midi[i] <= 559
        startp = int(midi[i]) - POSITION
        if CHORD <= midi[i + 1] <= 303:
            i += 2
            continue
        assert VELOCITY <= midi[i + 1] <= 194 and PITCH <= midi[i + 2] <= 130 and DURATION <= midi[i + 3] <= 431
        pitch = int(midi[i + 2])
        endp = startp + int(midi[i + 3]) - 303
        harmony[pitch % 12] += 1
        amp = (0.00542676376 * (int(midi[i + 1]) - 130) * 2 + 0.310801) ** 2
        for t in range(0, min(128, endp), 4):
            if t < startp:
                continue
            v = amp * max(0, 1 - (t - startp) / note_len)
            if v > tmp[t // 4]:
                tmp[t // 4] = f32(v)
        if cur_hi >= 0 and prev_startp != startp:
            if prev_hi >= 0:
                melody[(cur_hi - prev_hi) % 12] += 1
            prev_hi = cur_hi
            cur_hi = pitch
        cur_hi = max(pitch, cur_hi)
        prev_startp = startp
        i += 4
    return np.concatenate([rhythm / norm(rhythm), melody / norm(melody), harmony / norm(harmony)]).astype(f32)


def onnc(vectors):
    """metric.py:86-117: MSIM matrix (diagonal zeroed), most similar index per row, 1NNC over (first half GT, second generated)"""
    r, m, h = vectors[:, :32], vectors[:, 32:44], vectors[:, 44:]
    sim = (r @ r.T) * (m @ m.T) * (h @ h.T)
    np.fill_diagonal(sim, 0)
    most = sim.argmax(1)
    half = len(vectors) // 2
    return ((most[:half] < half).sum() + (most[half:] >= half).sum()) / len(vectors), sim, most


PITCH_RANGE = {631: (3, 38), 632: (39, 50), 633: (51, 62), 634: (63, 74), 635: (75, 86), 636: (87, 98), 637: (99, 130)}


def controllability(metas, midis):
    """metric.py:120-168: ((total, wrong) of Controllability_Pitch, (total, wrong) of Controllability_Velocity).  Pitch: a row
    whose meta[3] != 630 is wrong when the mean of its pitch tokens (3..130) leaves the meta's range.  Velocity: rows with
    meta[8] - 524 != 130 contribute their velocity tokens (131..194); one is wrong outside [meta[7] - 524, meta[8] - 524]
    (bound 130 / 195 = open)."""
    p_wrong, v_total, v_wrong = 0, 0, 0
    for meta, midi in zip(metas, midis):
        midi = np.asarray(midi)
        if meta[3] != 630:
            pitch = midi[(midi >= 3) & (midi <= 130)]
            lo, hi = PITCH_RANGE[int(meta[3])]
            if not (lo <= float(pitch.mean()) <= hi):
                p_wrong += 1
        lo_v, hi_v = int(meta[7]) - 524, int(meta[8]) - 524
        if hi_v != 130:
            vel = midi[(midi >= 131) & (midi <= 194)]
            v_total += len(vel)
            v_wrong += int(sum(not ((lo_v == 130 or lo_v <= e) and (hi_v == 195 or e <= hi_v)) for e in vel))
    return (len(metas), p_wrong), (v_total, v_wrong)
