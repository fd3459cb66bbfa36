#!/usr/bin/env python3
"""Generate tests/golden/batch.npz: outputs of the REFERENCE's batch producers and token validators on seeded inputs.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_batch.py

Imports (read-only, from /root/reference) MuseDiffusion.data.corruption, MuseDiffusion.data.wrapper.collate_batches and
MuseDiffusion.utils.decode_util (meta_to_batch's layout, SequenceToMidi.remove_padding / validate_once /
validate_rigidly).  decode_util imports the vendored `commu` package, which needs `logger`, `miditoolkit`, `parmap`
and `pretty_midi` (absent here, unused by the functions exercised): the harness registers empty stand-in modules
for those names in sys.modules before the import - nothing under /root/reference is touched.

The corruptions draw from a module-level random.Random in data-dependent order.  The generator is wrapped by a
recorder so the fixture holds, per call, the exact draws: the device kernels and the numpy oracle take them as inputs
("injected randomness", the same device the diffusion parity tests use for the Gaussian noise).
"""
import os
import random
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MUSE_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, REPO)


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        m = _Stub(self.__name__ + "." + k)
        setattr(self, k, m)
        return m

    def __call__(self, *a, **k):
        return self


for _name in ("logger", "miditoolkit", "parmap", "pretty_midi"):
    sys.modules.setdefault(_name, _Stub(_name))

from MuseDiffusion.data import corruption as rcorr  # noqa: E402
from MuseDiffusion.data.wrapper import collate_batches  # noqa: E402
from MuseDiffusion.utils import decode_util as rdec  # noqa: E402
from MuseDiffusion import metric as rmetric  # noqa: E402


class Recorder:
    """random.Random look-alike that logs every draw the corruption functions make."""

    def __init__(self, seed):
        self.r = random.Random(seed)
        self.log = []

    def random(self):
        v = self.r.random()
        self.log.append(("u", v))
        return v

    def randint(self, a, b):
        v = self.r.randint(a, b)
        self.log.append(("i", v))
        return v

    def sample(self, pop, k):
        v = self.r.sample(pop, k)
        self.log.append(("s", list(v)))
        return v

    def shuffle(self, x):
        self.r.shuffle(x)


def make_sequence(g, n_bars, notes_per_bar, drop_tail=False):
    """ComMU-shaped training sequence: 11 meta tokens, one 0 separator, bars of (position, velocity, pitch, duration), EOS."""
    meta = [g.randint(560, 600), g.randint(601, 625), g.randint(626, 629), g.randint(630, 637), g.randint(638, 640),
            g.randint(641, 649), g.randint(650, 652), g.randint(653, 717), g.randint(653, 718), g.randint(719, 725),
            g.randint(726, 728)]
    seq = meta + [0]
    for _ in range(n_bars):
        seq.append(2)
        pos = sorted(g.sample(range(432, 560), notes_per_bar))
        for p in pos:
            if g.random() < 0.15:
                seq += [p, g.randint(195, 303)]                   # chord event
            else:
                seq += [p, g.randint(131, 194), g.randint(3, 130), g.randint(304, 431)]
    seq.append(1)
    if drop_tail:                                                  # a velocity token right at the end (idx + 3 > len case)
        seq += [433, 150]
    return seq


def main():
    g = random.Random(2024)
    seqs = [make_sequence(g, g.randint(3, 8), g.randint(2, 6), drop_tail=(i % 5 == 4)) for i in range(12)]
    out = {}
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    out["values"] = np.concatenate([np.array(s, dtype=np.int32) for s in seqs])
    out["offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)

    # ---- the four corruptions, each on every sequence, draws recorded (data/corruption.py:100-195)
    for key, fn, kw in (("mt", rcorr.masking_token, dict(p=0.3)), ("mn", rcorr.masking_note, dict(p=0.5)),
                        ("rn", rcorr.randomize_note, dict(p=0.5)), ("rr", rcorr.random_rotating, dict(count=3))):
        res, logs = [], []
        for i, s in enumerate(seqs):
            rec = Recorder(100 + i)
            rcorr.generator = rec
            res.append(fn(torch.tensor(s, dtype=torch.long), **kw).numpy().astype(np.int32))
            logs.append(rec.log)
        out[key + "_out"] = np.concatenate(res)
        assert all(len(r) == len(s) for r, s in zip(res, seqs))
        if key in ("mt", "mn"):
            out[key + "_u"] = np.concatenate([np.array([v for _, v in lg] + [2.0] * (len(s) - len(lg)), dtype=np.float64)
                                              for lg, s in zip(logs, seqs)])      # padded to the row length with 2.0 (never < p)
        elif key == "rn":
            u, iv = [], []
            for lg, s in zip(logs, seqs):
                uu, ii = [], []
                k = 0
                while k < len(lg):
                    assert lg[k][0] == "u"
                    uu.append(lg[k][1])
                    if lg[k][1] < 0.5:
                        ii.append([lg[k + 1][1], lg[k + 2][1], lg[k + 3][1]])
                        k += 4
                    else:
                        ii.append([0, 0, 0])
                        k += 1
                uu += [2.0] * (len(s) - len(uu))
                ii += [[0, 0, 0]] * (len(s) - len(ii))
                u.append(np.array(uu, dtype=np.float64))
                iv.append(np.array(ii, dtype=np.int32))
            out["rn_u"], out["rn_new"] = np.concatenate(u), np.concatenate(iv)
        else:
            out["rr_pairs"] = np.array([[sorted(v) for _, v in lg] for lg in logs], dtype=np.int32)   # [rows, count, 2]

    # ---- collate_batches on a corrupted batch (data/wrapper.py:90-127)
    samples = []
    for i, s in enumerate(seqs[:6]):
        ids = torch.tensor(s, dtype=torch.long)
        mask = torch.ones(len(s), dtype=torch.long)
        mask[:12] = 0
        samples.append({"input_ids": torch.from_numpy(out["mt_out"][out["offsets"][i]:out["offsets"][i + 1]].astype(np.int64)),
                        "correct_ids": ids, "input_mask": mask, "length": len(s)})
    for L, tag in ((None, "max"), (256, "256")):
        col = collate_batches(samples, seq_len=L)
        for k, v in col.items():
            out["collate_%s_%s" % (tag, k)] = v.numpy().astype(np.int32)

    # ---- meta_to_batch layout (utils/decode_util.py:221-230) with an already-encoded meta prefix
    meta = seqs[0][:11] + [432, 200, 464, 210, 432, 220]
    ids = torch.zeros(5, 64, dtype=torch.int)
    enc = torch.tensor(meta)
    ids[:, :len(enc)] = enc
    msk = torch.ones(5, 64, dtype=torch.int)
    msk[:, :len(enc) + 1] = 0
    out["m2b_meta"], out["m2b_ids"], out["m2b_mask"] = np.array(meta, dtype=np.int32), ids.numpy(), msk.numpy()

    # ---- token validation (utils/decode_util.py:73-190): first-EOS cut, validate_once, validate_rigidly
    S = rdec.SequenceToMidi
    cases = [np.array(s[12:], dtype=np.int64) for s in seqs]
    gg = random.Random(7)
    for s in list(cases[:6]):
        t = s.copy()
        t[gg.randrange(len(t))] = gg.randint(0, 728)                # a random token somewhere
        cases.append(t)
    cases.append(np.array([2, 432, 140, 60, 310, 1, 0, 0], dtype=np.int64))
    cases.append(np.array([2, 2, 440, 200, 1], dtype=np.int64))      # chords only: rigid ok, once fails
    cases.append(np.array([2, 432, 140, 60, 310, 0, 0], dtype=np.int64))   # no EOS
    cases.append(np.array([432, 140, 60], dtype=np.int64))
    Lv = max(len(c) for c in cases)
    toks = np.zeros((len(cases), Lv), dtype=np.int32)
    res = np.zeros((len(cases), 3), dtype=np.int32)                  # [eos index or -1, validate_once ok, validate_rigidly ok]
    lens_v = np.zeros(len(cases), dtype=np.int32)
    for i, c in enumerate(cases):
        toks[i, :len(c)] = c
        lens_v[i] = len(c)
        try:
            cut = S.remove_padding(c)
            res[i, 0] = len(cut) - 1
        except rdec.SequenceToMidiError:
            res[i, 0] = -1
            cut = None
        if cut is not None:
            for j, f in ((1, S.validate_once), (2, S.validate_rigidly)):
                try:
                    f(cut)
                    res[i, j] = 1
                except rdec.SequenceToMidiError:
                    res[i, j] = 0
                except IndexError:                                   # validate_rigidly indexes past the end on truncated notes
                    res[i, j] = -2
    out["val_tokens"], out["val_len"], out["val_result"] = toks, lens_v, res

    # ---- MSIM feature vectors / MSIM / 1NNC (metric.py:4-117) on the note parts of the 12 sequences
    notes = [np.array(s[12:], dtype=np.int64) for s in seqs]
    vecs = [torch.cat(rmetric.get_vectors(n)).numpy() for n in notes]              # [32 rhythm | 12 melody | 12 harmony]
    out["msim_vectors"] = np.stack(vecs).astype(np.float32)
    out["msim_01"] = np.float32(rmetric.MSIM(notes[0], notes[1]))
    onnc, sim, most = rmetric.ONNC(notes, return_MSIM=True, return_mostsim=True)
    out["onnc"], out["onnc_msim"], out["onnc_mostsim"] = np.float32(onnc), sim.numpy().astype(np.float32), most.numpy().astype(np.int64)

    # ---- controllability counters (metric.py:120-168) on (meta, notes) pairs; metas 3 / 7 / 8 edited to hit every branch
    metas = np.array([s[:11] for s in seqs], dtype=np.int64)
    metas[0, 3] = 630                                   # pitch range "any": skipped
    metas[1, 3], metas[2, 3] = 631, 637
    metas[3, 8] = 130 + 524                             # max velocity "any": skipped
    metas[4, 7], metas[4, 8] = 130 + 524, 160 + 524     # no lower bound
    metas[5, 7], metas[5, 8] = 140 + 524, 195 + 524     # no upper bound
    for i in range(6, 12):
        metas[i, 7], metas[i, 8] = 135 + 524 + i, 150 + 524 + 2 * i
    for i in range(3, 12):
        metas[i, 3] = 631 + (i % 7)
    note_arrays = [np.array(s[12:], dtype=np.int64) for s in seqs]
    out["ctrl_metas"] = metas.astype(np.int32)
    out["ctrl_pitch"] = np.array(rmetric.Controllability_Pitch(metas, note_arrays), dtype=np.int64)
    out["ctrl_velocity"] = np.array(rmetric.Controllability_Velocity(metas, note_arrays), dtype=np.int64)

    path = os.path.join(REPO, "tests", "golden", "batch.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
