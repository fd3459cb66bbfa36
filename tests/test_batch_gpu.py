"""GPU: device batch producers / token validators through the C ABI against the reference's recorded outputs
(tests/golden/batch.npz) with the reference's own random draws injected.  Integer work: bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402
from musediffusion_amd import data as mdata  # noqa: E402
from musediffusion_amd.utils import decode_util as mdec  # noqa: E402
from oracle import batch as ob  # noqa: E402

DEV = "cuda"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def test_corruptions_bit_exact_with_injected_draws():
    g = load_golden("batch.npz")
    v, off = dev(g["values"]), dev(g["offsets"])
    assert torch.equal(mdata.masking_token(v, off, 0.3, u=dev(g["mt_u"], torch.float32)).cpu(), torch.from_numpy(g["mt_out"]))
    assert torch.equal(mdata.masking_note(v, off, 0.5, u=dev(g["mn_u"], torch.float32)).cpu(), torch.from_numpy(g["mn_out"]))
    assert torch.equal(mdata.randomize_note(v, off, 0.5, u=dev(g["rn_u"], torch.float32), new_tokens=dev(g["rn_new"])).cpu(),
                       torch.from_numpy(g["rn_out"]))
    out, status = mdata.random_rotating(v, off, 3, pairs=dev(g["rr_pairs"]), return_status=True)
    assert torch.equal(out.cpu(), torch.from_numpy(g["rr_out"])) and int(status.abs().sum()) == 0


def test_corruptions_with_device_draws_keep_the_invariants():
    """no injected randomness: outputs differ from the input only where the reference's rules allow it to"""
    g = load_golden("batch.npz")
    v, off = dev(g["values"]), dev(g["offsets"])
    torch.manual_seed(3)
    rows = lambda t: [t[g["offsets"][i]:g["offsets"][i + 1]] for i in range(len(g["offsets"]) - 1)]
    mt = mdata.masking_token(v, off, 0.5).cpu().numpy()
    for a, b in zip(rows(g["values"]), rows(mt)):
        eos = 12 + int(np.nonzero(a[12:] == 1)[0][0])
        ch = a != b
        assert not ch[:12].any() and not ch[eos:].any() and (b[ch] == 0).all() and ch.any()
    rn = mdata.randomize_note(v, off, 1.0).cpu().numpy()
    for a, b in zip(rows(g["values"]), rows(rn)):
        for idx in ob._eligible_velocity(a):
            assert 131 <= b[idx] <= 194
        assert np.array_equal(a == 2, b == 2) and np.array_equal(a[:12], b[:12])
    rr, st = mdata.random_rotating(v, off, 3, return_status=True)
    rr = rr.cpu().numpy()
    assert int(st.sum()) == 0
    for a, b in zip(rows(g["values"]), rows(rr)):
        assert np.array_equal(np.sort(a), np.sort(b)) and np.array_equal(a[:12], b[:12])        # a permutation of the note region
    c = mdata.Corruptions.from_config("mt,mn,rn,rr", 2, 1.0)
    assert c(v, off).shape == v.shape and "corr_max=2" in repr(c)


def test_collate_meta_to_batch_and_validators():
    g = load_golden("batch.npz")
    off6 = g["offsets"][:7]
    n6 = int(off6[-1])
    masks = np.ones(n6, np.int32)
    for i in range(6):
        masks[off6[i]:off6[i] + 12] = 0
    fields = {"input_ids": dev(g["mt_out"][:n6]), "correct_ids": dev(g["values"][:n6]), "input_mask": dev(masks)}
    for L, tag in ((None, "max"), (256, "256")):
        col = mdata.collate_batches(fields, dev(off6), L)
        for k in ("input_ids", "correct_ids", "input_mask", "length"):
            assert torch.equal(col[k].cpu(), torch.from_numpy(g["collate_%s_%s" % (tag, k)])), (tag, k)
    v2, o2 = mdata.to_ragged([list(g["values"][off6[i]:off6[i + 1]]) for i in range(6)], DEV)
    assert torch.equal(v2.cpu(), torch.from_numpy(g["values"][:n6])) and torch.equal(o2.cpu(), torch.from_numpy(off6))
    b = mdec.meta_to_batch(g["m2b_meta"], 5, 64, DEV)
    assert torch.equal(b["input_ids"].cpu(), torch.from_numpy(g["m2b_ids"])) and torch.equal(b["input_mask"].cpu(), torch.from_numpy(g["m2b_mask"]))
    res = mdec.validate_tokens(dev(g["val_tokens"]), dev(g["val_len"]))
    assert torch.equal(res.cpu(), torch.from_numpy(g["val_result"]))


def test_validators_and_corruptions_at_batch_scale_match_the_oracle():
    """BASELINE-sized batch (512 sequences of up to 1024 tokens) of random ComMU-shaped rows: kernels == numpy oracle"""
    rng = np.random.default_rng(5)
    rows = []
    for _ in range(512):
        n_notes = int(rng.integers(5, 200))
        seq = list(rng.integers(560, 729, 11)) + [0]
        for k in range(n_notes):
            if k % 3 == 0:
                seq.append(2)
            seq += [int(rng.integers(432, 560)), int(rng.integers(131, 195)), int(rng.integers(3, 131)), int(rng.integers(304, 432))]
        seq.append(1)
        rows.append(np.array(seq, np.int32))
    v, off = mdata.to_ragged(rows, DEV)
    n = v.numel()
    u = torch.from_numpy(rng.random(n).astype(np.float32))
    new = torch.from_numpy(np.stack([rng.integers(131, 195, n), rng.integers(3, 131, n), rng.integers(304, 432, n)], 1).astype(np.int32))
    offs = off.cpu().numpy()
    got_mt = mdata.masking_token(v, off, 0.3, u=u.to(DEV)).cpu().numpy()
    got_mn = mdata.masking_note(v, off, 0.5, u=u.to(DEV)).cpu().numpy()
    got_rn = mdata.randomize_note(v, off, 0.5, u=u.to(DEV), new_tokens=new.to(DEV)).cpu().numpy()
    pairs = np.zeros((len(rows), 3, 2), np.int32)
    for i, r in enumerate(rows):
        nb = int((r == 2).sum())
        for s in range(3):
            pairs[i, s] = sorted(rng.choice(nb, 2, replace=False))
    got_rr = mdata.random_rotating(v, off, 3, pairs=torch.from_numpy(pairs).to(DEV)).cpu().numpy()
    un, nn = u.numpy(), new.numpy()
    for i in range(0, len(rows), 7):
        a, b = offs[i], offs[i + 1]
        assert np.array_equal(got_mt[a:b], ob.masking_token(rows[i], un[a:b], np.float32(0.3))), i
        assert np.array_equal(got_mn[a:b], ob.masking_note(rows[i], un[a:b], np.float32(0.5))), i
        assert np.array_equal(got_rn[a:b], ob.randomize_note(rows[i], un[a:b], nn[a:b], np.float32(0.5))), i
        assert np.array_equal(got_rr[a:b], ob.random_rotating(rows[i], pairs[i])), i
    L = 1024
    col = mdata.collate_batches({"input_ids": torch.from_numpy(got_mn).to(DEV)}, off, L)
    notes = col["input_ids"][:, 12:].contiguous()
    res = mdec.validate_tokens(notes, (col["length"] - 12).clamp(min=0)).cpu().numpy()
    toks = notes.cpu().numpy()
    lens = (col["length"].cpu().numpy() - 12).clip(0)
    for i in range(0, len(rows), 5):
        assert tuple(res[i]) == ob.validate(toks[i], int(lens[i])), i


def test_msim_vectors_and_onnc_match_reference():
    from musediffusion_amd import metric as mm
    g = load_golden("batch.npz")
    off = g["offsets"]
    L = 256
    toks = np.zeros((12, L), np.int32)
    lens = np.zeros(12, np.int32)
    for i in range(12):
        s = g["values"][off[i] + 12:off[i + 1]]
        toks[i, :len(s)], lens[i] = s, len(s)
    vec, st = mm.get_vectors(dev(toks), dev(lens), return_status=True)
    assert int(st.abs().sum()) == 0
    np.testing.assert_allclose(vec.cpu().numpy(), g["msim_vectors"], rtol=2e-6, atol=2e-7)       # fp32 norm summation order
    onnc, sim, most = mm.ONNC(dev(toks), dev(lens), return_MSIM=True, return_mostsim=True)
    assert np.array_equal(most.cpu().numpy(), g["onnc_mostsim"]) and abs(float(onnc) - float(g["onnc"])) < 1e-7
    np.testing.assert_allclose(sim.cpu().numpy(), g["onnc_msim"], rtol=1e-5, atol=1e-7)
    ms = mm.MSIM(dev(toks[:1]), dev(toks[1:2]), dev(lens[:1]), dev(lens[1:2]))
    assert abs(float(ms[0]) - float(g["msim_01"])) < 1e-6
    # malformed rows are flagged, not crashed on: no BAR / a position token followed by garbage
    bad = np.zeros((2, 16), np.int32)
    bad[0, :4] = [432, 140, 60, 310]
    bad[1, :5] = [2, 432, 7, 7, 1]
    _, st2 = mm.get_vectors(dev(bad), return_status=True)
    assert st2.cpu().tolist() == [1, 2]
    # batch scale against the oracle
    rng = np.random.default_rng(9)
    rows = []
    for _ in range(256):
        seq = []
        for k in range(int(rng.integers(5, 180))):
            if k % 5 == 0:
                seq.append(2)
            seq += [int(rng.integers(432, 560)), int(rng.integers(131, 195)), int(rng.integers(3, 131)), int(rng.integers(304, 432))]
        seq.append(1)
        rows.append(np.array(seq, np.int32))
    Lb = max(len(r) for r in rows)
    tb = np.zeros((len(rows), Lb), np.int32)
    for i, r in enumerate(rows):
        tb[i, :len(r)] = r
    got = mm.get_vectors(dev(tb)).cpu().numpy()
    for i in range(0, len(rows), 9):
        np.testing.assert_allclose(got[i], ob.msim_vectors(rows[i]), rtol=3e-6, atol=3e-7)


def test_controllability_matches_reference():
    from musediffusion_amd import metric as mm
    g = load_golden("batch.npz")
    off = g["offsets"]
    toks = np.zeros((12, 256), np.int32)
    lens = np.zeros(12, np.int32)
    for i in range(12):
        s = g["values"][off[i] + 12:off[i + 1]]
        toks[i, :len(s)], lens[i] = s, len(s)
    assert list(mm.Controllability_Pitch(dev(g["ctrl_metas"]), dev(toks), dev(lens))) == g["ctrl_pitch"].tolist()
    assert list(mm.Controllability_Velocity(dev(g["ctrl_metas"]), dev(toks), dev(lens))) == g["ctrl_velocity"].tolist()
