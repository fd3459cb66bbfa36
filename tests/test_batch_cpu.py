"""CPU: the numpy oracle of the batch producers / token validators (oracle/batch.py) against the reference's own outputs
recorded in tests/golden/batch.npz (tools/make_golden_batch.py).  Integer work: everything bit-exact."""
import numpy as np

from conftest import load_golden
from oracle import batch as ob


def rows(g, key="values"):
    off = g["offsets"]
    return [g[key][off[i]:off[i + 1]] for i in range(len(off) - 1)]


def test_corruptions_match_reference():
    g = load_golden("batch.npz")
    seqs = rows(g)
    for i, s in enumerate(seqs):
        assert np.array_equal(ob.masking_token(s, rows(g, "mt_u")[i], 0.3), rows(g, "mt_out")[i]), ("mt", i)
        assert np.array_equal(ob.masking_note(s, rows(g, "mn_u")[i], 0.5), rows(g, "mn_out")[i]), ("mn", i)
        assert np.array_equal(ob.randomize_note(s, rows(g, "rn_u")[i], rows(g, "rn_new")[i], 0.5), rows(g, "rn_out")[i]), ("rn", i)
        assert np.array_equal(ob.random_rotating(s, g["rr_pairs"][i]), rows(g, "rr_out")[i]), ("rr", i)
    # the fixture exercises every branch: something was masked / randomised / rotated, and a truncated note was skipped
    assert (g["mt_out"] != g["values"]).any() and (g["mn_out"] != g["values"]).any()
    assert (g["rn_out"] != g["values"]).any() and (g["rr_out"] != g["values"]).any()


def test_collate_and_meta_to_batch_match_reference():
    g = load_golden("batch.npz")
    seqs, mt = rows(g)[:6], rows(g, "mt_out")[:6]
    masks = []
    for s in seqs:
        m = np.ones(len(s), np.int32)
        m[:12] = 0
        masks.append(m)
    for L, tag in ((None, "max"), (256, "256")):
        col = ob.collate(mt, masks, seqs, L)
        for k in ("input_ids", "correct_ids", "input_mask", "length"):
            assert np.array_equal(col[k], g["collate_%s_%s" % (tag, k)]), (tag, k)
    ids, msk = ob.meta_to_batch(g["m2b_meta"], 5, 64)
    assert np.array_equal(ids, g["m2b_ids"]) and np.array_equal(msk, g["m2b_mask"])


def test_validators_match_reference():
    g = load_golden("batch.npz")
    for i in range(len(g["val_len"])):
        got = ob.validate(g["val_tokens"][i], int(g["val_len"][i]))
        assert tuple(int(v) for v in g["val_result"][i]) == got, (i, got, g["val_result"][i])
    r = g["val_result"]
    assert (r[:, 0] == -1).any() and (r[:, 1] == 0).any() and (r[:, 2] == 0).any() and (r[:, 2] == 1).any()


def test_msim_vectors_and_onnc_match_reference():
    g = load_golden("batch.npz")
    seqs = rows(g)
    vec = np.stack([ob.msim_vectors(s[12:]) for s in seqs])
    np.testing.assert_allclose(vec, g["msim_vectors"], rtol=2e-6, atol=2e-7)       # torch.norm's fp32 summation order differs
    val, sim, most = ob.onnc(g["msim_vectors"])
    assert np.array_equal(most, g["onnc_mostsim"]) and abs(val - float(g["onnc"])) < 1e-7
    np.testing.assert_allclose(sim, g["onnc_msim"], rtol=1e-5, atol=1e-7)
    r = g["msim_vectors"]
    assert abs(float((r[0, :32] @ r[1, :32]) * (r[0, 32:44] @ r[1, 32:44]) * (r[0, 44:] @ r[1, 44:])) - float(g["msim_01"])) < 1e-6


def test_controllability_matches_reference():
    g = load_golden("batch.npz")
    (pt, pw), (vt, vw) = ob.controllability(g["ctrl_metas"], [s[12:] for s in rows(g)])
    assert [pt, pw] == g["ctrl_pitch"].tolist() and [vt, vw] == g["ctrl_velocity"].tolist()
    assert pw > 0 and vw > 0 and vw < vt
