"""Pins oracle/ (plain-C restatement) to golden vectors produced by the
reference's own compiled Cython code (tests/golden/make_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest

import oracle
from npore_amd import synth
from conftest import load_json, enc, expand_cigar, collapse_cigar, GOLDEN


def sha(s):
    return hashlib.sha256(s.encode()).hexdigest()


def test_np_info_golden():
    seqs = load_json("np_info_seqs.json")
    z = np.load(os.path.join(GOLDEN, "np_info.npz"))
    for i, s in enumerate(seqs):
        got = oracle.get_np_info(enc(s))
        assert got.shape == z[f"info_{i}"].shape
        assert np.array_equal(got, z[f"info_{i}"]), (i, s)


def test_np_info_docstring_example():
    # reference src/aln.pyx:182-194
    info = oracle.get_np_info(enc("ATATATATTTTTTAAAGCGCGC"))
    assert info[:, 0, 0].tolist() == [0, 0, 0, 0, 0, 0, 0, 6, 6, 6, 6, 6, 6, 3, 3, 3, 0, 0, 0, 0, 0, 0]
    assert info[:, 1, 0].tolist() == [0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 0, 1, 2, 0, 0, 0, 0, 0, 0]
    assert info[:, 0, 1].tolist() == [4, 3, 4, 3, 4, 3, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 0, 3, 0, 3, 0]
    assert info[:, 1, 1].tolist() == [0, 0, 1, 1, 2, 2, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 2, 0]
    assert not info[:, :, 2].any()


def test_unit_aligns(tables):
    sub, nps = tables
    for c in load_json("unit_aligns.json"):
        ex = expand_cigar(c["cigar"])
        a = oracle.align(enc(c["ref"]), enc(c["seq"]), ex, sub, nps, max_b_rows=20, r=10)
        b = oracle.align(enc(c["ref"]), enc(c["seq"]), ex, sub, nps)
        assert a == c["aln_20_10"], c
        assert b == c["aln_default"], c
        assert collapse_cigar(a) == c["collapsed_20_10"]


def test_reads_e2e_raw(tables):
    sub, nps = tables
    fasta = "".join(l.strip() for l in open(os.path.join(GOLDEN, "data", "ref.fasta")) if not l.startswith(">")).upper()
    want = {r["name"]: r for r in load_json("reads_e2e.json")}
    n = 0
    for line in open(os.path.join(GOLDEN, "data", "reads.sam")):
        if line.startswith("@"):
            continue
        f = line.rstrip("\n").split("\t")
        ex = expand_cigar(f[5]).replace("S", "").replace("H", "")
        start = int(f[3]) - 1
        rlen = sum(1 for ch in ex if ch in "XD=M")
        got, st = oracle.align(enc(fasta[start:start + rlen]), enc(f[9]), ex, sub, nps, return_status=True)
        assert st == 0
        assert got == want[f[0]]["raw_align"], f[0]
        n += 1
    assert n == 10


@pytest.mark.parametrize("big", [False, True])
def test_synthetic(tables, big):
    sub, nps = tables
    recs = [r for r in load_json("synthetic.json") if (r["ref_len"] >= 10_000) == big]
    assert recs
    for r in recs:
        ref, seq, cig = synth.make_pair(r["base_seed"], r["index"], r["ref_len"], r["p_np"], r["p_cnv"], r["mixed"])
        got, st = oracle.align(ref, seq, cig, sub, nps, max_b_rows=r["max_b_rows"], r=r["r"], return_status=True)
        assert st == 0
        assert len(got) == r["len"] and sha(got) == r["sha256"], r
        if "collapsed" in r:
            assert collapse_cigar(got) == r["collapsed"]


def test_edge_inputs(tables):
    sub, nps = tables
    assert oracle.align(enc(""), enc(""), "", sub, nps) == ""
    assert oracle.align(enc("ACGT"), enc(""), "DDDD", sub, nps) == "DDDD"
    assert oracle.align(enc(""), enc("ACGT"), "IIII", sub, nps) == "IIII"
    with pytest.raises(ValueError):
        oracle.align(enc("ACGT"), enc("ACGT"), "===", sub, nps)      # lengths disagree
    with pytest.raises(ValueError):
        oracle.align(enc("ACGT"), enc("ACGT"), "==N=", sub, nps)     # unsupported op


def test_fullsize_digests_are_the_oracles(tables):
    """tests/golden/fullsize_digests.npz (per-read length + sha256[:16] of every full-size configuration, which the GPU
    tests compare all reads against) really is what the pinned oracle gives: spot checks of each configuration."""
    import hashlib
    from npore_amd import synth
    sub, nps = tables
    z = np.load(os.path.join(GOLDEN, "fullsize_digests.npz"))
    # (round 5: c2 / r30 cover what the ranks of an 8-GPU run hold -- indices 0 ... 7 999, and 0 ... 31 999 sampled; c4 is new)
    assert [len(z[k + "_idx"]) for k in ("c2", "r30", "c5", "c3", "c4")] == [8000, 8000, 256, 19000, 17822]
    for name, seed, ref_len, mixed, r, i in (("c2", 2, 10_000, False, 100, 511), ("c2", 2, 10_000, False, 100, 7_777),
                                             ("r30", 2, 10_000, False, 30, 2048), ("r30", 2, 10_000, False, 30, 4000 + 7 * 3999),
                                             ("c3", 3, 10_000, True, 100, 99_990), ("c3", 3, 10_000, True, 100, 4_321),
                                             ("c4", 4, 10_000, True, 100, 7_001), ("c4", 4, 10_000, True, 100, 8000 + 101 * 9821)):
        ref, seq, cig = synth.make_pair(seed, i, ref_len, mixed=mixed)
        s = oracle.align(ref, seq, cig, sub, nps, r=r)
        k = int(np.nonzero(z[name + "_idx"] == i)[0][0])
        assert len(s) == z[name + "_len"][k] and int(hashlib.sha256(s.encode()).hexdigest()[:16], 16) == int(z[name + "_dig"][k])
