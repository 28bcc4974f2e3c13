"""CPU: the product's cell recurrence (npore_amd/csrc/cell.hpp) and packing
(prep.hpp), executed cell by cell by tests/model, against the oracle."""
import numpy as np

import oracle
from model import model
from npore_amd import synth
from conftest import load_json, enc, expand_cigar


def test_np_info_formulation_vs_oracle():
    seqs = [enc(s) for s in load_json("np_info_seqs.json")]
    rng = np.random.default_rng(3)
    for k in range(300):
        n = int(rng.integers(1, 600))
        alpha = rng.integers(0, 5, size=int(rng.integers(1, 4)))
        seqs.append(rng.choice(alpha, size=n).astype(np.uint8))
    for max_l in (100, 7):
        for s in seqs:
            assert np.array_equal(oracle.get_np_info(s, max_l=max_l), model.get_np_info(s, max_l=max_l))


def test_unit_cases(tables):
    sub, nps = tables
    for c in load_json("unit_aligns.json"):
        got, st = model.align(enc(c["ref"]), enc(c["seq"]), expand_cigar(c["cigar"]), sub, nps, max_b_rows=20, r=10)
        assert got == c["aln_20_10"] and st == 0


def test_fuzz(tables):
    sub, nps = tables
    rng = np.random.default_rng(5)
    for k in range(500):
        ref_len = int(rng.integers(1, 900))
        ref, seq, cig = synth.make_pair(77, k, ref_len, float(rng.choice([0, 0.05, 0.15, 0.4])),
                                        float(rng.choice([0, 0.3, 0.9])))
        if k % 5 == 0 and len(ref) > 3:
            ref = ref.copy(); ref[rng.integers(0, len(ref), size=2)] = 0
        if k % 7 == 0 and len(seq) > 3:
            seq = seq.copy(); seq[rng.integers(0, len(seq), size=2)] = 0
        r = int(rng.choice([1, 2, 3, 5, 10, 30, 64, 100]))
        mbr = int(rng.choice([2, 3, 7, 20, 64, 500, 20000]))
        ist, iex = (5.0, 1.0) if k % 4 else (float(rng.integers(1, 8)), float(rng.integers(0, 3)))
        a, sa = oracle.align(ref, seq, cig, sub, nps, indel_start=ist, indel_extend=iex, max_b_rows=mbr, r=r,
                             return_status=True)
        b, sb = model.align(ref, seq, cig, sub, nps, indel_start=ist, indel_extend=iex, max_b_rows=mbr, r=r)
        assert a == b and sa == sb, (k, ref_len, r, mbr)


def test_long_polymers(tables):
    """Homopolymers / STRs longer than max_l and than the device score-table window."""
    sub, nps = tables
    rng = np.random.default_rng(9)
    for k in range(30):
        unit = rng.integers(1, 5, size=int(rng.integers(1, 4))).astype(np.uint8)
        copies = int(rng.integers(20, 140))
        flank = lambda: rng.integers(1, 5, size=int(rng.integers(5, 40))).astype(np.uint8)
        ref = np.concatenate([flank(), np.tile(unit, copies), flank()])
        d = int(rng.integers(-6, 7))
        seq = np.concatenate([ref[:40], np.tile(unit, max(1, copies + d)), ref[-30:]])
        # crude input path: match as far as possible then indel at the end
        m = min(len(ref), len(seq))
        cig = "=" * 0 + "M" * m + "D" * (len(ref) - m) + "I" * (len(seq) - m)
        for r in (10, 30):
            a, sa = oracle.align(ref, seq, cig, sub, nps, r=r, return_status=True)
            b, sb = model.align(ref, seq, cig, sub, nps, r=r)
            assert a == b and sa == sb, (k, r)
