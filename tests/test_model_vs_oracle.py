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


def small_tables(nps, max_n, max_l, seed):
    """Tables of another shape: the shipped ones cut / edge-padded, entries jittered so that clamped rows differ."""
    rng = np.random.default_rng(seed)
    m = min(max_l, 100)
    t = nps[:max_n, :m + 1, :m + 1]
    if max_l > 100:
        t = np.pad(t, ((0, 0), (0, max_l - 100), (0, max_l - 100)), mode="edge")
    t = np.ascontiguousarray(t).copy()
    t[:, 3:, :] += (rng.integers(0, 8, t[:, 3:, :].shape) / 8.0).astype(np.float32)
    return t


def polymer_pairs(seed, count):
    """Reads whose n-polymers reach and exceed small max_l values (copies 3..60)."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(count):
        parts_r, parts_s, cig = [], [], []
        for _ in range(int(rng.integers(2, 7))):
            f = rng.integers(1, 5, size=int(rng.integers(3, 25))).astype(np.uint8)
            parts_r.append(f); parts_s.append(f); cig.append("=" * len(f))
            unit = rng.integers(1, 5, size=int(rng.integers(1, 7))).astype(np.uint8)
            c = int(rng.integers(3, 60)); d = int(rng.integers(-3, 4)); c2 = max(1, c + d)
            parts_r.append(np.tile(unit, c)); parts_s.append(np.tile(unit, c2))
            m = min(c, c2) * len(unit)
            cig.append("=" * m + ("D" * ((c - c2) * len(unit)) if c > c2 else "I" * ((c2 - c) * len(unit))))
        out.append((np.concatenate(parts_r), np.concatenate(parts_s), "".join(cig)))
    return out


def test_other_table_shapes(tables):
    """max_n / max_l other than 6 / 100: np_score clamps rows and call lengths to max_l - 1, and with max_l below
    the device table's 32 rows the capped repeat count max_l itself goes through the table-address shortcut."""
    sub, nps = tables
    # (6, 5), (6, 3), (4, 2): periods ABOVE max_l -- np_score's `n > max_n` test reads n > max_l (src/aln.pyx:265 as
    # called) and such a candidate scores the constant 100
    for max_n, max_l in ((6, 20), (4, 20), (1, 5), (6, 31), (6, 32), (3, 127), (6, 5), (6, 3), (4, 2)):
        t = small_tables(nps, max_n, max_l, max_l)
        for k, (ref, seq, cig) in enumerate(polymer_pairs(100 + max_l, 25)):
            r = (5, 30, 64)[k % 3]
            a, sa = oracle.align(ref, seq, cig, sub, t, max_b_rows=(20000, 64)[k % 2], r=r, max_n=max_n, max_l=max_l,
                                 return_status=True)
            b, sb = model.align(ref, seq, cig, sub, t, max_b_rows=(20000, 64)[k % 2], r=r, max_n=max_n, max_l=max_l)
            assert a == b and sa == sb, (max_n, max_l, k)


def test_ties_and_negative_scores(tables):
    """Tables full of exact ties (every score the same; scores on a coarse grid, some negative): the MAT state picks
    the FIRST candidate in the reference's order that attains the minimum -- cell.hpp does that on the plain path by
    minima and equality tests (Env::MIN3), the oracle by the reference's chain of strict '<'."""
    sub0, nps0 = tables
    rng = np.random.default_rng(21)
    variants = []
    eq_sub, eq_nps = sub0.copy(), nps0.copy()
    eq_sub[:] = 1.0; eq_nps[:] = 1.0
    variants.append((eq_sub, eq_nps))
    g_sub = (rng.integers(0, 24, (5, 5)) / 4.0).astype(np.float32); g_sub[0, :] = 0; g_sub[:, 0] = 0
    g_nps = (rng.integers(-2, 40, nps0.shape) / 4.0).astype(np.float32); g_nps[:, :3, :] = 20.0
    variants.append((g_sub, g_nps))
    for sub, nps in variants:
        for k in range(60):
            ref, seq, cig = synth.make_pair(91, k, int(rng.integers(5, 500)), float(rng.choice([0.05, 0.2, 0.4])),
                                            float(rng.choice([0.3, 0.9])))
            r = int(rng.choice([2, 5, 12, 30, 70]))
            ist, iex = float(rng.integers(1, 4)), float(rng.integers(0, 2))
            a, sa = oracle.align(ref, seq, cig, sub, nps, indel_start=ist, indel_extend=iex, r=r, return_status=True)
            b, sb = model.align(ref, seq, cig, sub, nps, indel_start=ist, indel_extend=iex, r=r)
            assert a == b and sa == sb, (k, r)


def test_period_above_max_l_scores_100(tables):
    """max_l < max_n: a 6-mer repeated five times under max_l = 5 (found by the GPU fuzz, round 3).  np_score is called
    with max_l where its signature says max_n (src/aln.pyx:615,629,650,663), so `n > max_n` (:265) makes every
    period-6 LEN / SHR candidate cost 100 -- the recurrence must not look its score up in the table."""
    sub, nps = tables
    t = small_tables(nps, 6, 5, 5)
    enc = lambda s_: np.array(["NACGT".index(c) for c in s_], np.uint8)
    ref, seq, cig = enc("TACATC" * 5), enc("TACCATCTACATGTTCATCTACATCTAGATC"), "===I========X=X============X==="
    for r in (3, 10, 30):
        for mbr in (20000, 16):
            a, sa = oracle.align(ref, seq, cig, sub, t, max_b_rows=mbr, r=r, max_n=6, max_l=5, return_status=True)
            b, sb = model.align(ref, seq, cig, sub, t, max_b_rows=mbr, r=r, max_n=6, max_l=5)
            assert a == b and sa == sb, (r, mbr)
    assert oracle.align(ref, seq, cig, sub, t, r=10, max_n=6, max_l=5) == "==I=========X=X============X==="


def _annot_cases():
    seqs = [enc(s) for s in load_json("np_info_seqs.json")]
    rng = np.random.default_rng(31)
    for k in range(120):                              # low-entropy random sequences: runs of every period, incl. N
        n = int(rng.integers(1, 400))
        alpha = rng.integers(0, 5, size=int(rng.integers(1, 4)))
        seqs.append(rng.choice(alpha, size=n).astype(np.uint8))
    A, C, G, T = 1, 2, 3, 4
    pieces = [[A] * 150, [A, C] * 70, [A, C, G] * 45, [A] * 40 + [C] * 33, [A, A, C, A, A, C] * 30, [T] * 101, [C, A, G, T] * 36,
              [A] * 12 + [A, C] * 9 + [A, C, G] * 7, [0] * 80, [A] * 8 + [C, A, A] * 5, [A, A, A, A, G] * 4, [G] * 65, [A, C, A, C, A, G] * 13]
    for k in range(40):                               # engineered: polymers longer than max_l / than a window, nested periods
        s = []
        for b in rng.permutation(len(pieces))[:5]:
            s += [int(x) for x in rng.integers(0 if k % 4 == 0 else 1, 5, int(rng.integers(0, 20)))] + pieces[b]
        seqs.append(np.array(s, np.uint8))
    for k in range(30):                               # the bench generator's reads
        seqs.append(synth.make_pair(41, k, 700, 0.15, 0.3)[0])
    return seqs


def test_wave_annotation_formulation_vs_oracle():
    """The wave-local restatement of get_np_info that csrc/annot_wave.hpp evaluates (tests/model/annot_wave_model.py):
    the stride-n recurrence and its per-window closed form, against the oracle's literal loop -- at several window
    widths (narrow windows put every carry / look-ahead case on a window boundary) and max_l / max_n."""
    from model import annot_wave_model as awm
    seqs = _annot_cases()
    for max_n, max_l in ((6, 100), (6, 7), (6, 2), (4, 3), (6, 127), (3, 1)):
        for k, s in enumerate(seqs):
            want = oracle.get_np_info(s, max_n=max_n, max_l=max_l)
            assert np.array_equal(awm.recurrence(s, max_n, max_l), want), (max_n, max_l, k)
            for W in ((64, 8) if max_l != 127 else (64,)):
                got = awm.windows(s, max_n, max_l, W=W)
                assert np.array_equal(got, want), (max_n, max_l, W, k, np.argwhere(got != want)[:4])


def test_segments_with_warm_up_equal_the_whole_sequence():
    """What csrc/annot_wave.hpp's np_info_wave_kernel relies on (the get_np_info() API and the genome-scale region
    kernels: one wave per segment of a long sequence): get_np_info of the suffix that starts `warm` = sum over n of
    (max_l + 2) n positions in front of a segment, taken as a sequence of its own, equals get_np_info of the whole
    sequence on the segment -- for arrays of every period far longer than max_l, nested periods, two-letter sequences
    (periodic everywhere) and N stretches, segment starts drawn at random (also inside the arrays)."""
    rng = np.random.default_rng(5)

    def make(n, kind):
        if kind == 0:
            s = synth.make_ref(rng, n, 0.15)[0].copy()
        elif kind == 1:                              # long arrays of every period with short spacers
            parts = []
            while sum(map(len, parts)) < n:
                per = int(rng.integers(1, 7))
                parts.append(np.tile(rng.integers(1, 5, per).astype(np.uint8), int(rng.integers(3, 400))))
                parts.append(rng.integers(1, 5, int(rng.integers(0, 6))).astype(np.uint8))
            s = np.concatenate(parts)[:n]
        else:                                        # two letters, with planted arrays
            s = rng.integers(1, 3, n).astype(np.uint8)
            for _ in range(12):
                a, per = int(rng.integers(0, n - 900)), int(rng.integers(1, 7))
                s[a:a + 800] = np.tile(s[a:a + per], 800 // per + 1)[:800]
        for _ in range(3):
            a = int(rng.integers(0, n - 50))
            s[a:a + int(rng.integers(1, 40))] = 0
        return s

    for max_n, max_l in ((6, 100), (6, 127), (4, 20)):
        warm = (sum((max_l + 2) * n for n in range(1, max_n + 1)) + 63) & ~63
        for trial in range(9):
            s = make(16000, trial % 3)
            full = oracle.get_np_info(s, max_n=max_n, max_l=max_l)
            for s0 in rng.integers(warm, 14000, 6):
                s0 = int(s0) & ~63                   # segments begin on window boundaries
                g0 = s0 - warm
                part = oracle.get_np_info(s[g0:], max_n=max_n, max_l=max_l)
                assert np.array_equal(part[s0 - g0:s0 - g0 + 1500], full[s0:s0 + 1500]), (max_n, max_l, trial, s0)
