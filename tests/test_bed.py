"""n-polymer BED generation (reference src/bed.py): the device batch
(npore_np_regions) against the literal loop over the oracle's get_np_info, and the numpy restatement of
the reference's bedtools / sort / sed pipeline against brute-force bitmaps."""
import argparse
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO
from npore_amd import bam, bed, cfg
from npore_amd.cig import bases_to_int

DATA = os.path.join(GOLDEN, "data")


def literal_np_regions(np_info, start, max_n):
    """The loop of get_np_regions, src/bed.py:65-72, on one slice: per period, [(start, stop)]."""
    out = [[] for _ in range(max_n)]
    for idx in range(np_info.shape[0]):
        for n in range(1, max_n + 1):
            if np_info[idx, 0, n - 1] and not np_info[idx, 1, n - 1]:
                out[n - 1].append((start + idx, start + idx + n * int(np_info[idx, 0, n - 1])))
    return out


def tricky_sequences():
    rng = np.random.default_rng(3)
    seqs = []
    from npore_amd import synth
    for k in range(6):
        ref, _s, _c = synth.make_pair(41, k, 700 + 531 * k, 0.2, 0.0, False)
        seqs.append(np.asarray(ref, np.uint8))
    rnd = lambda n: rng.integers(1, 5, n).astype(np.uint8)
    seqs.append(np.concatenate([rnd(50), np.zeros(5000, np.uint8), rnd(70)]))                 # assembly gap
    seqs.append(np.concatenate([rnd(10), np.full(3000, 1, np.uint8), rnd(10)]))               # homopolymer beyond max_l
    seqs.append(np.concatenate([rnd(33), np.tile(np.array([3, 3, 1, 1, 4, 2], np.uint8), 700), rnd(5)]))   # satellite, n=6
    seqs.append(np.concatenate([np.tile(np.array([1, 4], np.uint8), 120), np.zeros(3, np.uint8), np.tile(np.array([2, 2, 3], np.uint8), 101)]))
    seqs.append(np.full(64 * 9, 4, np.uint8))                                                 # run = whole slice
    seqs.append(np.tile(np.array([1, 2, 3], np.uint8), 33)[:98])                              # just below / at max_l
    seqs.append(np.tile(np.array([1, 2, 3], np.uint8), 102))
    seqs.append(np.zeros(0, np.uint8))
    seqs.append(np.array([2], np.uint8))
    seqs.append(np.array([1, 1, 1], np.uint8))
    return seqs


# ---------------------------------------------------------------------------
def brute_union(n_ctg, length, ctg, start, stop):
    cov = np.zeros((n_ctg, length + 2), bool)
    for c, s, e in zip(ctg, start, stop):
        cov[c, s:e] = True
    return cov


def test_bed_merge_sort_complement_against_bitmaps():
    rng = np.random.default_rng(1)
    names = ["chr2", "chr10", "chrX", "1", "chr1_alt", "scaffold7"]
    assert bed.sort_names(names) == [(2, "chr2"), (10, "chr10"), (0, "X"), (1, "chr1"), (1, "chr1_alt"), (0, "scaffold7")]
    for trial in range(30):
        m = int(rng.integers(0, 60))
        ctg = np.sort(rng.integers(0, len(names), m)).astype(np.int64)
        start = np.zeros(m, np.int64)
        for c in range(len(names)):
            k = ctg == c
            start[k] = np.sort(rng.integers(0, 300, int(k.sum())))
        stop = start + rng.integers(1, 25, m)
        mc, ms, me = bed.bed_merge(ctg, start, stop)
        # same coverage, and merged intervals are strictly separated (book-ended ones are joined)
        assert np.array_equal(brute_union(len(names), 330, ctg, start, stop), brute_union(len(names), 330, mc, ms, me))
        for i in range(1, len(mc)):
            assert mc[i] != mc[i - 1] or ms[i] > me[i - 1]
        printed, sc, ss, se = bed.bed_sort(names, mc, ms, me)
        bare = [nm[3:] if nm.startswith("chr") else nm for nm in names]
        lines = [(bed.sort_names(names)[c][0], bare[c], s, e) for c, s, e in zip(sc, ss, se)]
        assert lines == sorted(lines) and sorted(zip(sc, ss, se)) == sorted(zip(mc, ms, me))
        # merging the sorted lines again changes nothing (blocks of one contig, not ascending contig indices)
        assert [x.tolist() for x in bed.bed_merge(sc, ss, se)] == [sc.tolist(), ss.tolist(), se.tolist()]
    # complement: per contig with records, what no record covers within the genome length
    printed = ["chr1", "chr2", "X"]
    ctg = np.array([0, 0, 2], np.int64); start = np.array([0, 10, 5], np.int64); stop = np.array([4, 20, 50], np.int64)
    cc, cs, ce = bed.bed_complement(printed, ctg, start, stop, {"chr1": 30, "chr2": 99, "X": 50})
    assert list(zip(cc, cs, ce)) == [(0, 4, 10), (0, 20, 30), (2, 0, 5)]


def test_ranges_and_regions(tmp_path):
    assert bed.get_ranges([("a", 0, 25), ("b", 5, 6)], 10) == [("a", 0, 10), ("a", 10, 20), ("a", 20, 25), ("b", 5, 6)]
    ref = {"a": "ACGT" * 10, "b": "AC" * 10}
    bp = tmp_path / "r.bed"
    bp.write_text("a\t0\t40\nb\t2\t20\n")
    old = cfg.args
    try:
        cfg.args = argparse.Namespace(contig=None, contigs=None, contig_beg=None, contig_end=None, bed=str(bp), ref="x.fa")
        assert bed.get_regions(ref) == [("a", 0, 40), ("b", 2, 20)]
        cfg.args = argparse.Namespace(contig="a", contigs=None, contig_beg=5, contig_end=1000, bed=str(bp), ref="x.fa")
        assert bed.get_regions(ref) == [("a", 5, 39)]
        cfg.args = argparse.Namespace(contig=None, contigs="b,a", contig_beg=None, contig_end=None, bed=str(bp), ref="x.fa")
        assert bed.get_regions(ref) == [("b", 0, 19), ("a", 0, 39)]
        for bad in (dict(contig="zz", contigs=None), dict(contig="a", contigs="b"), dict(contig=None, contigs="a,zz")):
            cfg.args = argparse.Namespace(contig_beg=None, contig_end=None, bed=str(bp), ref="x.fa", **bad)
            with pytest.raises(SystemExit):
                bed.get_regions(ref)
    finally:
        cfg.args = old


def test_save_beds_from_oracle_annotation(tmp_path):
    """The whole file pipeline with the regions taken from the oracle's get_np_info (no GPU): every output file
    has the coverage a brute-force bitmap gives."""
    import oracle
    ref_seqs = bam.read_fasta(os.path.join(DATA, "synth_std.fasta"))
    regions = [(c, 0, len(s)) for c, s in ref_seqs.items()]
    ranges = bed.get_ranges(regions, 4000)
    names = list(ref_seqs)
    per = [([], [], []) for _ in range(6)]
    for c, s, e in ranges:
        info = np.asarray(oracle.get_np_info(bases_to_int(ref_seqs[c][s:e])))
        for n, regs in enumerate(literal_np_regions(info, s, 6)):
            for a, b in regs:
                per[n][0].append(names.index(c)); per[n][1].append(a); per[n][2].append(b)
    per_n = [tuple(np.array(x, np.int64) for x in p) for p in per]
    bp = tmp_path / "g.bed"
    bp.write_text("".join(f"{c}\t0\t{len(s)}\n" for c, s in ref_seqs.items()))
    prefix = str(tmp_path / "np")
    bed.save_np_region_beds(names, per_n, prefix, str(bp))
    assert (tmp_path / "g.genome").read_text() == "".join(f"{c}\t{len(s)}\n" for c, s in ref_seqs.items())
    L = max(len(s) for s in ref_seqs.values()) + 8

    def read(path):
        rows = [l.split("\t") for l in open(path).read().splitlines()]
        return [(names.index(r[0]), int(r[1]), int(r[2])) for r in rows]
    allcov = np.zeros((len(names), L), bool)
    for n in range(1, 7):
        rows = read(f"{prefix}_{n}.bed")
        want = np.zeros((len(names), L), bool)
        for c, a, b in zip(*per_n[n - 1]):
            want[c, max(0, a - 1):b + 1] = True
        got = np.zeros((len(names), L), bool)
        for c, a, b in rows:
            got[c, a:b] = True
        assert np.array_equal(got, want) and rows == sorted(rows)
        allcov |= want
    rows = read(f"{prefix}_all.bed")
    got = np.zeros((len(names), L), bool)
    for c, a, b in rows:
        got[c, a:b] = True
    assert np.array_equal(got, allcov)
    comp = np.zeros((len(names), L), bool)
    for c, a, b in read(f"{prefix}_0.bed"):
        comp[c, a:b] = True
    for ci, c in enumerate(names):
        if allcov[ci].any():
            assert np.array_equal(comp[ci, :len(ref_seqs[c])], ~allcov[ci, :len(ref_seqs[c])])
        else:
            assert not comp[ci].any()                          # -L: contigs without records are left out


# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx(tables):
    from npore_amd import aln
    c = aln.Context(*tables)
    yield c
    c.close()


@pytest.mark.gpu
def test_np_info_long_runs_equal_oracle(ctx):
    """get_np_info on the device where runs exceed what the kernel follows (assembly gaps, satellites)."""
    import oracle
    for seq in tricky_sequences():
        assert np.array_equal(ctx.get_np_info(seq), np.asarray(oracle.get_np_info(seq)))


@pytest.mark.gpu
def test_np_regions_equal_literal_loop(ctx):
    import oracle
    seqs = tricky_sequences()
    got = ctx.np_regions(seqs)
    assert len(got) == 6
    total = 0
    for k, seq in enumerate(seqs):
        want = literal_np_regions(np.asarray(oracle.get_np_info(seq)), 0, 6)
        for n in range(6):
            pos, reps = got[n][k]
            assert [(int(p), int(p) + (n + 1) * int(r)) for p, r in zip(pos, reps)] == want[n], (k, n)
            total += len(pos)
    assert total > 300
    assert ctx.np_regions([]) == [[] for _ in range(6)]
    assert all(len(p) == 0 for n in range(6) for p, _r in ctx.np_regions([np.zeros(0, np.uint8)])[n])


@pytest.mark.gpu
def test_bed_cli(tmp_path):
    ref_seqs = bam.read_fasta(os.path.join(DATA, "synth_std.fasta"))
    bp = tmp_path / "g.bed"
    bp.write_text("".join(f"{c}\t0\t{len(s)}\n" for c, s in ref_seqs.items()))
    prefix = str(tmp_path / "cli")
    res = subprocess.run([sys.executable, "-m", "npore_amd.bed", "--ref", os.path.join(DATA, "synth_std.fasta"), "--bed", str(bp),
                          "--out_prefix", prefix, "-chunk_width", "4000"], cwd=REPO, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    # same files as the oracle-fed pipeline
    import oracle
    names = list(ref_seqs)
    per = [([], [], []) for _ in range(6)]
    for c, s, e in bed.get_ranges([(c, 0, len(s)) for c, s in ref_seqs.items()], 4000):
        info = np.asarray(oracle.get_np_info(bases_to_int(ref_seqs[c][s:e])))
        for n, regs in enumerate(literal_np_regions(info, s, 6)):
            for a, b in regs:
                per[n][0].append(names.index(c)); per[n][1].append(a); per[n][2].append(b)
    want_prefix = str(tmp_path / "want")
    bed.save_np_region_beds(names, [tuple(np.array(x, np.int64) for x in p) for p in per], want_prefix, str(bp))
    for sfx in ("_0", "_1", "_2", "_3", "_4", "_5", "_6", "_all"):
        assert open(f"{prefix}{sfx}.bed").read() == open(f"{want_prefix}{sfx}.bed").read(), sfx
    assert os.path.getsize(f"{prefix}_1.bed") > 100
