"""standardize_vcf path: VCF reader / split / apply / CIGAR -> records /
merge (npore_amd/vcf.py, counterparts of reference src/vcf.py) and the realign_hap batch in the middle.

CPU tests pin the host logic (hand-derived expectations on the reference's own fixture, a literal
per-character restatement of gen_vcf's loop, round trips) and the golden vectors against the oracle;
GPU tests run the whole pipeline through the C ABI and compare with what the reference's compiled
realign_hap returned (tests/golden/std_vcf.json, made by tests/golden/make_golden_vcf.py)."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO, load_json
from npore_amd import bam, cfg, vcf as V
from npore_amd.cig import bases_to_int, collapse_cigar, standardize

DATA = os.path.join(GOLDEN, "data")


def sha(s):
    return hashlib.sha256(s.encode()).hexdigest()


def load_case(tag):
    g = load_json("std_vcf.json")[tag]
    ref_seqs = bam.read_fasta(os.path.join(DATA, g["fasta"]))
    vcf = V.VcfFile(os.path.join(DATA, g["vcf"]))
    old = cfg.args
    try:
        import argparse
        cfg.args = argparse.Namespace(max_n=6, max_l=100, contig=None, contigs=None, contig_beg=None, contig_end=None)
        regions = V.get_vcf_regions(ref_seqs, vcf)
    finally:
        cfg.args = old
    assert [list(r) for r in regions] == g["regions"]
    return g, ref_seqs, vcf, regions


def haps_of(g, ref_seqs, vcf, regions):
    r1, r2 = V.split_vcf(vcf, regions)
    return V.apply_vcf(r1, 1, ref_seqs, regions, g["min_qual"]) + V.apply_vcf(r2, 2, ref_seqs, regions, g["min_qual"])


# ---------------------------------------------------------------------------
# literal restatement of the per-character loop of gen_vcf (src/vcf.py:300-378), the checker of gen_records
def gen_records_literal(hap_data):
    out = []
    for contig, _hap, seq, ref, cigar in hap_data:
        ref_ptr = seq_ptr = cig_ptr = 0
        n = len(cigar)
        while cig_ptr < n:
            op = cigar[cig_ptr]
            if op == "=":
                ref_ptr += 1; seq_ptr += 1; cig_ptr += 1
            elif op == "X" or op == "M":
                if op == "X" or ref[ref_ptr] != seq[seq_ptr]:
                    out.append((contig, ref_ptr + 1, (ref[ref_ptr], seq[seq_ptr])))
                ref_ptr += 1; seq_ptr += 1; cig_ptr += 1
            elif op == "D":
                k = 0
                while cig_ptr < n and cigar[cig_ptr] == "D":
                    k += 1; cig_ptr += 1
                if ref_ptr > 0:
                    out.append((contig, ref_ptr, (ref[ref_ptr - 1:ref_ptr + k], ref[ref_ptr - 1])))
                ref_ptr += k
            elif op == "I":
                k = 0
                while cig_ptr < n and cigar[cig_ptr] == "I":
                    k += 1; cig_ptr += 1
                if ref_ptr > 0 and seq_ptr > 0:
                    out.append((contig, ref_ptr, (ref[ref_ptr - 1], ref[ref_ptr - 1] + seq[seq_ptr:seq_ptr + k])))
                seq_ptr += k
            else:
                raise ValueError(op)
    return out


def test_fixture_split_and_apply():
    """The reference's own test input (test/test_std_vcf.vcf + test_std_ref.fasta), expectations derived by hand
    from src/vcf.py:37-137 and :214-272."""
    g, ref_seqs, vcf, regions = load_case("fixture")
    assert regions == [("chr18", 0, 31), ("chr19", 0, 31)]       # chr20/21: header only, not in the FASTA
    r1, r2 = V.split_vcf(vcf, regions)
    assert [(r.contig, r.pos, r.alleles) for r in r1] == [("chr18", 1, ("A", "G")), ("chr19", 1, ("C", "CAA")),
                                                           ("chr19", 5, ("G", "GAG")), ("chr19", 15, ("C", "G"))]
    assert [(r.contig, r.pos, r.alleles) for r in r2] == [("chr18", 1, ("A", "G")), ("chr18", 3, ("A", "ACCCTA")),
                                                           ("chr19", 1, ("C", "CAA")), ("chr19", 15, ("C", "G"))]
    h = {(x[0], x[1]): x for x in V.apply_vcf(r1, 1, ref_seqs, regions) + V.apply_vcf(r2, 2, ref_seqs, regions)}
    ref18, ref19 = ref_seqs["chr18"], ref_seqs["chr19"]
    assert h[("chr18", 1)][2:] == ("G" + ref18[1:], ref18, "X" + "=" * 31)
    assert h[("chr18", 2)][2:] == ("GC" + "ACCCTA" + ref18[3:], ref18, "X==" + "IIIII" + "=" * 29)
    assert h[("chr19", 1)][2:] == ("CAA" + ref19[1:4] + "GAG" + ref19[5:14] + "G" + ref19[15:], ref19,
                                   "=II" + "===" + "=II" + "=" * 9 + "X" + "=" * 17)
    assert h[("chr19", 2)][2:] == ("CAA" + ref19[1:14] + "G" + ref19[15:], ref19, "=II" + "=" * 13 + "X" + "=" * 17)


@pytest.mark.parametrize("tag", ["fixture", "synthetic", "synthetic_q30"])
def test_apply_vcf_matches_golden_inputs_and_oracle(tables, tag):
    """The haplotype tuples are the ones the golden vectors were made from, every sequence/CIGAR pair is
    consistent, and oracle align() + one-pass standardisation reproduces what the reference's compiled
    realign_hap returned."""
    import oracle
    sub, nps = tables
    g, ref_seqs, vcf, regions = load_case(tag)
    haps = haps_of(g, ref_seqs, vcf, regions)
    assert len(haps) == len(g["haps"])
    for h, want in zip(haps, g["haps"]):
        contig, hap, seq, ref, cig = h
        assert (contig, hap, len(seq), sha(seq), sha(cig)) == (want["contig"], want["hap"], want["seq_len"],
                                                               want["seq_sha256"], want["cigar_sha256"])
        assert sum(c != "I" for c in cig) == len(ref) and sum(c != "D" for c in cig) == len(seq)
        if len(seq) > 8000:
            continue                                   # the Python standardisation is slow; the GPU test covers these
        raw = oracle.align(bases_to_int(ref), bases_to_int(seq), cig, sub, nps, r=30)
        final = standardize(raw, bases_to_int(ref), bases_to_int(seq))
        assert (len(final), sha(final)) == (want["final_len"], want["final_sha256"])
        if "final_collapsed" in want:
            assert collapse_cigar(final) == want["final_collapsed"]


def test_min_qual_and_overlap_rules():
    """apply_vcf's special cases (src/vcf.py:226-243): quality filter, variants overlapping a previous deletion."""
    ref = {"c": "ACGTACGTACGTACGT"}
    regions = [("c", 0, 15)]
    R = lambda pos, a0, a1, q=60.0: V.VcfRecord("c", pos, (a0, a1), q)
    # deletion of CGT after A(1); an insertion inside it is kept, a SNP inside it is dropped,
    # a deletion whose anchor is the last deleted base is kept
    recs = [R(1, "ACGT", "A"), R(3, "G", "GTT"), R(3, "G", "C"), R(4, "TA", "T"), R(9, "A", "C", 10.0), R(11, "G", "T", None)]
    (_c, _h, seq, _r, cig), = V.apply_vcf(recs, 1, ref, regions)
    assert cig == "=DDD" + "II" + "D" + "===" + "X" + "=" + "X" + "=" * 5 and seq == "A" + "TT" + "CGT" + "C" + "C" + "T" + "TACGT"
    (_c, _h, seq, _r, cig), = V.apply_vcf(recs, 1, ref, regions, min_qual=30)
    assert cig == "=DDD" + "II" + "D" + "=" * 11 and seq == "A" + "TT" + "CGTACGTACGT"      # QUAL 10 and '.' dropped
    # region end is exclusive (tabix fetch): a variant at the last base of the contig is not applied
    (_c, _h, seq, _r, cig), = V.apply_vcf([R(16, "T", "A")], 1, ref, regions)
    assert seq == ref["c"] and cig == "=" * 16


def test_gen_records_equals_literal_loop():
    rng = np.random.default_rng(5)
    for trial in range(60):
        n = int(rng.integers(1, 400))
        ops = "".join(rng.choice(list("==MMMXDI" if trial % 3 else "MDI"), n))
        ref = "".join(rng.choice(list("ACGT"), sum(c != "I" for c in ops)))
        seq = "".join(rng.choice(list("ACGT"), sum(c != "D" for c in ops)))
        hd = [("c", 1, seq, ref, ops)]
        got = [(r.contig, r.pos, r.alleles) for r in V.gen_records(hd)]
        assert got == gen_records_literal(hd), ops
        assert all(r.qual == 60.0 and r.filter == "PASS" for r in V.gen_records(hd))
    assert V.gen_records([("c", 1, "", "", "")]) == []
    with pytest.raises(SystemExit):
        V.gen_records([("c", 1, "A", "A", "S")])


def test_apply_then_gen_round_trip_and_merge():
    """Simple, well separated variants survive apply_vcf -> gen_records unchanged; merge_records rebuilds the
    genotypes (src/vcf.py:141-210)."""
    rng = np.random.default_rng(9)
    ref = "".join(rng.choice(list("ACGT"), 4000))
    regions = [("c", 0, len(ref) - 1)]
    haps = {1: [], 2: []}
    both = []
    pos = 5
    while pos < len(ref) - 30:
        r0 = ref[pos - 1]
        kind = int(rng.integers(0, 3))
        if kind == 0:
            alleles = (r0, "ACGT".replace(r0, "")[int(rng.integers(0, 3))])
        elif kind == 1:
            alleles = (r0, r0 + "".join(rng.choice(list("ACGT"), int(rng.integers(1, 6)))))
        else:
            alleles = (ref[pos - 1:pos + int(rng.integers(1, 6))], r0)
        gt = ((1, 1), (1, 0), (0, 1))[int(rng.integers(0, 3))]
        rec = V.VcfRecord("c", pos, alleles, 60.0, gt=gt)
        both.append(rec)
        for k in (1, 2):
            if gt[k - 1]:
                haps[k].append(rec)
        pos += int(rng.integers(10, 40))
    out = {}
    for k in (1, 2):
        hd = V.apply_vcf(haps[k], k, {"c": ref}, regions)
        out[k] = V.gen_records(hd)
        assert [(r.pos, r.alleles) for r in out[k]] == [(r.pos, r.alleles) for r in haps[k]]
    merged = V.merge_records(out[1], out[2], regions)
    assert [(r.pos, r.alleles, r.gt) for r in merged] == [(r.pos, r.alleles, r.gt) for r in both]
    # same position, different alleles -> two lines
    a, b = V.VcfRecord("c", 10, ("A", "G"), 60.0), V.VcfRecord("c", 10, ("A", "T"), 60.0)
    assert [(r.alleles, r.gt) for r in V.merge_records([a], [b], regions)] == [(("A", "G"), (1, 0)), (("A", "T"), (0, 1))]


def test_split_vcf_genotype_cases():
    """src/vcf.py:59-122: multi-allelic sites, spanning deletions, 0|0, missing genotypes."""
    import argparse
    text = ("##fileformat=VCFv4.2\n##contig=<ID=c,length=100>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"
            "c\t5\t.\tA\tG,T\t50\tPASS\t.\tGT\t1|2\n"
            "c\t9\t.\tA\tG,*\t50\tPASS\t.\tGT:GQ\t2|1:9\n"
            "c\t12\t.\tA\tG\t50\tPASS\t.\tGT\t0|0\n"
            "c\t15\t.\tA\tA\t50\tPASS\t.\tGT\t0|0\n"
            "c\t20\t.\tA\tC\t.\tPASS\t.\tGT\t0/1\n"
            "c\t25\t.\tA\tC\t50\tPASS\t.\tGT\t1\n")
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "t.vcf.gz")
        V.bgzf_write(p, text.encode())
        vcf = V.VcfFile(p)
    assert vcf.samples == ["S"] and vcf.header_contigs == ["c"]
    r1, r2 = V.split_vcf(vcf, [("c", 0, 99)])
    k = lambda rs: [(r.pos, r.alleles) for r in rs]
    assert k(r1) == [(5, ("A", "G")), (12, ("A", "G")), (25, ("A", "C"))]
    assert k(r2) == [(5, ("A", "T")), (9, ("A", "G")), (12, ("A", "G")), (20, ("A", "C")), (25, ("A", "C"))]
    assert r2[3].qual is None and r1[0].qual == 50.0


def test_write_and_reread(tmp_path):
    recs = [V.VcfRecord("c1", 3, ("A", "AGG"), 60.0, gt=(1, 0)), V.VcfRecord("c2", 7, ("ACC", "A"), 60.0, gt=(1, 1))]
    hd = [("c1", 1, "", "A" * 50, ""), ("c2", 1, "", "C" * 70, "")]
    for name in ("o.vcf.gz", "o.vcf"):
        p = V.write_vcf(str(tmp_path / name), V.gen_header(hd), recs)
        back = V.VcfFile(p)
        assert back.header[:4] == ["##fileformat=VCFv4.2", '##FILTER=<ID=PASS,Description="All filters passed">',
                                   "##contig=<ID=c1,length=50>", "##contig=<ID=c2,length=70>"]
        assert back.samples == ["SAMPLE"]
        got = [r for c in back.contigs for r in back.by_contig[c]]
        assert [(r.contig, r.pos, r.alleles, r.qual, r.gt, r.filter) for r in got] == \
               [(r.contig, r.pos, r.alleles, r.qual, r.gt, r.filter) for r in recs]
    raw = open(tmp_path / "o.vcf.gz", "rb").read()
    assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC" and raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    assert "c1\t3\t.\tA\tAGG\t60\tPASS\t.\tGT\t1|0" in open(tmp_path / "o.vcf").read()


def test_region_selection_errors():
    import argparse
    ref = {"a": "ACGT" * 10, "b": "AC" * 10}

    class FakeVcf:
        header_contigs = ["a", "zz", "b"]

        def fetch(self, c, s, e):
            return [1] if c == "a" else []
    old = cfg.args
    try:
        cfg.args = argparse.Namespace(contig="a", contigs=None, contig_beg=None, contig_end=None)
        assert V.get_vcf_regions(ref, FakeVcf()) == [("a", 0, 39)]
        cfg.args = argparse.Namespace(contig="a", contigs=None, contig_beg=3, contig_end=9)
        assert V.get_vcf_regions(ref, FakeVcf()) == [("a", 3, 9)]
        cfg.args = argparse.Namespace(contig=None, contigs="b,a", contig_beg=None, contig_end=None)
        assert V.get_vcf_regions(ref, FakeVcf()) == [("b", 0, 19), ("a", 0, 39)]
        cfg.args = argparse.Namespace(contig=None, contigs=None, contig_beg=None, contig_end=None)
        assert V.get_vcf_regions(ref, FakeVcf()) == [("a", 0, 39)]          # zz not in FASTA, b has no variants
        for bad in (dict(contig="a", contigs="b", contig_beg=None, contig_end=None),
                    dict(contig=None, contigs="a,b", contig_beg=1, contig_end=None),
                    dict(contig=None, contigs=None, contig_beg=None, contig_end=5),
                    dict(contig="nope", contigs=None, contig_beg=None, contig_end=None)):
            cfg.args = argparse.Namespace(**bad)
            with pytest.raises(SystemExit):
                V.get_vcf_regions(ref, FakeVcf())
    finally:
        cfg.args = old


# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx(tables):
    from npore_amd import aln
    c = aln.Context(*tables)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["fixture", "synthetic", "synthetic_q30"])
def test_standardize_matches_reference_realign_hap(ctx, tag, tmp_path):
    """Whole pipeline on the GPU: the standardised haplotype CIGARs are the ones the reference's compiled
    realign_hap returned; the output VCF re-read and applied again gives back the same haplotype sequences;
    standardising the output again changes nothing (idempotence)."""
    from npore_amd import standardize_vcf as S
    g, ref_seqs, vcf, regions = load_case(tag)
    prefix = str(tmp_path / "std")
    merged, hap1, hap2 = S.standardize(vcf, ref_seqs, regions, ctx, prefix, g["min_qual"])
    got = {(h[0], h[1]): h for h in hap1 + hap2}
    for want in g["haps"]:
        h = got[(want["contig"], want["hap"])]
        assert (sha(h[2]), len(h[4]), sha(h[4])) == (want["seq_sha256"], want["final_len"], want["final_sha256"])
        if "final_collapsed" in want:
            assert collapse_cigar(h[4]) == want["final_collapsed"]
    for f in ("pre1", "pre2", "1", "2", ""):
        assert os.path.getsize(f"{prefix}{f}.vcf.gz") > 28
    out = V.VcfFile(prefix + ".vcf.gz")
    assert [(r.contig, r.pos, r.alleles, r.gt) for c in out.contigs for r in out.by_contig[c]] == \
           [(r.contig, r.pos, r.alleles, r.gt) for r in merged]
    assert out.header_contigs == [h[0] for h in hap1]
    # the standardised VCF describes the same two haplotypes (except where a haplotype's CIGAR starts with an
    # insertion / deletion: gen_vcf has no anchor base for those and drops them, src/vcf.py:346, :361) ...
    whole = {c for c, _b, _e in regions if all(h[4][0] == "M" for h in hap1 + hap2 if h[0] == c)}
    assert len(whole) >= len(regions) - 1
    o1, o2 = V.split_vcf(out, regions)
    again = V.apply_vcf(o1, 1, ref_seqs, regions) + V.apply_vcf(o2, 2, ref_seqs, regions)
    assert [(a[0], a[1], a[2]) for a in again if a[0] in whole] == [(h[0], h[1], h[2]) for h in hap1 + hap2 if h[0] in whole]
    # ... and is a fixed point
    merged2, _h1, _h2 = S.standardize(out, ref_seqs, regions, ctx)
    key = lambda rs: [(r.contig, r.pos, r.alleles, r.gt) for r in rs if r.contig in whole]
    assert key(merged2) == key(merged)


@pytest.mark.gpu
def test_standardize_vcf_cli(tmp_path):
    prefix = str(tmp_path / "cli")
    res = subprocess.run([sys.executable, "-m", "npore_amd.standardize_vcf", "--vcf", os.path.join(DATA, "test_std_vcf.vcf"),
                          "--ref", os.path.join(DATA, "test_std_ref.fasta"), "--out_prefix", prefix, "--contigs", "chr19,chr18"],
                         cwd=REPO, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    out = V.VcfFile(prefix + ".vcf.gz")
    assert out.header_contigs == ["chr19", "chr18"]
    recs = [(r.contig, r.pos, r.alleles, r.gt) for c in out.contigs for r in out.by_contig[c]]
    g = load_json("std_vcf.json")["fixture"]
    assert len(recs) >= 4 and {r[0] for r in recs} == {"chr18", "chr19"}
    res = subprocess.run([sys.executable, "-m", "npore_amd.standardize_vcf", "--vcf", "/nonexistent.vcf", "--ref",
                          os.path.join(DATA, "test_std_ref.fasta"), "--out_prefix", prefix], cwd=REPO, capture_output=True, text=True)
    assert res.returncode == 1 and "ERROR: could not open VCF" in res.stdout
