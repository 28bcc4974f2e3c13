#!/usr/bin/env python3
"""Golden vectors for the standardize_vcf path, from the
reference's own compiled Cython `bam.realign_hap` (src/bam.pyx:93-123).

Runs ONLY in the build container (needs /root/reference + Cython + gcc); see make_golden.py.

    python tests/golden/make_golden_vcf.py        # rewrites tests/golden/std_vcf.json and
                                                  # tests/golden/data/synth_std.{vcf,fasta}

G6 std_vcf.json: for the reference's test fixture (test/test_std_vcf.vcf + test_std_ref.fasta) and
for a seeded synthetic VCF: every haplotype tuple (contig, hap, seq, ref, cigar) that
npore_amd.vcf.apply_vcf derives (stored as sha256 of seq and of the input cigar), and the
standardised expanded CIGAR the reference's realign_hap returns for it (collapsed form for short
ones, sha256 for all).  The VCF-side functions of the reference (src/vcf.py) cannot run here
(they are written against pysam, which is absent), so they are pinned by the literal restatement in
tests/test_vcf.py instead.
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REPO, REF, build_reference, import_reference  # noqa: E402

sys.path.insert(0, REPO)


def sha(s):
    return hashlib.sha256(s.encode()).hexdigest()


def synth_vcf(seed=17):
    """Deterministic small genome + phased VCF exercising every branch of split_vcf / apply_vcf."""
    from npore_amd import synth
    rng = np.random.default_rng(seed)
    contigs, lines = [], []
    for k, (name, length) in enumerate((("ctgA", 6000), ("ctgB", 21000), ("ctgC", 900), ("ctgD", 300))):
        ref, _s, _c = synth.make_pair(seed, k, length, 0.08, 0.0, False)
        ref = "".join("NACGT"[c] for c in ref[:length])
        contigs.append((name, ref))
        if name == "ctgD":
            continue                                      # a contig without variants
        pos = int(rng.integers(1, 30))
        while pos < len(ref) - 40:
            kind = rng.integers(0, 10)
            gt = ("1|1", "0|1", "1|0", "1|0", "0|1")[int(rng.integers(0, 5))]
            qual = int(rng.integers(5, 61))
            r0 = ref[pos - 1]
            if kind < 4:                                  # SNP
                alt = "ACGT".replace(r0, "")[int(rng.integers(0, 3))]
                lines.append((name, pos, r0, alt, qual, gt))
            elif kind < 6:                                # insertion (often a copy of what follows)
                n = int(rng.integers(1, 9))
                ins = ref[pos:pos + n] if rng.random() < 0.6 else "".join(rng.choice(list("ACGT"), n))
                lines.append((name, pos, r0, r0 + ins, qual, gt))
            elif kind < 8:                                # deletion
                n = int(rng.integers(1, 9))
                lines.append((name, pos, ref[pos - 1:pos + n], r0, qual, gt))
                if rng.random() < 0.3:                    # a variant inside / at the edge of the deletion
                    p2 = pos + int(rng.integers(0, n + 1))
                    r2 = ref[p2 - 1]
                    what = rng.integers(0, 3)
                    if what == 0:
                        lines.append((name, p2, r2, r2 + "GA", qual, gt))
                    elif what == 1:
                        lines.append((name, p2, ref[p2 - 1:p2 + 2], r2, qual, gt))
                    else:
                        lines.append((name, p2, r2, "ACGT".replace(r2, "")[0], qual, gt))
            elif kind == 8:                               # two different alleles
                alts = "ACGT".replace(r0, "")
                lines.append((name, pos, r0, alts[0] + "," + r0 + "TT", qual, ("1|2", "2|1", "0|2")[int(rng.integers(0, 3))]))
            else:                                         # MNP with an unchanged first base
                lines.append((name, pos, ref[pos - 1:pos + 2], r0 + "".join("ACGT".replace(c, "")[1] for c in ref[pos:pos + 2]), qual, gt))
            pos += int(rng.integers(8, 160))
    header = ["##fileformat=VCFv4.2", '##FILTER=<ID=PASS,Description="All filters passed">'] + \
             [f"##contig=<ID={n},length={len(s)}>" for n, s in contigs] + ["##contig=<ID=ctgZ,length=10>"] + \
             ['##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
              "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE"]
    body = [f"{c}\t{p}\t.\t{r}\t{a}\t{q}\tPASS\t.\tGT\t{g}" for c, p, r, a, q, g in lines]
    fasta = "".join(f">{n}\n" + "\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + "\n" for n, s in contigs)
    return "\n".join(header + body) + "\n", fasta


def main():
    from npore_amd import cfg, vcf as V, bam as B
    data_dir = os.path.join(HERE, "data")
    vtext, ftext = synth_vcf()
    with open(os.path.join(data_dir, "synth_std.vcf"), "w") as fh:
        fh.write(vtext)
    with open(os.path.join(data_dir, "synth_std.fasta"), "w") as fh:
        fh.write(ftext)

    with tempfile.TemporaryDirectory(prefix="npore_ref_") as wd:
        src = build_reference(wd)
        rcfg, raln, rcig = import_reference(src)
        import types
        bio = sys.modules["Bio"]                        # empty import stubs, as in make_golden.py: the path
        bio.SeqIO = types.ModuleType("Bio.SeqIO")       # exercised (realign_hap) never touches pysam / Bio
        sys.modules["Bio.SeqIO"] = bio.SeqIO
        import bam as rbam
        stats = {k: np.load(f"{REF}/guppy5_stats/{k}_cm.npy") for k in ("subs", "nps", "inss", "dels")}
        rcfg.args.sub_scores, rcfg.args.np_scores, _i, _d = raln.calc_score_matrices(
            stats["subs"], stats["nps"], stats["inss"], stats["dels"])
        rcfg.counter = mp.Value("i", 0)

        out = {}
        for tag, vcf_fn, fa_fn, min_qual in (("fixture", "test_std_vcf.vcf", "test_std_ref.fasta", 0),
                                             ("synthetic", "synth_std.vcf", "synth_std.fasta", 0),
                                             ("synthetic_q30", "synth_std.vcf", "synth_std.fasta", 30)):
            ref_seqs = B.read_fasta(os.path.join(data_dir, fa_fn))
            vcf = V.VcfFile(os.path.join(data_dir, vcf_fn))
            cfg.args.contig = cfg.args.contigs = cfg.args.contig_beg = cfg.args.contig_end = None
            regions = V.get_vcf_regions(ref_seqs, vcf)
            r1, r2 = V.split_vcf(vcf, regions)
            haps = V.apply_vcf(r1, 1, ref_seqs, regions, min_qual) + V.apply_vcf(r2, 2, ref_seqs, regions, min_qual)
            recs = []
            for h in haps:
                contig, hap, seq, ref, cig = h
                res = rbam.realign_hap(h)
                final = res[4]
                rec = {"contig": contig, "hap": hap, "seq_sha256": sha(seq), "cigar_sha256": sha(cig),
                       "seq_len": len(seq), "final_sha256": sha(final), "final_len": len(final)}
                if len(final) < 2000:
                    rec["final_collapsed"] = rcig.collapse_cigar(final)
                recs.append(rec)
            out[tag] = {"vcf": vcf_fn, "fasta": fa_fn, "min_qual": min_qual, "regions": regions, "haps": recs}
            print(f"\nG6 {tag}: {len(recs)} haplotype sequences")
        with open(os.path.join(HERE, "std_vcf.json"), "w") as fh:
            json.dump(out, fh, indent=0)


if __name__ == "__main__":
    main()
