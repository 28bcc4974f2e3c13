#!/usr/bin/env python3
"""Golden vectors for calc_confusion_matrices, from the reference's own
compiled Cython `bam.calc_confusion_matrices` (src/bam.pyx:351-499).

Runs ONLY in the build container (needs /root/reference + Cython + gcc); see make_golden.py.

    python tests/golden/make_golden_cms.py        # rewrites tests/golden/cms.json

The reference reads its pileups from `samtools mpileup ... | cut -f5` (src/bam.pyx:301-316); samtools is absent here,
so the INPUT lines of the fixture are made by the small pileup writer below (mpileup's column-5 syntax: bases, '^' +
mapping quality, '$', '*', '+nSEQ', '-nSEQ', lower case on the reverse strand) from the reference's own test reads
(test/data/reads.sam on ref.fasta), plus hand-written lines that hit the branches those reads do not.  The module-level
`get_pileups` of the compiled module is replaced by one that serves these lines; everything downstream (get_np_info,
the character loop, the matrices) is the reference's compiled code.
"""
import argparse
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


def pileup_lines(sam_path, ctg_len):
    """mpileup column 5 for every position of the contig from the SAM records (no base-quality filter, no -f)."""
    import re
    cols = [[] for _ in range(ctg_len)]
    for line in open(sam_path):
        if line.startswith("@"):
            continue
        f = line.rstrip("\n").split("\t")
        flag, pos, mapq, cigar, seq = int(f[1]), int(f[3]) - 1, int(f[4]), f[5], f[9]
        if flag & (4 | 256 | 2048):
            continue
        rev = bool(flag & 16)
        case = (lambda x: x.lower()) if rev else (lambda x: x.upper())
        ops = [(int(n), op) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", cigar)]
        q, r = 0, pos
        entries = []          # (ref position, text)
        for k, (n, op) in enumerate(ops):
            if op in "M=X":
                for t in range(n):
                    entries.append([r, case(seq[q])])
                    q += 1; r += 1
            elif op == "I":
                if entries:
                    entries[-1][1] += f"+{n}{case(seq[q:q + n])}"
                q += n
            elif op == "D":
                if entries:
                    entries[-1][1] += f"-{n}{case('N' * n)}"
                for t in range(n):
                    entries.append([r, "*"])
                    r += 1
            elif op == "S":
                q += n
        if not entries:
            continue
        entries[0][1] = "^" + chr(min(mapq, 93) + 33) + entries[0][1]
        entries[-1][1] += "$"
        for rp, text in entries:
            if 0 <= rp < ctg_len:
                cols[rp].append(text)
    return ["".join(c) for c in cols]


def main():
    import types
    with tempfile.TemporaryDirectory(prefix="npore_ref_") as wd:
        rcfg, raln, rcig = import_reference(build_reference(wd))
        bio = sys.modules["Bio"]                # empty import stubs, as in make_golden.py / make_golden_vcf.py: the
        bio.SeqIO = types.ModuleType("Bio.SeqIO")   # path exercised (calc_confusion_matrices) never touches pysam / Bio
        sys.modules["Bio.SeqIO"] = bio.SeqIO
        import bam as rbam                      # the compiled module
        ref_name, ref_seq = None, []
        for line in open(f"{REF}/test/data/ref.fasta"):
            if line.startswith(">"):
                ref_name = line[1:].split()[0]
            else:
                ref_seq.append(line.strip().upper())
        ref_seq = "".join(ref_seq)
        lines = pileup_lines(f"{REF}/test/data/reads.sam", len(ref_seq))
        # a second contig with engineered n-polymers and lines that exercise every branch
        ctg2 = "ACGT" + "A" * 7 + "CG" * 5 + "TTAGGG" * 4 + "ACGTAC" + "T" * 120 + "GATTACA"
        eng = [""] * len(ctg2)
        eng[0] = "^~A^!a^]C"                                   # read starts with odd mapq characters
        eng[3] = "T+1A t-7AAAAAAA T+2AA T-2AA T-3AAA".replace(" ", "")     # homopolymer (starts at 4): ins / del of copies and others
        eng[4] = "A*a$A$"
        eng[10] = "A+2CG A+4CGCG A-2CG A-10CGCGCGCGCG A+2GC A+3CGC".replace(" ", "")   # dinucleotide repeat starts at 11
        eng[20] = "G+6TTAGGG G-6TTAGGG G-12TTAGGGTTAGGG G+6TTAGGA g+12ttagggttaggg".replace(" ", "")
        eng[50] = "T+1T T-1T T+150" + "T" * 150 + "T-150" + "T" * 150          # inside the long homopolymer, lengths beyond max_l
        eng[51] = "N n * $"
        eng[52] = "A+12ACGTACGTACGT C-3NNN"
        eng[53] = "AC?GT"                                      # unexpected character: the rest of the line is dropped
        eng[len(ctg2) - 1] = "A+2GG$"
        eng = [e.replace(" ", "") for e in eng]
        for k in range(len(eng)):
            if not eng[k]:
                eng[k] = "ACgt"[k % 4] * (1 + k % 3)
        cases = []
        rcfg.counter = mp.Value("i", 0)
        for name, seq, all_lines, ranges in ((ref_name, ref_seq, lines, [(0, len(ref_seq) - 1), (100, 400), (0, 1)]),
                                            ("eng", ctg2, eng, [(0, len(ctg2) - 1), (3, 60), (40, len(ctg2) - 1)])):
            for start, end in ranges:
                sub = all_lines[start:end]
                rcfg.args = argparse.Namespace(max_n=6, max_l=100, stats_dir=f"{REF}/guppy5_stats", recalc_cms=False,
                                               out_prefix=os.path.join(wd, "out"), bam="unused.bam", refs={name: seq},
                                               regions=[(name, start, end)], chunk_width=100000)
                rbam.get_pileups = lambda bam, ctg, s, e, _l=sub: iter([x.upper().strip() for x in _l])
                subs, nps, inss, dels = rbam.calc_confusion_matrices((name, start, end))
                nz = np.argwhere(np.asarray(nps))
                cases.append({"contig": name, "seq": seq, "start": start, "end": end, "lines": sub,
                              "subs": np.asarray(subs).tolist(), "inss": np.asarray(inss).tolist(),
                              "dels": np.asarray(dels).tolist(),
                              "nps_nonzero": [[int(a), int(b), int(c), int(np.asarray(nps)[a, b, c])] for a, b, c in nz]})
                print(name, start, end, "subs", int(np.asarray(subs).sum()), "nps", int(np.asarray(nps).sum()),
                      "inss", int(np.asarray(inss).sum()), "dels", int(np.asarray(dels).sum()))
        with open(os.path.join(HERE, "cms.json"), "w") as fh:
            json.dump({"max_n": 6, "max_l": 100, "cases": cases}, fh)
            fh.write("\n")


if __name__ == "__main__":
    main()
