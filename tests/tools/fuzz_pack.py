"""Randomized check of the device pack (csrc/unpack_kernels.hpp) against the host pack, through the file pipeline:
`python tests/tools/fuzz_pack.py [seconds=120] [seed=0]` (GPU box).  Every round writes a BAM of random records over a few
contigs -- lengths 1 ... 3 000, soft and hard clips of both ends (odd and even lead), ambiguity codes in the read, N and
lower-complexity stretches in the reference, reads that start at position 0 or end at the contig's last base, CIGAR
operations of every kind align() takes, batches cut into several groups of reads -- and runs npore_bam_realign_file twice
on one context: align()'s inputs unpacked on the device, and packed on the host; and a third time with realign_read's glue
on the host as well (no standardize_kernel, no compaction of the texts).  The three SAM files and status arrays
must be identical (the host pack itself is pinned by the suite: test_native_realign_batch_matches_python_pipeline and
the CLI tests against the reference's golden SAM)."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from npore_amd import aln, bam


def random_record(rng, k, contigs):
    cid = int(rng.integers(len(contigs)))
    ctg = contigs[cid]
    n = int(rng.choice([1, 2, 7, 60, 300, 1200, 3000]))
    n = min(n, len(ctg))
    where = rng.random()
    pos = 0 if where < 0.15 else len(ctg) - n if where < 0.3 else int(rng.integers(0, len(ctg) - n + 1))
    ref = ctg[pos:pos + n]
    ops, seq, j = [], [], 0
    while j < n:
        e = rng.random()
        ln = int(min(n - j, rng.integers(1, 30)))
        if e < 0.55:
            ops.append((7, ln)); seq.append(ref[j:j + ln]); j += ln                       # '='
        elif e < 0.65:
            ops.append((8, ln)); seq.append("".join(rng.choice(list("ACGT"), ln))); j += ln     # 'X'
        elif e < 0.75:
            ops.append((0, ln)); seq.append(ref[j:j + ln]); j += ln                       # 'M'
        elif e < 0.88:
            ops.append((1, ln)); seq.append("".join(rng.choice(list("ACGT"), ln)))        # 'I'
        else:
            ops.append((2, ln)); j += ln                                                  # 'D'
    merged = []
    for op, ln in ops:
        if merged and merged[-1][0] == op:
            merged[-1] = (op, merged[-1][1] + ln)
        else:
            merged.append((op, ln))
    body = "".join(seq).replace("-", "A")
    if rng.random() < 0.3 and len(body) > 4:
        b = list(body)
        for p in rng.integers(0, len(b), 3):
            b[int(p)] = str(rng.choice(list("NRYKM")))
        body = "".join(b)
    lead, trail = int(rng.choice([0, 0, 1, 2, 5, 18])), int(rng.choice([0, 0, 1, 3, 10]))
    cig = ([(4, lead)] if lead else []) + merged + ([(4, trail)] if trail else [])
    if rng.random() < 0.3:
        cig = [(5, int(rng.integers(1, 50)))] + cig
    if rng.random() < 0.3:
        cig = cig + [(5, int(rng.integers(1, 50)))]
    seq = "".join(rng.choice(list("ACGT"), lead)) + body + "".join(rng.choice(list("ACGT"), trail))
    return dict(name=f"r{k}", flag=int(rng.choice([0, 16])), ref_id=cid, pos=pos, cigar=cig, seq=seq,
                qual=None if rng.random() < 0.2 else bytes(rng.integers(0, 60, len(seq), dtype=np.uint8)), hp=int(rng.integers(0, 3)))


def fuzz(budget, seed, ctx, log=print):
    """-> (rounds, reads, reads with status 0, mismatching rounds)"""
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    rounds = reads = bad = clean = 0
    with tempfile.TemporaryDirectory() as tmp:
        while time.time() < t_end:
            contigs = []
            for c in range(int(rng.integers(1, 4))):
                L = int(rng.choice([40, 500, 4000, 9000]))
                s = "".join(rng.choice(list("ACGT"), L))
                if rng.random() < 0.5:
                    p = int(rng.integers(0, max(1, L - 30)))
                    s = s[:p] + str(rng.choice(["N" * 12, "A" * 25, "AC" * 12, "n" * 5])) + s[p:]
                contigs.append(s)
            recs = [random_record(rng, k, [c.upper() for c in contigs]) for k in range(int(rng.integers(1, 60)))]
            recs.sort(key=lambda r: (r["ref_id"], r["pos"]))
            fa, bp = os.path.join(tmp, f"c{rounds}.fa"), os.path.join(tmp, f"s{rounds}.bam")
            # the FASTA as files come: one line per contig or wrapped, LF or CRLF, sometimes an empty contig in between or an
            # empty line inside one (then the library makes its own copy of the bases instead of taking them by position)
            with open(fa, "w", newline="") as fh:
                nl = "\r\n" if rng.random() < 0.25 else "\n"
                w = int(rng.choice([0, 0, 7, 60, 61, 1000]))
                for c, s in enumerate(contigs):
                    if rng.random() < 0.1:
                        fh.write(f">none{c}{nl}")
                    fh.write(f">ctg{c} some text{nl}")
                    lines = [s[i:i + w] for i in range(0, len(s), w)] if w else [s]
                    if rng.random() < 0.1 and len(lines) > 2:
                        lines.insert(int(rng.integers(1, len(lines))), "")
                    fh.write(nl.join(lines))
                    if c + 1 < len(contigs) or rng.random() < 0.8:
                        fh.write(nl)
            bam.write_bam(bp, [(f"ctg{c}", len(s)) for c, s in enumerate(contigs)], recs, level=1)
            nb, nf = bam.NativeBam(bp), bam.NativeFasta(fa)
            idx = nb.select([(f"ctg{c}", 0, len(s) - 1) for c, s in enumerate(contigs)])
            r = int(rng.choice([5, 30, 30, 100]))
            batch = int(rng.choice([1, 3, 16, 64]))
            ctx.set("tb_budget_mb", int(rng.choice([0, 0, 1])))
            outs = []
            # device pack + device glue (texts compacted on the device), host pack + device glue, host pack + host glue (op
            # strings at their slot positions, standardised on the host: no compaction)
            for dp, dg in ((1, 1), (0, 1), (0, 0)):
                ctx.set("device_pack", dp)
                ctx.set("device_glue", dg)
                out = os.path.join(tmp, f"o{rounds}_{dp}{dg}.sam")
                st = nb.realign_file(ctx, nf, idx, out, batch_reads=batch, r=r)
                outs.append((open(out, "rb").read(), st.copy()))
                os.remove(out)
            if any(o[0] != outs[0][0] or not np.array_equal(o[1], outs[0][1]) for o in outs[1:]):
                bad += 1
                log(f"MISMATCH seed={seed} round={rounds} reads={len(recs)} r={r} batch={batch}")
            nb.close(); nf.close()
            os.remove(fa); os.remove(bp)
            rounds += 1
            reads += len(idx)
            clean += int((outs[0][1] == 0).sum())
    ctx.set("device_pack", 1)
    ctx.set("device_glue", 1)
    ctx.set("tb_budget_mb", 0)
    return rounds, reads, clean, bad


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sub, nps, _, _ = aln.load_default_tables()
    ctx = aln.Context(sub, nps)
    rounds, reads, clean, bad = fuzz(budget, seed, ctx)
    ctx.close()
    print(f"fuzz_pack: {rounds} rounds, {reads} reads ({clean} with status 0), {bad} mismatches in {budget:.0f}s (seed {seed})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
