"""standardize_vcf on a chromosome-scale synthetic contig: timing of the stages and the two size-independent
properties -- the standardised VCF describes the same haplotype sequences, and standardising it again changes
nothing (except where an insertion and a deletion that would cancel sit on either side of a chunk border of the first
run: chunks are aligned independently between end points fixed by the input path, src/aln.pyx:445-456, and the second
run's borders fall elsewhere; 60 Mbp: 1 such site in 60 780 records).  usage: check_vcf_scale.py [mbases=5] [variants_per_kb=1.0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from npore_amd import aln, synth, vcf as V, standardize_vcf as S

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
dens = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
L = int(mb * 1e6)
rng = np.random.default_rng(4)
base = synth.make_batch(2, 64, ref_len=10_000)[0]
ref = "".join("NACGT"[c] for c in np.concatenate([base[k] for k in rng.integers(0, 64, (L + 9999) // 10000)])[:L])
recs = []
pos = 50
while pos < L - 100:
    r0 = ref[pos - 1]
    kind = int(rng.integers(0, 4))
    if kind < 2:
        alleles = (r0, "ACGT".replace(r0, "")[int(rng.integers(0, 3))])
    elif kind == 2:
        n = int(rng.integers(1, 7)); alleles = (r0, r0 + (ref[pos:pos + n] if rng.random() < 0.7 else "".join(rng.choice(list("ACGT"), n))))
    else:
        alleles = (ref[pos - 1:pos + int(rng.integers(1, 7))], r0)
    recs.append(V.VcfRecord("chrS", pos, alleles, 60.0, gt=((1, 1), (1, 0), (0, 1))[int(rng.integers(0, 3))]))
    pos += int(rng.integers(20, max(21, int(2000 / dens))))


class Mem:                        # a VcfFile-like holder of in-memory records
    header, samples = ["##fileformat=VCFv4.2", f"##contig=<ID=chrS,length={L}>"], ["SAMPLE"]
    def __init__(self, r): self.by_contig = {"chrS": r}
    contigs = ["chrS"]; header_contigs = ["chrS"]
    def fetch(self, c, s, e): return [r for r in self.by_contig.get(c, ()) if r.start < e and r.stop > s]

sub, nps, _, _ = aln.load_default_tables()
ctx = aln.Context(sub, nps)
regions = [("chrS", 0, L - 1)]
refs = {"chrS": ref}
t = time.time()
merged, h1, h2 = S.standardize(Mem(recs), refs, regions, ctx)
t1 = time.time() - t
print(f"{L} bases, {len(recs)} input variants -> {len(merged)} standardised in {t1:.2f}s", flush=True)
o1, o2 = V.split_vcf(Mem(merged), regions)
again = V.apply_vcf(o1, 1, refs, regions) + V.apply_vcf(o2, 2, refs, regions)
same_seq = [a[2] == h[2] for a, h in zip(again, h1 + h2)]
t = time.time()
merged2, _a, _b = S.standardize(Mem(merged), refs, regions, ctx)
key = lambda rs: [(r.pos, r.alleles, r.gt) for r in rs]
print(f"same haplotype sequences: {same_seq}; idempotent: {key(merged2) == key(merged)} ({time.time() - t:.2f}s)")
if key(merged2) != key(merged):
    a, b = key(merged), key(merged2)
    diff = [k for k in range(min(len(a), len(b))) if a[k] != b[k]][:5]
    print("first differences:", [(a[k], b[k]) for k in diff], len(a), len(b))
assert all(same_seq)
