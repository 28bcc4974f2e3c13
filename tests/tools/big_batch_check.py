"""One very large batch of short reads through the HIP path, every string against the oracle (scale edge: grid
sizes, scans over > 10^5 reads / chunks).  usage: big_batch_check.py [reads=120000] [ref_len=150] [r=30]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from npore_amd import aln, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
r = int(sys.argv[3]) if len(sys.argv) > 3 else 30
sub, nps, _, _ = aln.load_default_tables()
ctx = aln.Context(sub, nps)
t = time.time()
base = synth.make_batch(9, 3000, ref_len=L, p_np=0.15)
rng = np.random.default_rng(0)
pick = rng.integers(0, 3000, n)
refs = [base[0][k] for k in pick]; seqs = [base[1][k] for k in pick]; cigs = [base[2][k] for k in pick]
print(f"gen {time.time() - t:.1f}s", flush=True)
for mbr in (20000, 40):
    t = time.time()
    got, st = ctx.align_batch(refs, seqs, cigs, r=r, max_b_rows=mbr, return_status=True)
    dt = time.time() - t
    want = {}
    for k in np.unique(pick):
        want[k] = oracle.align(base[0][k], base[1][k], base[2][k], sub, nps, r=r, max_b_rows=mbr)
    bad = sum(got[i] != want[k] for i, k in enumerate(pick)) + int((st != 0).sum())
    print(f"max_b_rows={mbr}: {n} reads in {dt:.2f}s, mismatches {bad}", flush=True)
    assert bad == 0
