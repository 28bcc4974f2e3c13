"""Device glue vs op strings for one batch: python scripts/exp_glue.py <r> <reads> [reps]   (behind rocprofv3 --kernel-trace --stats
for the per-kernel times: standardize_kernel vs gather_kernel)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npore_amd import aln, synth
r, n = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sub, nps, _, _ = aln.load_default_tables()
ctx = aln.Context(sub, nps)
refs, seqs, cigs = synth.make_batch(2, n)
for fin in (False, True):
    for rep in range(reps):
        t0 = time.time()
        out, st = ctx.align_batch(refs, seqs, cigs, r=r, return_status=True, final_cigars=fin)
        dt = time.time() - t0
    t = ctx.timing()
    print("final" if fin else "ops  ", "wall", round(dt * 1e3, 1), "fill", round(t["fill_ms"], 2), "tb+post", round(t["traceback_ms"], 2), "bad", int((st != 0).sum()),
          "bytes", sum(len(x) for x in out))
