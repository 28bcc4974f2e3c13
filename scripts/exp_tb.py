"""Traceback kernels side by side: python scripts/exp_tb.py <r> <reads> <mode> [reps]   (mode 1 = windows, 2 = rows; put behind
rocprofv3 --kernel-trace --stats for the per-kernel times)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npore_amd import aln, synth
r, n, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
sub, nps, _, _ = aln.load_default_tables()
ctx = aln.Context(sub, nps)
ctx.set("traceback_kernel", mode)
refs, seqs, cigs = synth.make_batch(2, n)
for rep in range(reps):
    out, st = ctx.align_batch(refs, seqs, cigs, r=r, return_status=True)
t = ctx.timing()
if os.environ.get("TB_COUNT_RUNS"):
    import numpy as np
    lut = np.zeros(256, np.uint8); lut[ord("I")] = 1; lut[ord("D")] = 2
    hops = 0
    for o in out:
        c = lut[np.frombuffer(o.encode() if isinstance(o, str) else o, np.uint8)]
        hops += 1 + int(np.count_nonzero(c[1:] != c[:-1])) if len(c) else 0
    print("runs (=/X as one class) per read:", round(hops / len(out), 1))
print("mode", mode, "fill", round(t["fill_ms"], 2), "prep", round(t["dev_prep_ms"], 2), "tb", round(t["traceback_ms"], 2), "bad", int((st != 0).sum()))
