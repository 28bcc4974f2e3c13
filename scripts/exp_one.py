"""One configuration through the library, synchronous calls: python scripts/exp_one.py <r> <reads> [reps] [ref_len]
(the program to put behind `rocprofv3 ... --`; NPORE_AMD_LIB selects another build).  Prints the stage times of the last call."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npore_amd import aln, synth
r, n = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ref_len = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
sub, nps, _, _ = aln.load_default_tables()
ctx = aln.Context(sub, nps)
refs, seqs, cigs = synth.make_batch(2, n, ref_len=ref_len)
for rep in range(reps):
    out, st = ctx.align_batch(refs, seqs, cigs, r=r, return_status=True)
t = ctx.timing()
print("fill", round(t["fill_ms"], 2), "prep", round(t["dev_prep_ms"], 2), "tb", round(t["traceback_ms"], 2), "bad", int((st != 0).sum()))
