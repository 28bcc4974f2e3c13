#!/usr/bin/env python3
"""What a cold process pays before its first batch is on the GPU: context creation, the first align call (the code object
is loaded at the first launch), the second call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from npore_amd import aln, synth

sub, nps, _, _ = aln.load_default_tables()
refs, seqs, cigs = synth.make_batch(9, 8, ref_len=1500)
t0 = time.perf_counter()
ctx = aln.Context(sub, nps, device=0)
t1 = time.perf_counter()
print(f"context      {1e3 * (t1 - t0):8.1f} ms")
for k in range(3):
    t1 = time.perf_counter()
    ctx.align_batch(refs, seqs, cigs, r=30, max_b_rows=20000)
    t2 = time.perf_counter()
    print(f"align call {k} {1e3 * (t2 - t1):8.1f} ms")
ctx.close()
