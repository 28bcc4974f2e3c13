import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from npore_amd import aln, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [100, 30]
ref_len = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
sub, nps, _, _ = aln.load_default_tables()
ctx = aln.Context(sub, nps)
for k in ('force_chunks',):
    if os.environ.get('NPORE_'+k.upper()): ctx.set(k, int(os.environ['NPORE_'+k.upper()]))
t = time.time(); refs, seqs, cigs = synth.make_batch(2, n, ref_len=ref_len); print("gen", time.time() - t, flush=True)
for r in rs:
    for rep in range(3):
        t = time.time()
        out, st = ctx.align_batch(refs, seqs, cigs, r=r, return_status=True)
        dt = time.time() - t
        tm = ctx.timing()
        print(f"r={r} rep={rep} wall={dt*1e3:.1f}ms reads/s={n/dt:.0f} fill={tm['fill_ms']:.1f} tb={tm['traceback_ms']:.1f} "
              f"h2d={tm['h2d_ms']:.1f} d2h={tm['d2h_ms']:.1f} prep={tm['dev_prep_ms']:.1f} cells={tm['cells']:.3g} "
              f"Gcells/s(fill)={tm['cells']/tm['fill_ms']/1e6:.2f} bad={int((st!=0).sum())}", flush=True)
