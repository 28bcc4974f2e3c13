"""Time the n-polymer region batch (npore_np_regions, the bed.py path) on a synthetic genome.
usage: bench_bed.py [mbases=256] [chunk_width=1000000]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from npore_amd import aln

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cw = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
rng = np.random.default_rng(0)
g = rng.integers(1, 5, mb * 1_000_000).astype(np.uint8)
# low-complexity stretches and assembly gaps
for _ in range(mb * 40):
    p = int(rng.integers(0, len(g) - 4000)); n = int(rng.integers(1, 7)); l = int(rng.integers(3, 40))
    g[p:p + n * l] = np.tile(g[p:p + n], l)
for _ in range(max(1, mb // 16)):
    p = int(rng.integers(0, len(g) - 600_000)); g[p:p + int(rng.integers(10_000, 500_000))] = 0
slices = [g[i:i + cw] for i in range(0, len(g), cw)]
ctx = aln.Context(np.zeros((5, 5), np.float32), np.zeros((6, 101, 101), np.float32))
for rep in range(2):
    t = time.time()
    res = ctx.np_regions(slices)
    dt = time.time() - t
    print(f"rep {rep}: {len(slices)} slices of {cw} bases: {dt:.2f}s = {len(g) / dt / 1e6:.0f} Mbases/s; starts per period: "
          f"{[int(sum(len(p) for p, _ in res[n])) for n in range(6)]}", flush=True)
