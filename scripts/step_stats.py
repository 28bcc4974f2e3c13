"""Debug: how often each part of the cell update runs, per wave-step (library built with _lib.build(defines=("NPORE_STATS",), out=...): -DNPORE_EXPERIMENTS -DNPORE_STATS).
usage: python scripts/step_stats.py build_exp/lib_stats.so [r:reads ...]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from npore_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from npore_amd import aln, synth

NAMES = ["wave-steps", "any candidate", "any SHR", "SHR small", "SHR small TWO", "LEN filter iterations", "LEN candidate passes", "single, all n=1", "single, all n<=2"]
sub, nps, _, _ = aln.load_default_tables()
ctx = aln.Context(sub, nps)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
for case in sys.argv[2:] or ["100:1000", "30:1000"]:
    r, n = (int(x) for x in case.split(":"))
    refs, seqs, cigs = synth.make_batch(2, n, ref_len=10000)
    lib.npore_debug_stats(buf, 1)
    ctx.align_batch(refs, seqs, cigs, r=r)
    lib.npore_debug_stats(buf, 1)
    print(f"r={r} reads={n}")
    for k, name in enumerate(NAMES):
        print(f"  {name:24s} {buf[k]:12d}  {buf[k] / max(1, buf[0]):.3f} per wave-step")
