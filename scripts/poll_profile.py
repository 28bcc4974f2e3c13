"""Debug: failed looks of the fill kernel's step poll per role, from a -DNPORE_PROFILE_POLL build
(usage: python scripts/poll_profile.py build_exp/lib_prof.so [r=100] [reads=1000])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npore_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
from npore_amd import aln, synth
r = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
sub, nps, _, _ = aln.load_default_tables()
ctx = aln.Context(sub, nps)
refs, seqs, cigs = synth.make_batch(2, n, ref_len=10000)
for rep in range(2):
    ctx.align_batch(refs, seqs, cigs, r=r)
buf = np.zeros(32, np.int64)
rc = ctx.lib.npore_debug_fetch(ctx.handle, 7, buf.ctypes.data, 256)
assert rc == 0
d = buf[4:13].reshape(3, 3)
t = ctx.timing()
print(f"r={r} reads={n} fill {t['fill_ms']:.2f} ms")
for name, (steps, failed, lo) in zip(("first", "middle", "last"), d):
    if steps:
        print(f"  {name:6s} wave-steps {steps:>10d}  failed looks {failed:>10d} = {failed / steps:.2f} per step, {lo / max(failed, 1) * 100:.0f} % waiting for the wave below")
