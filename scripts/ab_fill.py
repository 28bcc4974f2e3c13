"""A/B of fill-kernel builds on the GPU box: for each library given, fill / prep / traceback times of C2
(1 000 x 10 kb, r=100) and of one full round at the tool's default band (4 000 x 10 kb, r=30), each build in its own
process, three repetitions, best and median printed; the strings of every build are hashed and must agree.
usage: python scripts/ab_fill.py [--cases r:reads,...] libA.so libB.so ...        (child: --child lib cases)"""
import hashlib
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child(lib, cases):
    from npore_amd import _lib
    _lib.LIB_PATH = os.path.abspath(lib)
    import ctypes
    probe = ctypes.CDLL(_lib.LIB_PATH)          # older builds lack the newest entry points: bind what is there
    for name in list(_lib.SIGNATURES):
        if not hasattr(probe, name):
            del _lib.SIGNATURES[name]
    from npore_amd import aln, synth
    import numpy as np
    sub, nps, _, _ = aln.load_default_tables()
    ctx = aln.Context(sub, nps)
    if os.environ.get("NPORE_FORCE_CHUNKS"):
        ctx.set("force_chunks", int(os.environ["NPORE_FORCE_CHUNKS"]))
    batches = {}
    for r, n in cases:
        if n not in batches:
            batches[n] = synth.make_batch(2, n, ref_len=10000)
        refs, seqs, cigs = batches[n]
        fills, preps, tbs = [], [], []
        h = None
        for rep in range(int(os.environ.get('AB_REPS', '4'))):
            out, st = ctx.align_batch(refs, seqs, cigs, r=r, return_status=True)
            t = ctx.timing()
            if rep:
                fills.append(t["fill_ms"]); preps.append(t["dev_prep_ms"]); tbs.append(t["traceback_ms"])
            h = hashlib.sha256("\n".join(out).encode()).hexdigest()[:12]
        print(f"  r={r:3d} reads={n:5d} fill best {min(fills):7.3f} med {sorted(fills)[len(fills) // 2]:7.3f}  prep {min(preps):.2f} tb {min(tbs):.2f} "
              f"bad={int((st != 0).sum())} sha={h}", flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--child":
        child(a[1], [tuple(int(x) for x in c.split(":")) for c in a[2].split(",")])
        sys.exit(0)
    cases = "100:1000,30:4000"
    if a and a[0] == "--cases":
        cases = a[1]
        a = a[2:]
    for lib in a:
        print(lib, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, cases], timeout=300)
