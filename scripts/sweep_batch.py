"""Batch-size sweep on the GPU box: reads/s (device time of the whole path: prep + fill + traceback/gather) and
fill time for batches of 2 000 ... 12 000 reads of 10 kb at band half-width r (default 30, the tool's default), one
library build per child process.  The persistent chunk queue of the fill kernel should make the curve follow
ceil(chunks / resident chunks) without the extra cliffs of static dealing.
usage: python scripts/sweep_batch.py [--r 30] [--lo 2000 --hi 12000 --step 500] [--out file.csv] lib.so [lib2.so ...]"""
import argparse
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child(lib, r, lo, hi, step):
    from npore_amd import _lib
    _lib.LIB_PATH = os.path.abspath(lib)
    import ctypes
    probe = ctypes.CDLL(_lib.LIB_PATH)
    for name in list(_lib.SIGNATURES):
        if not hasattr(probe, name):
            del _lib.SIGNATURES[name]
    import multiprocessing as mp
    from npore_amd import synth
    span = 250
    with mp.get_context("fork").Pool(16) as pool:       # before anything touches HIP
        parts = pool.map(synth.make_span, [(2, span, 10_000, False, k, 1) for k in range(0, hi, span)])
    refs = [x for p in parts for x in p[0]]; seqs = [x for p in parts for x in p[1]]; cigs = [x for p in parts for x in p[2]]
    from npore_amd import aln
    sub, nps, _, _ = aln.load_default_tables()
    ctx = aln.Context(sub, nps)
    for n in range(lo, hi + 1, step):
        best = None
        for rep in range(3):
            out, st = ctx.align_batch(refs[:n], seqs[:n], cigs[:n], r=r, return_status=True)
            t = ctx.timing()
            dev = t["fill_ms"] + t["dev_prep_ms"] + t["traceback_ms"]
            if rep and (best is None or dev < best[0]):
                best = (dev, t["fill_ms"], t["dev_prep_ms"], t["traceback_ms"])
        print(f"{os.path.basename(lib)},{r},{n},{best[0]:.3f},{best[1]:.3f},{best[2]:.3f},{best[3]:.3f},{n / best[0] * 1e3:.0f},{n / best[1] * 1e3:.0f},{int((st != 0).sum())}",
              flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--r", type=int, default=30)
    ap.add_argument("--lo", type=int, default=2000)
    ap.add_argument("--hi", type=int, default=12000)
    ap.add_argument("--step", type=int, default=500)
    ap.add_argument("--child", default=None)
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    if a.child:
        child(a.child, a.r, a.lo, a.hi, a.step)
        sys.exit(0)
    print("lib,r,reads,device_ms,fill_ms,prep_ms,traceback_ms,reads_per_s_device,reads_per_s_fill,bad", flush=True)
    for lib in a.libs:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, "--r", str(a.r), "--lo", str(a.lo),
                        "--hi", str(a.hi), "--step", str(a.step)], timeout=900)
