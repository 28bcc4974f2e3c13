#!/usr/bin/env python3
"""Host stages of the BAM -> SAM pipeline WITHOUT a GPU (they are what bounds the file pipeline): open (inflate + index),
record fetch + pack, standardise, SAM text -- timed one by one through the library's own entry points on a generated
BAM of 10 kb reads.  The alignment strings handed to the standardisation are the reads' true edit scripts (what
align() returns has the same alphabet and density).
    python scripts/bench_host_stages.py [--reads 4000] [--threads 8] [--reps 3]"""
import argparse
import ctypes as C
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from npore_amd import _lib, bam
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_realign


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4000)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    lib = _lib.load()
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        bp, fa, clen = bench_realign.build_inputs(tmp, a.reads, a.reads, 10000, 3)
        best = {}

        def clock(name, fn):
            t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
            best[name] = min(best.get(name, 1e9), dt)
            return r
        for rep in range(a.reps):
            nb = clock("open", lambda: bam.NativeBam(bp, stream=False, threads=a.threads, share=False))
            nf = bam.NativeFasta(fa)
            idx = nb.select([("ctg", 0, clen - 1)])
            n = len(idx)
            fmap = nb.fasta_map(nf)
            ro, so, co = (np.zeros(n + 1, np.int64) for _ in range(3))
            clock("pack_sizes", lambda: nb._check(lib.npore_bam_pack_sizes(nb.handle, idx.ctypes.data, n, ro.ctypes.data, so.ctypes.data, co.ctypes.data)))
            refs = np.zeros(int(ro[-1]) + 64, np.uint8); seqs = np.zeros(int(so[-1]) + 64, np.uint8); cigs = np.zeros(int(co[-1]) + 64, np.uint8)
            nb._check(lib.npore_bam_pack(nb.handle, nf.handle, fmap.ctypes.data, idx.ctypes.data, n, refs.ctypes.data, ro.ctypes.data,      # (first touch of the buffers: the pipeline reuses its slots)
                                         seqs.ctypes.data, so.ctypes.data, cigs.ctypes.data, co.ctypes.data, a.threads))
            clock("pack", lambda: nb._check(lib.npore_bam_pack(nb.handle, nf.handle, fmap.ctypes.data, idx.ctypes.data, n, refs.ctypes.data, ro.ctypes.data,
                                                               seqs.ctypes.data, so.ctypes.data, cigs.ctypes.data, co.ctypes.data, a.threads)))
            # standardise: the CIGAR ops as alignment strings (M -> =)
            alns = cigs.copy()
            oo = np.zeros(n + 1, np.int64); np.cumsum(2 * np.diff(co) + 16, out=oo[1:])
            out = np.empty(int(oo[-1]) + 1, np.uint8); olen = np.zeros(n, np.int64)
            out[:] = 0
            clock("standardize", lambda: nb._check(lib.npore_standardize_batch(n, alns.ctypes.data, co.ctypes.data, refs.ctypes.data, ro.ctypes.data, seqs.ctypes.data,
                                                                                so.ctypes.data, out.ctypes.data, oo.ctypes.data, olen.ctypes.data, a.threads)))
            st = np.zeros(n, np.int32)
            sam, sam_len = C.c_void_p(), C.c_int64()
            clock("format_sam", lambda: nb._check(lib.npore_bam_format_sam(nb.handle, idx.ctypes.data, n, out.ctypes.data, oo.ctypes.data, olen.ctypes.data,
                                                                           st.ctypes.data, a.threads, C.byref(sam), C.byref(sam_len))))
            with open(os.path.join(tmp, "o.sam"), "wb") as fh:
                clock("write", lambda: fh.write(C.string_at(sam.value, sam_len.value)))
            nb.close(); nf.close()
        print(f"{n} reads, {a.threads} threads, best of {a.reps} (ms):", {k: round(v * 1e3, 1) for k, v in best.items()},
              "sam MB", round(sam_len.value / 1e6, 1))


if __name__ == "__main__":
    main()
