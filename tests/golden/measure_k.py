#!/usr/bin/env python3
"""k = (reference Cython align() reads/s) / (oracle/npore_oracle.c reads/s), same reads, same core.

Build container only (needs /root/reference + Cython + gcc): the reference cannot travel to the GPU
box, so bench.py times the oracle (a plain-C port) on the box's host cores and multiplies by this k to
quote a "Cython-equivalent" CPU rate (SURVEY.md section 8(d) "CPU baseline").
Both are run single-process on the first reads of config C2 (npore_amd.synth, base_seed=2, 10 kb) at
r=30 (the tool's default) and r=100 (C2); outputs are compared string by string.

    python tests/golden/measure_k.py [n_reads]        # rewrites tests/golden/k_cython_over_port.json
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import build_reference, import_reference  # noqa: E402
import oracle  # noqa: E402
from npore_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    z = np.load(os.path.join(HERE, "tables.npz"))
    sub, nps = z["sub_scores"], z["np_scores"]
    refs, seqs, cigs = synth.make_batch(2, n, ref_len=10_000)
    oracle.build()
    out = {"reads": n, "ref_len": 10_000, "generator": "npore_amd.synth base_seed=2 (config C2)",
           "host": f"build container, {os.cpu_count()} cpus, one process", "by_r": {}}
    with tempfile.TemporaryDirectory(prefix="npore_ref_") as wd:
        _, raln, _ = import_reference(build_reference(wd))
        for r in (30, 100):
            best_c = best_p = 1e30
            for _ in range(2):            # best of two: the container's cores are shared
                t0 = time.perf_counter()
                want = [raln.align(refs[k], seqs[k], cigs[k].decode(), sub, nps, r=r) for k in range(n)]
                best_c = min(best_c, time.perf_counter() - t0)
                t0 = time.perf_counter()
                got, _ = oracle.align_batch(refs, seqs, cigs, sub, nps, r=r)
                best_p = min(best_p, time.perf_counter() - t0)
                assert got == want, f"oracle != Cython at r={r}"
            out["by_r"][str(r)] = {"cython_reads_per_s": round(n / best_c, 3), "port_reads_per_s": round(n / best_p, 3),
                                   "k": round(best_p / best_c, 4)}
            print(r, out["by_r"][str(r)], flush=True)
    with open(os.path.join(HERE, "k_cython_over_port.json"), "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    main()
