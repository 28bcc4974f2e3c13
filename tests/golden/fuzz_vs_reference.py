#!/usr/bin/env python3
"""Dev-time fuzz: oracle/ (C restatement) vs the reference's compiled Cython
align()/get_np_info(), directly, on thousands of generated cases.  Build
container only (needs /root/reference); not part of the pytest suite.

    python tests/golden/fuzz_vs_reference.py [n_cases] [seed]
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import build_reference, import_reference  # noqa: E402
import oracle  # noqa: E402
from npore_amd import synth  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    z = np.load(os.path.join(HERE, "tables.npz"))
    sub, nps = z["sub_scores"], z["np_scores"]
    with tempfile.TemporaryDirectory(prefix="npore_ref_") as wd:
        _, raln, _ = import_reference(build_reference(wd))
        bad = 0
        for k in range(n_cases):
            ref_len = int(rng.integers(1, 900))
            p_np = float(rng.choice([0.0, 0.05, 0.15, 0.4]))
            ref, seq, cig = synth.make_pair(1000 + seed, k, ref_len, p_np, float(rng.choice([0.0, 0.3, 0.9])))
            if k % 7 == 0 and len(ref) > 3:     # sprinkle N
                ref = ref.copy(); ref[rng.integers(0, len(ref), size=3)] = 0
            if k % 11 == 0 and len(seq) > 3:
                seq = seq.copy(); seq[rng.integers(0, len(seq), size=3)] = 0
            r = int(rng.choice([1, 2, 3, 5, 10, 30, 64, 100]))
            mbr = int(rng.choice([2, 3, 7, 20, 64, 500, 20000]))
            ist, iex = (5.0, 1.0) if k % 5 else (float(rng.integers(1, 8)), float(rng.integers(0, 3)))
            if len(seq) == 0 or len(ref) == 0:
                continue
            want = raln.align(ref, seq, cig.decode(), sub, nps, indel_start=ist, indel_extend=iex, max_b_rows=mbr, r=r)
            got, st = oracle.align(ref, seq, cig, sub, nps, indel_start=ist, indel_extend=iex, max_b_rows=mbr, r=r,
                                   return_status=True)
            if got != want or st:
                bad += 1
                print("MISMATCH", k, ref_len, p_np, r, mbr, ist, iex, st)
            a = np.asarray(raln.get_np_info(ref)); b = oracle.get_np_info(ref)
            if not np.array_equal(a, b):
                bad += 1
                print("NPINFO MISMATCH", k)
        print(f"fuzz: {n_cases} cases, {bad} mismatches")
        return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
