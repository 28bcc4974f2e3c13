#!/usr/bin/env python3
"""Per-read digests of the FULL-SIZE configurations, from the pinned oracle.

The oracle (oracle/npore_oracle.c) is pinned against the reference's own compiled
Cython align() by make_golden.py; this script runs it over every read of the
bench configurations in the build container, where CPU time
is free, and stores (len, sha256[:16]) per read, so that the GPU tests compare
EVERY read of every full-size configuration bit for bit at no cost on the GPU box:

  c2     8 000 reads, seed 2, 10 kb, r=100                      (BASELINE configs[1]; indices 0 ... 7 999 are what
                                                                 the ranks of `bench.py --gpus 8` hold: rank k has k, k + 8, ...)
  r30    first 4 000 reads + every 7th up to 32 000, seed 2, r=30   (the tool's default band; the `production_default` leg of
                                                                 8 ranks x 4 000 reads; 7 is coprime to 1 / 2 / 4 / 8 ranks, so
                                                                 every rank of every world size holds sampled reads)
  c5       256 reads, seed 5, 50 kb, r=200                      (BASELINE configs[4])
  c3    first 10 000 reads + every 10th after, seed 3 mixed     (BASELINE configs[2])
  c4    first 8 000 reads + every 101st up to 1 000 000, seed 4 mixed, r=100   (BASELINE configs[3]; 101 is prime: every
                                                                 rank of a round-robin deal holds sampled reads)

    python tests/golden/make_fullsize_digests.py [--procs 8] [--only c2,r30]

Output: tests/golden/fullsize_digests.npz (<name>_idx int32, <name>_len int32,
<name>_dig uint64 = first 16 hex digits of sha256 of the raw align() string).
All at max_b_rows=20000, indel_start=5, indel_extend=1, max_n=6, max_l=100, tables G1.
"""
import argparse
import hashlib
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

CONFIGS = {
    # name: (base_seed, ref_len, mixed, r, indices)
    "c2": (2, 10_000, False, 100, np.arange(8000)),
    "r30": (2, 10_000, False, 30, np.concatenate([np.arange(4000), np.arange(4000, 32_000, 7)])),
    "c5": (5, 50_000, False, 200, np.arange(256)),
    "c3": (3, 10_000, True, 100, np.concatenate([np.arange(10_000), np.arange(10_000, 100_000, 10)])),
    "c4": (4, 10_000, True, 100, np.concatenate([np.arange(8000), np.arange(8000, 1_000_000, 101)])),
}


def digest(s):
    return int(hashlib.sha256(s.encode()).hexdigest()[:16], 16)


def _work(job):
    import oracle
    from npore_amd import synth
    seed, ref_len, mixed, r, idx = job
    z = np.load(os.path.join(HERE, "tables.npz"))
    sub, nps = z["sub_scores"], z["np_scores"]
    lens, digs = [], []
    for i in idx:
        ref, seq, cig = synth.make_pair(seed, int(i), ref_len, mixed=mixed)
        s, st = oracle.align(ref, seq, cig, sub, nps, r=r, return_status=True)
        assert st == 0, (seed, i, st)
        lens.append(len(s)); digs.append(digest(s))
    return lens, digs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=os.cpu_count())
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    out_path = os.path.join(HERE, "fullsize_digests.npz")
    res = dict(np.load(out_path)) if os.path.exists(out_path) else {}
    import oracle
    oracle.build()
    names = [n for n in CONFIGS if not a.only or n in a.only.split(",")]
    with mp.get_context("fork").Pool(a.procs) as pool:
        for name in names:
            seed, ref_len, mixed, r, idx = CONFIGS[name]
            t0 = time.time()
            # reads the file already holds are kept (a digest is a function of (seed, index, r) alone)
            have = {}
            if name + "_idx" in res:
                have = {int(i): (int(l), int(d)) for i, l, d in zip(res[name + "_idx"], res[name + "_len"], res[name + "_dig"])}
            todo = np.array([i for i in idx if int(i) not in have], np.int64)
            span = 8 if ref_len > 20_000 else 50
            jobs = [(seed, ref_len, mixed, r, todo[k:k + span]) for k in range(0, len(todo), span)]
            parts = pool.map(_work, jobs, chunksize=1)
            for i, l, d in zip(todo, (x for p in parts for x in p[0]), (x for p in parts for x in p[1])):
                have[int(i)] = (l, d)
            res[name + "_idx"] = idx.astype(np.int32)
            res[name + "_len"] = np.array([have[int(i)][0] for i in idx], np.int32)
            res[name + "_dig"] = np.array([have[int(i)][1] for i in idx], np.uint64)
            np.savez_compressed(out_path, **res)
            print(f"{name}: {len(todo)} new of {len(idx)} reads in {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
