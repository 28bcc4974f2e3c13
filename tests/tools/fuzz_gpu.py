"""Long randomized parity run of the HIP path against the oracle.  `python tests/tools/fuzz_gpu.py [seconds=300] [seed=0]`
runs for minutes; tests/test_gpu_parity.py::test_fuzz_time_boxed runs a time-boxed leg of it (fixed seeds) inside the
driver-run GPU suite, half of it FOCUSED on the two regions that have failed before (tiny max_b_rows with 4-8 waves per
chunk; max_l < 32).
Every round draws a band half-width, chunk height, gap penalties, a score-table variant (the shipped tables, tables
with random entries incl. ties/negatives, other max_n / max_l) and a batch of short-to-medium reads (random n-polymer
density, N bases, input paths that hug the band edge) and one of the two traceback kernels, and compares every string
and status with the oracle's."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from npore_amd import aln, synth, cig as cig_mod

_tables0 = None


def tables(rng, kind, max_n, max_l):
    global _tables0
    if _tables0 is None:
        _tables0 = aln.load_default_tables()[:2]
    sub0, nps0 = _tables0
    if kind == 0 and (max_n, max_l) == (6, 100):
        return sub0, nps0
    r2 = np.random.default_rng(int(rng.integers(1 << 30)))
    sub = sub0.copy()
    nps = nps0[:max_n, :min(max_l, 100) + 1, :min(max_l, 100) + 1]
    if max_l > 100:
        nps = np.pad(nps, ((0, 0), (0, max_l - 100), (0, max_l - 100)), mode="edge")
    nps = np.ascontiguousarray(nps).copy()
    if kind == 1:       # random but structured like the real ones (a few exact ties, some negatives)
        nps = (r2.integers(-2, 40, nps.shape) / 4.0).astype(np.float32)
        nps[:, :3, :] = 20.0
        sub = (r2.integers(0, 24, (5, 5)) / 4.0).astype(np.float32)
        sub[0, :] = 0; sub[:, 0] = 0
    elif kind == 2:     # everything equal: maximal tie pressure on the strict-< rules
        nps[:] = 1.0
        sub[:] = 1.0
    return sub, nps


def random_script_pair(r2, L, r):
    """A pair from a random edit script with long indel runs (up to ~3r: the band follows the input path, so the new
    path can run along the band edges), low-complexity reference."""
    alpha = r2.integers(1, 5, size=int(r2.integers(1, 5)))
    ref = r2.choice(alpha, size=L).astype(np.uint8)
    ops = []
    j = 0
    seq = []
    while j < L:
        kind = r2.integers(0, 10)
        if kind < 6:
            n = int(min(L - j, r2.integers(1, 40)))
            ops.append(b"=" * n); seq.append(ref[j:j + n]); j += n
        elif kind == 6:
            n = int(min(L - j, r2.integers(1, 4)))
            ops.append(b"X" * n); seq.append((ref[j:j + n] % 4 + 1).astype(np.uint8)); j += n
        elif kind == 7:
            n = int(r2.integers(1, max(2, 3 * r)))
            ops.append(b"I" * n); seq.append(r2.choice(alpha, size=n).astype(np.uint8))
        elif kind == 8:
            n = int(min(L - j, r2.integers(1, max(2, 3 * r))))
            ops.append(b"D" * n); j += n
        else:
            n = int(min(L - j, r2.integers(1, 30)))
            ops.append(b"M" * n); seq.append(ref[j:j + n]); j += n
    return ref, (np.concatenate(seq) if seq else np.zeros(0, np.uint8)), b"".join(ops)


def fuzz(budget, seed, focus=False, log=print):
    """Rounds until `budget` seconds are used up; returns (rounds, reads, mismatches).  focus=True draws only the
    shapes that have failed before: 4-8 waves per chunk with tiny chunk heights, tables with max_l < 32."""
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    rounds = reads = bad = 0
    while time.time() < t_end:
        max_n = int(rng.choice([6, 6, 6, 4, 1]))
        max_l = int(rng.choice([5, 12, 20, 31] if focus else [100, 100, 100, 20, 127]))
        kind = int(rng.choice([0, 0, 1, 2]))
        sub, nps = tables(rng, kind, max_n, max_l)
        ctx = aln.Context(sub, nps, max_n=max_n, max_l=max_l)
        r = int(rng.choice([100, 127, 128, 160, 192, 200, 255, 256, 320, 511] if focus else
                           [1, 2, 3, 7, 15, 30, 31, 32, 33, 64, 65, 100, 127, 128, 160, 192, 200, 255, 288, 448]))
        mbr = int(rng.choice([2, 3, 5, 7, 16, 64] if focus else [2, 3, 5, 16, 64, 65, 200, 1000, 20000, 60000]))
        if r > 255:
            mbr = min(mbr, 1000)          # (the oracle allocates and zeroes 60 B x max_b_rows x (2r+1) per read)
        ist, iex = (float(x) for x in rng.choice([[5, 1], [5, 1], [3, 0], [0, 0], [7.5, 2.25], [1, 1]]))
        n = int(rng.integers(1, 40))
        refs, seqs, cigs = [], [], []
        for k in range(n):
            L = int(rng.choice([0, 1, 2, 5, 40, 150, 700, 2500]))
            if L == 0:
                ref, seq, cig = np.zeros(0, np.uint8), np.zeros(0, np.uint8), b""
            elif rng.random() < 0.35:
                ref, seq, cig = random_script_pair(np.random.default_rng(int(rng.integers(1 << 30))), L, min(r, 60))
            else:
                ref, seq, cig = synth.make_pair(int(rng.integers(1 << 30)), k, L, float(rng.choice([0.0, 0.05, 0.2, 0.6])),
                                                float(rng.choice([0.0, 0.3, 1.0])))
            ref, seq = np.array(ref, np.uint8), np.array(seq, np.uint8)
            if rng.random() < 0.3 and len(ref) > 3:
                ref[rng.integers(0, len(ref), size=3)] = 0
            if rng.random() < 0.3 and len(seq) > 3:
                seq[rng.integers(0, len(seq), size=3)] = 0
            if rng.random() < 0.1 and len(ref) > 30:        # long homopolymer / satellite in both
                p = int(rng.integers(0, len(ref) - 20)); unit = ref[p:p + int(rng.integers(1, 7))]
                rep = np.tile(unit, 150)
                ref = np.concatenate([ref[:p], rep, ref[p:]])
                # keep the pair consistent: insert the same repeat into the read at the matching path position
                cg = np.frombuffer(cig, np.uint8)
                consumed_ref = np.cumsum(cg != ord("I"))
                at = int(np.searchsorted(consumed_ref, p, side="right")) if p > 0 else 0
                sp = int(np.sum(cg[:at] != ord("D")))
                seq = np.concatenate([seq[:sp], rep, seq[sp:]])
                cig = bytes(cg[:at]) + b"=" * len(rep) + bytes(cg[at:])
            refs.append(ref); seqs.append(seq); cigs.append(cig)
        got, st = ctx.align_batch(refs, seqs, cigs, indel_start=ist, indel_extend=iex, r=r, max_b_rows=mbr, return_status=True)
        # every fourth round also through the device glue (standardize_kernel): its text must be what the host glue makes
        # of the op strings just returned (the strings themselves are compared with the oracle below)
        if rounds % 4 == 0:
            fin, st2 = ctx.align_batch(refs, seqs, cigs, indel_start=ist, indel_extend=iex, r=r, max_b_rows=mbr, return_status=True,
                                       final_cigars=True)
            host = cig_mod.standardize_batch(got, refs, seqs)
            for k in range(n):
                if fin[k] != host[k] or st2[k] != st[k]:
                    bad += 1
                    log(f"GLUE MISMATCH seed={seed} round={rounds} read={k} r={r} mbr={mbr} len={len(refs[k])}/{len(seqs[k])} status={st[k]}/{st2[k]}")
        ctx.close()
        for k in range(n):
            try:
                want, wst = oracle.align(refs[k], seqs[k], cigs[k], sub, nps, indel_start=ist, indel_extend=iex, r=r,
                                         max_b_rows=mbr, max_n=max_n, max_l=max_l, return_status=True)
            except ValueError:
                want, wst = None, -1
            if want is None:
                ok = st[k] & 32
            else:
                ok = got[k] == want and st[k] == wst
            if not ok:
                bad += 1
                log(f"MISMATCH seed={seed} round={rounds} read={k} r={r} mbr={mbr} gaps=({ist},{iex}) tables={kind} max_n={max_n} "
                      f"max_l={max_l} len={len(refs[k])}/{len(seqs[k])} status={st[k]} want_status={wst}")
        rounds += 1
        reads += n
    return rounds, reads, bad


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rounds, reads, bad = fuzz(budget, seed, focus=len(sys.argv) > 3 and sys.argv[3] == "focus")
    print(f"fuzz: {rounds} rounds, {reads} reads, {bad} mismatches in {budget:.0f}s (seed {seed})")
    sys.exit(1 if bad else 0)
