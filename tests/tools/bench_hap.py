"""Time realign_haps (the standardize_vcf path) on chromosome-scale haplotype sequences: synthetic pairs
concatenated to the requested length, optional check of the raw alignment against the oracle.
usage: bench_hap.py [total_bases=4000000] [n_haps=2] [check=0]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from npore_amd import aln, bam, synth, vcf as V

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
n_haps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
check = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sub, nps, _, _ = aln.load_default_tables()
ctx = aln.Context(sub, nps)
t = time.time()
piece = 10_000
base_refs, base_seqs, base_cigs = synth.make_batch(2, 64, ref_len=piece)
dec = np.frombuffer(b"NACGT", np.uint8)
haps = []
for h in range(n_haps):
    order = np.random.default_rng(h).integers(0, 64, size=(L + piece - 1) // piece)
    ref = np.concatenate([base_refs[k] for k in order])
    seq = np.concatenate([base_seqs[k] for k in order])
    cig = b"".join(bytes(base_cigs[k]) for k in order)
    haps.append((f"chr{h}", h + 1, dec[seq].tobytes().decode(), dec[ref].tobytes().decode(), cig.decode()))
print(f"gen {time.time() - t:.2f}s: {n_haps} haplotypes of {len(haps[0][3])} reference bases", flush=True)
for rep in range(2):
    t = time.time()
    out = bam.realign_haps(ctx, haps, r=30)
    dt = time.time() - t
    tm = ctx.timing()
    print(f"realign_haps rep={rep}: {dt:.2f}s ({sum(len(h[3]) for h in haps) / dt / 1e6:.1f} Mbp/s)  fill={tm['fill_ms']:.0f}ms "
          f"tb={tm['traceback_ms']:.0f}ms prep={tm['dev_prep_ms']:.0f}ms h2d={tm['h2d_ms']:.0f} d2h={tm['d2h_ms']:.0f}", flush=True)
if os.environ.get("NPORE_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); bam.realign_haps(ctx, haps, r=30); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
t = time.time()
recs = V.gen_records(out)
print(f"gen_records {time.time() - t:.2f}s: {len(recs)} variants", flush=True)
if check:
    import oracle
    from npore_amd.cig import bases_to_int
    t = time.time()
    h = haps[0]
    r_, s_ = bases_to_int(h[3]), bases_to_int(h[2])
    want = oracle.align(r_, s_, h[4], sub, nps, r=30)
    got = ctx.align_batch([r_], [s_], [h[4]], r=30)[0]
    print(f"oracle {time.time() - t:.1f}s  raw alignment equal: {got == want}", flush=True)
