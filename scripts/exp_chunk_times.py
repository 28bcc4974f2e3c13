#!/usr/bin/env python3
"""Experiment: when do the chunks of one fill launch start and finish, and on which XCD / shader engine / CU / SIMD?
Needs a THROWAWAY build of the library (LABNOTES round 5, "Why 8 000 reads per launch ..."), not the product:
  kernels.hpp, fill_body: `const unsigned long long x_t0 = __builtin_amdgcn_s_memrealtime();` behind the chunk's slot is known, and
  behind `pbase += d.nrows;`: lane 0 of the chunk's first wave stores {x_t0, s_memrealtime(), s_getreg(HW_ID), s_getreg(XCC_ID),
  d.nrows, blockIdx.x} as eight 32-bit words at p.dbg + 8 * slot_id;
  npore_api.cpp: `w->dbg.ensure(max_chunks * 32 + 64)` + a memset beside the other buffers of a group, and &w->dbg as selector 8
  of npore_debug_fetch.
    python scripts/exp_chunk_times.py ab_libs/libnpore_chunktime.so [reads=4000] [r=30]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from npore_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
from npore_amd import aln, synth

n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
r = int(sys.argv[3]) if len(sys.argv) > 3 else 30
sub, nps, _, _ = aln.load_default_tables()
ctx = aln.Context(sub, nps)
refs, seqs, cigs = synth.make_batch(2, n, ref_len=10000)
for rep in range(3):
    ctx.align_batch(refs, seqs, cigs, r=r)
    t = ctx.timing()
    nch = 2 * n
    raw = np.zeros(nch * 8, np.uint32)
    assert ctx.lib.npore_debug_fetch(ctx.handle, 8, raw.ctypes.data, raw.nbytes) == 0, _lib.last_error()
    a = raw.reshape(nch, 8)
    t0 = a[:, 0].astype(np.uint64) | (a[:, 1].astype(np.uint64) << np.uint64(32))
    t1 = a[:, 2].astype(np.uint64) | (a[:, 3].astype(np.uint64) << np.uint64(32))
    rows = a[:, 6].astype(np.int64)
    ok = t1 > 0
    big = ok & (rows > 5000)
    base = t0[ok].min()
    us = lambda x: (x.astype(np.int64) - int(base)) / 100.0            # 100 MHz -> microseconds
    s, e = us(t0[big]), us(t1[big])
    d = e - s
    hw = a[big, 4]
    xcc = a[big, 5] & 0xF
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    print(f"rep {rep}: fill_ms {t['fill_ms']:.2f}; {big.sum()} big chunks ({ok.sum()} recorded); launch span {us(t1[ok]).max() / 1e3:.2f} ms")
    q = [0, 1, 10, 50, 90, 99, 100]
    print("  start us  pct", q, np.percentile(s, q).round(0))
    print("  dur   ms  pct", q, (np.percentile(d, q) / 1e3).round(2))
    print("  end   ms  pct", q, (np.percentile(e, q) / 1e3).round(2))
    print("  rows      pct", q, np.percentile(rows[big], q).round(0))
    print("  dur per row ns: mean %.1f std %.1f; corr(dur, rows) %.2f" % ((d / rows[big]).mean() * 1e3, (d / rows[big]).std() * 1e3, np.corrcoef(d, rows[big])[0, 1]))
    for name, key in (("xcc", xcc), ("simd", simd), ("se", se)):
        print("  mean end ms by", name, {int(k): round(float(e[key == k].mean() / 1e3), 2) for k in np.unique(key)})
    # the slowest chunks: where are they
    worst = np.argsort(e)[-8:]
    print("  slowest:", [(int(xcc[i]), int(se[i]), int(cu[i]), int(simd[i]), round(float(d[i] / 1e3), 2)) for i in worst], "(xcc, se, cu, simd, ms)")
    # per-CU spread: group by (xcc, se, sh, cu)
    key = (xcc.astype(np.int64) << 12) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu
    ends = np.array([e[key == k].max() for k in np.unique(key)])
    cnt = np.array([(key == k).sum() for k in np.unique(key)])
    print(f"  CUs seen {len(ends)}; chunks per CU min/max {cnt.min()}/{cnt.max()}; CU end ms pct", q, (np.percentile(ends, q) / 1e3).round(2))
ctx.close()
