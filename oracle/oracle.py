"""ctypes wrapper around oracle/libnpore_oracle.so (plain-C restatement of
reference src/aln.pyx:179-251 get_np_info and src/aln.pyx:379-787 align).

TEST INFRASTRUCTURE ONLY: see oracle/npore_oracle.c header.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib_path():
    return os.path.join(_HERE, "libnpore_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "npore_oracle.c")
    so = lib_path()
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libnpore_oracle.so"])
    return so


def load():
    global _LIB
    if _LIB is None:
        so = lib_path()
        if not os.path.exists(so):
            build()
        lib = C.CDLL(so)
        lib.npore_oracle_get_np_info.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        lib.npore_oracle_get_np_info.restype = None
        lib.npore_oracle_align.argtypes = [
            C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_char_p, C.c_int64,
            C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
            C.c_int, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]
        lib.npore_oracle_align.restype = C.c_int64
        lib.npore_oracle_align_batch.argtypes = [
            C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.npore_oracle_align_batch.restype = C.c_int64
        _LIB = lib
    return _LIB


def _u8(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint8))


def get_np_info(seq, max_n=6, max_l=100):
    lib = load()
    seq = _u8(seq)
    out = np.zeros((len(seq), 2, max_n), dtype=np.int32)
    lib.npore_oracle_get_np_info(seq.ctypes.data, len(seq), max_n, max_l, out.ctypes.data)
    return out


def align(full_ref, full_seq, cigar, sub_scores, np_scores, indel_start=5, indel_extend=1,
          max_b_rows=20000, r=30, max_n=6, max_l=100, return_status=False):
    lib = load()
    ref = _u8(full_ref)
    seq = _u8(full_seq)
    cig = cigar.encode() if isinstance(cigar, str) else bytes(cigar)
    sub = np.ascontiguousarray(sub_scores, dtype=np.float32)
    nps = np.ascontiguousarray(np_scores, dtype=np.float32)
    assert sub.shape == (5, 5) and nps.shape == (max_n, max_l + 1, max_l + 1)
    cap = len(ref) + len(seq) + 16
    out = C.create_string_buffer(cap)
    st = C.c_int32(0)
    n = lib.npore_oracle_align(ref.ctypes.data, len(ref), seq.ctypes.data, len(seq), cig, len(cig),
                               sub.ctypes.data, nps.ctypes.data, max_n, max_l, indel_start, indel_extend,
                               max_b_rows, r, C.addressof(out), cap, C.byref(st))
    if n < 0:
        raise ValueError(f"oracle align failed, status={st.value}")
    s = out.raw[:n].decode()
    return (s, st.value) if return_status else s


def align_batch(refs, seqs, cigars, sub_scores, np_scores, indel_start=5, indel_extend=1,
                max_b_rows=20000, r=30, max_n=6, max_l=100):
    """Serial batch (one core).  refs/seqs: lists of uint8 arrays; cigars: list of bytes/str."""
    lib = load()
    n = len(refs)
    refs = [_u8(x) for x in refs]
    seqs = [_u8(x) for x in seqs]
    cigs = [c.encode() if isinstance(c, str) else bytes(c) for c in cigars]
    ro = np.zeros(n + 1, np.int64); np.cumsum([len(x) for x in refs], out=ro[1:])
    so = np.zeros(n + 1, np.int64); np.cumsum([len(x) for x in seqs], out=so[1:])
    co = np.zeros(n + 1, np.int64); np.cumsum([len(x) for x in cigs], out=co[1:])
    oo = np.zeros(n + 1, np.int64); np.cumsum([len(a) + len(b) + 16 for a, b in zip(refs, seqs)], out=oo[1:])
    rb = np.concatenate(refs) if n else np.zeros(0, np.uint8)
    sb = np.concatenate(seqs) if n else np.zeros(0, np.uint8)
    cb = np.frombuffer(b"".join(cigs), dtype=np.uint8).copy() if n else np.zeros(0, np.uint8)
    out = np.zeros(int(oo[-1]) + 1, np.uint8)
    olen = np.zeros(n, np.int64)
    st = np.zeros(n, np.int32)
    sub = np.ascontiguousarray(sub_scores, dtype=np.float32)
    nps = np.ascontiguousarray(np_scores, dtype=np.float32)
    lib.npore_oracle_align_batch(n, rb.ctypes.data, ro.ctypes.data, sb.ctypes.data, so.ctypes.data,
                                 cb.ctypes.data, co.ctypes.data, sub.ctypes.data, nps.ctypes.data,
                                 max_n, max_l, indel_start, indel_extend, max_b_rows, r,
                                 out.ctypes.data, oo.ctypes.data, olen.ctypes.data, st.ctypes.data)
    res = [out[oo[i]:oo[i] + max(olen[i], 0)].tobytes().decode() for i in range(n)]
    return res, st


def _pool_worker(args):
    refs, seqs, cigs, sub, nps, kw = args
    return align_batch(refs, seqs, cigs, sub, nps, **kw)


def align_batch_procs(refs, seqs, cigars, sub_scores, np_scores, procs, **kw):
    """align_batch on `procs` forked worker processes, the way the reference itself runs align()
    (multiprocessing.Pool over reads, src/realign.py:110-114).  Processes rather than threads: every
    align() allocates and zeroes its own 60 B/cell state matrix (241 MB at max_b_rows=20000, r=100), and
    threads of one address space serialise on those page faults."""
    import multiprocessing as mp
    n = len(refs)
    procs = max(1, min(int(procs), n))
    if procs == 1:
        return align_batch(refs, seqs, cigars, sub_scores, np_scores, **kw)
    step = (n + 4 * procs - 1) // (4 * procs)          # a few slices per worker: reads differ in length
    jobs = [(refs[i:i + step], seqs[i:i + step], cigars[i:i + step], sub_scores, np_scores, kw) for i in range(0, n, step)]
    load()                                            # build / load before forking
    with mp.get_context("fork").Pool(procs) as pool:
        parts = pool.map(_pool_worker, jobs)
    out = [a for p in parts for a in p[0]]
    st = np.concatenate([p[1] for p in parts]) if parts else np.zeros(0, np.int32)
    return out, st
