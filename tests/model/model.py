"""ctypes wrapper for tests/model/libpull_model.so (test infrastructure: the
product's cell recurrence + packing, executed cell by cell on the host)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libpull_model.so")
    deps = [os.path.join(_HERE, "pull_model.cpp")] + [
        os.path.join(_HERE, "..", "..", "npore_amd", "csrc", f) for f in ("cell.hpp", "layout.hpp")] + [os.path.join(_HERE, "host_prep.hpp")]
    if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
                               "-o", so, deps[0]])
    return so


def load():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(build())
        lib.pull_model_align.argtypes = [
            C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p,
            C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int64,
            C.POINTER(C.c_int32)]
        lib.pull_model_align.restype = C.c_int64
        lib.pull_model_np_info.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        lib.pull_model_np_info.restype = None
        _LIB = lib
    return _LIB


def align(ref, seq, cigar, sub, nps, indel_start=5, indel_extend=1, max_b_rows=20000, r=30,
          max_n=6, max_l=100):
    lib = load()
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    cig = cigar.encode() if isinstance(cigar, str) else bytes(cigar)
    sub = np.ascontiguousarray(sub, dtype=np.float32)
    nps = np.ascontiguousarray(nps, dtype=np.float32)
    cap = len(ref) + len(seq) + 16
    out = C.create_string_buffer(cap)
    st = C.c_int32(0)
    n = lib.pull_model_align(ref.ctypes.data, len(ref), seq.ctypes.data, len(seq), cig, len(cig),
                             sub.ctypes.data, nps.ctypes.data, max_n, max_l, indel_start, indel_extend,
                             max_b_rows, r, C.addressof(out), cap, C.byref(st))
    if n < 0:
        raise ValueError(f"model align failed status={st.value}")
    return out.raw[:n].decode(), st.value


def prep(ref, seq, cigar, max_b_rows=20000, max_n=6, max_l=100):
    """Host-prepared arrays of one read (dict of numpy arrays)."""
    lib = load()
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    cig = cigar.encode() if isinstance(cigar, str) else bytes(cigar)
    cap = 2 * len(cig) + 8
    nchmax = cap // max(1, max_b_rows - 1) + 4
    steps = np.zeros(cap, np.uint8); inss = np.zeros(cap + 1, np.int32)
    geom = np.zeros(7 * nchmax, np.int32)
    seqw = np.zeros(len(seq) + nchmax + 8, np.uint32)
    refw = np.zeros(4 * (len(ref) + nchmax + 8), np.uint32)
    refl = np.zeros(8 * (len(ref) + nchmax + 8), np.uint8)
    ns, ni, nsw, nrw = (C.c_int64() for _ in range(4))
    lib.pull_model_prep.restype = C.c_int64
    lib.pull_model_prep.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_char_p, C.c_int64, C.c_int, C.c_int,
                                    C.c_int] + [C.c_void_p] * 10
    nch = lib.pull_model_prep(ref.ctypes.data, len(ref), seq.ctypes.data, len(seq), cig, len(cig), max_n, max_l,
                              max_b_rows, steps.ctypes.data, C.addressof(ns), inss.ctypes.data, C.addressof(ni),
                              geom.ctypes.data, seqw.ctypes.data, C.addressof(nsw), refw.ctypes.data,
                              refl.ctypes.data, C.addressof(nrw))
    if nch < 0:
        raise ValueError("bad input")
    return dict(n_chunks=int(nch), steps=steps[:ns.value], inss=inss[:ni.value], geom=geom[:7 * nch].reshape(-1, 7),
                seqw=seqw[:nsw.value], refw=refw[:4 * nrw.value].reshape(-1, 4), refl=refl[:8 * nrw.value].reshape(-1, 8))


def get_np_info(seq, max_n=6, max_l=100):
    lib = load()
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    out = np.zeros((len(seq), 2, max_n), dtype=np.int32)
    lib.pull_model_np_info(seq.ctypes.data, len(seq), max_n, max_l, out.ctypes.data)
    return out
