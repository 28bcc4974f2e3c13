"""Drop-in counterpart of the reference's Cython module `aln` (src/aln.pyx) for
the realignment path: same names, argument meaning and defaults.

    align(full_ref, full_seq, cigar, sub_scores, np_scores, indel_start=5,
          indel_extend=1, max_b_rows=20000, r=30, verbose=0) -> str     src/aln.pyx:379-382
    get_np_info(seq) -> int32[len, 2, max_n]                            src/aln.pyx:179
    calc_score_matrices(subs, nps, inss, dels, eps=0.01)                src/aln.pyx:62-96

The DP runs on an MI355X through libnpore_amd.so (ctypes); `align()` is a batch
of one, `align_batch()` is what throughput-minded callers (realign.py) use.
Like the reference, align()/get_np_info() read max_n / max_l from cfg.args.
"""
import ctypes as C

import numpy as np

from . import _lib, cfg

_CTX = {}      # (device, max_n, max_l, digest of the tables) -> Context: one live table set at a time
_NP_CTX = {}   # (device, max_n, max_l) -> annotation-only Context (get_np_info before any align())


class NporeError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise NporeError(f"libnpore_amd error {rc}: {_lib.last_error()}")


def _u8(a):
    if isinstance(a, (bytes, bytearray, memoryview)):
        return np.frombuffer(a, dtype=np.uint8)
    a = np.asarray(a)
    if a.dtype != np.uint8:
        a = a.astype(np.uint8)      # Cython char[::1] buffers arrive as int8
    return np.ascontiguousarray(a)


def device_count():
    """Number of visible gfx950 devices."""
    return int(_lib.load().npore_device_count())


class Context:
    """One per GPU: owns the device copy of the penalty tables and work buffers.
    sub_scores = np_scores = None makes an annotation-only context (get_np_info / np_regions)."""

    def __init__(self, sub_scores, np_scores, max_n=None, max_l=None, device=0):
        self.lib = _lib.load()
        self.max_n = int(cfg.args.max_n if max_n is None else max_n)
        self.max_l = int(cfg.args.max_l if max_l is None else max_l)
        if sub_scores is None and np_scores is None:
            self.handle = self.lib.npore_ctx_create(None, None, self.max_n, self.max_l, device)
        else:
            sub = np.ascontiguousarray(sub_scores, dtype=np.float32)
            nps = np.asarray(np_scores, dtype=np.float32)
            if sub.shape != (5, 5):
                raise ValueError("sub_scores must be float32[5,5]")
            want = (self.max_n, self.max_l + 1, self.max_l + 1)
            if nps.ndim == 3 and all(a >= b for a, b in zip(nps.shape, want)):
                # the reference hands align() the shipped [6,101,101] table whatever --max_n / --max_l say and
                # np_score clamps its indices at max_l - 1 (src/aln.pyx:257-274 as called): the part beyond
                # [max_n, max_l+1, max_l+1] is never read
                nps = nps[:want[0], :want[1], :want[2]]
            if nps.shape != want:
                raise ValueError(f"np_scores must be float32[{want[0]},{want[1]},{want[2]}] (or larger)")
            nps = np.ascontiguousarray(nps)
            self.handle = self.lib.npore_ctx_create(sub.ctypes.data, nps.ctypes.data, self.max_n, self.max_l, device)
        if not self.handle:
            raise NporeError(f"npore_ctx_create failed: {_lib.last_error()}")
        self.device = device

    def close(self):
        if getattr(self, "handle", None):
            self.lib.npore_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set(self, key, value):
        _check(self.lib.npore_ctx_set(self.handle, key.encode(), int(value)))

    def round_chunks(self, r=30):
        """Chunks (units of at most max_b_rows anti-diagonals) the GPU works on at a time at band half-width r: a
        batch costs about ceil(full-size chunks / this) chains of dependent anti-diagonals (include/npore_amd.h)."""
        return int(self.lib.npore_round_chunks(self.handle, int(r)))

    def fill_shape(self, r=30):
        """Launch geometry of the fill kernel at band half-width r (npore_fill_shape)."""
        v = (C.c_int32 * 5)()
        _check(self.lib.npore_fill_shape(self.handle, int(r), v, 5))
        nw, cpg, wg_cu, res_wg, lds = list(v)
        return {"waves_per_chunk": nw, "chunks_per_workgroup": cpg, "workgroups_per_cu": wg_cu,
                "resident_workgroups": res_wg, "resident_chunks": res_wg * cpg,
                "resident_waves_per_cu": nw * cpg * wg_cu, "lds_bytes": lds}

    _TIMING_KEYS = ("dev_prep_ms", "fill_ms", "traceback_ms", "h2d_ms", "d2h_ms", "reserved", "cells", "launches")

    def timing(self):
        """Stage times of the last completed call (npore_last_timing)."""
        t = (C.c_double * 8)()
        _check(self.lib.npore_last_timing(self.handle, t, 8))
        return dict(zip(self._TIMING_KEYS, list(t)))

    def total_timing(self):
        """Stage times summed over every group completed since the context was made (npore_total_timing)."""
        t = (C.c_double * 8)()
        _check(self.lib.npore_total_timing(self.handle, t, 8))
        return dict(zip(self._TIMING_KEYS, list(t)))

    def total_launches(self):
        return int(self.total_timing()["launches"])

    def wait(self):
        """Wait for the batches enqueued with sync=0 (npore_ctx_wait)."""
        _check(self.lib.npore_ctx_wait(self.handle))

    def align_batch(self, refs, seqs, cigars, indel_start=5, indel_extend=1, max_b_rows=20000, r=30,
                    return_status=False, final_cigars=False):
        """refs/seqs: sequences of uint8 code arrays; cigars: expanded op strings/bytes.
        Returns list[str] (and int32 status array).  final_cigars=True: what realign_read makes of the strings
        (reference src/bam.pyx:59-78: standardised, collapsed CIGAR text), done on the device (npore_align_batch_cigars)."""
        n = len(refs)
        if n == 0:
            return ([], np.zeros(0, np.int32)) if return_status else []
        refs = [_u8(x) for x in refs]
        seqs = [_u8(x) for x in seqs]
        cigs = [c.encode() if isinstance(c, str) else bytes(c) for c in cigars]
        ro = np.zeros(n + 1, np.int64); np.cumsum([len(x) for x in refs], out=ro[1:])
        so = np.zeros(n + 1, np.int64); np.cumsum([len(x) for x in seqs], out=so[1:])
        co = np.zeros(n + 1, np.int64); np.cumsum([len(x) for x in cigs], out=co[1:])
        oo = np.zeros(n + 1, np.int64)
        np.cumsum([(2 * (len(a) + len(b)) + 16) if final_cigars else (len(a) + len(b)) for a, b in zip(refs, seqs)], out=oo[1:])
        rb = np.concatenate(refs + [np.zeros(1, np.uint8)])
        sb = np.concatenate(seqs + [np.zeros(1, np.uint8)])
        cb = np.frombuffer(b"".join(cigs) + b"\0", dtype=np.uint8)
        out = np.zeros(int(oo[-1]) + 1, np.uint8)
        olen = np.zeros(n, np.int64)
        st = np.zeros(n, np.int32)
        _check((self.lib.npore_align_batch_cigars if final_cigars else self.lib.npore_align_batch)(
            self.handle, n, rb.ctypes.data, ro.ctypes.data, sb.ctypes.data, so.ctypes.data,
            cb.ctypes.data, co.ctypes.data, indel_start, indel_extend, max_b_rows, r,
            out.ctypes.data, oo.ctypes.data, olen.ctypes.data, st.ctypes.data))
        res = [out[oo[i]:oo[i] + max(int(olen[i]), 0)].tobytes().decode() for i in range(n)]
        return (res, st) if return_status else res

    def get_np_info(self, seq):
        seq = _u8(seq)
        out = np.zeros((len(seq), 2, self.max_n), dtype=np.int32)
        _check(self.lib.npore_get_np_info(self.handle, seq.ctypes.data, len(seq), out.ctypes.data))
        return out

    def np_regions(self, slices):
        """get_np_regions (src/bed.py:56-76) for a batch of independent base-code slices: per period n
        (index n-1) and slice, (positions within the slice, repeat counts L) of the n-polymer starts, in
        position order -- the region of an entry is [pos, pos + n * L)."""
        import ctypes as C
        slices = [_u8(x) for x in slices]
        k = len(slices)
        if k == 0:
            return [[] for _ in range(self.max_n)]
        off = np.zeros(k + 1, np.int64)
        np.cumsum([len(x) for x in slices], out=off[1:])
        buf = np.concatenate(slices + [np.zeros(1, np.uint8)])
        counts = np.zeros(self.max_n * k, np.int64)
        pos, reps, total = C.c_void_p(), C.c_void_p(), C.c_int64()
        _check(self.lib.npore_np_regions(self.handle, buf.ctypes.data, off.ctypes.data, k, counts.ctypes.data,
                                         C.byref(pos), C.byref(reps), C.byref(total)))
        t = total.value
        if t:
            ap = np.ctypeslib.as_array(C.cast(pos, C.POINTER(C.c_int32)), shape=(t,)).copy()
            ar = np.ctypeslib.as_array(C.cast(reps, C.POINTER(C.c_int32)), shape=(t,)).copy()
        else:
            ap = ar = np.zeros(0, np.int32)
        cuts = np.concatenate(([0], np.cumsum(counts)))
        return [[(ap[cuts[n * k + j]:cuts[n * k + j + 1]], ar[cuts[n * k + j]:cuts[n * k + j + 1]]) for j in range(k)]
                for n in range(self.max_n)]


def _context_for(sub_scores, np_scores, device=0):
    """The context holding these tables: keyed on their CONTENT (a caller that rebuilds equal arrays per call,
    as src/bam.pyx does per process, keeps its context; 245 KB hash per call, ~0.1 ms)."""
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    for a in (sub_scores, np_scores):
        a = np.ascontiguousarray(a, dtype=np.float32)
        h.update(repr(a.shape).encode())
        h.update(a.data)
    key = (device, int(cfg.args.max_n), int(cfg.args.max_l), h.digest())
    ctx = _CTX.get(key)
    if ctx is None:
        ctx = Context(sub_scores, np_scores, device=device)
        for old in _CTX.values():     # one live table set at a time, like cfg.args.*_scores
            old.close()
        _CTX.clear()
        _CTX[key] = ctx
    return ctx


def align(full_ref, full_seq, cigar, sub_scores, np_scores, indel_start=5, indel_extend=1,
          max_b_rows=20000, r=30, verbose=0):
    """Reference src/aln.pyx:379-787.  `verbose` (debug matrix printer, :744-785) is
    accepted and ignored.  Traceback inconsistencies, which the reference prints
    and logs before returning a truncated string (:689-716), raise NporeError
    only for malformed input; otherwise the (possibly truncated) string is returned."""
    ctx = _context_for(sub_scores, np_scores)
    res, st = ctx.align_batch([full_ref], [full_seq], [cigar], indel_start, indel_extend, max_b_rows, r,
                              return_status=True)
    if st[0] & 32:
        raise NporeError("align(): CIGAR does not match sequence lengths, or unsupported op / base code")
    return res[0]


def get_np_info(seq):
    """Reference src/aln.pyx:179-251: int32 [len(seq), 2, max_n], [pos, L=0 / L_IDX=1, n-1].
    Needs no tables and no prior align() (callers: src/bed.py:62, src/bam.pyx:381): it runs on an
    annotation-only context for the current cfg.args.max_n / max_l, made on first use."""
    key = (0, int(cfg.args.max_n), int(cfg.args.max_l))
    ctx = _NP_CTX.get(key)
    if ctx is None:
        ctx = _NP_CTX[key] = Context(None, None, device=0)
    return ctx.get_np_info(seq)


def fix_matrix_properties(scores, delta=0.01):
    """Reference src/aln.pyx:11-58.  `scores` is float32; every update is a
    float32 add of `delta` (NumPy >= 2 scalar semantics: np.float32 + python float
    stays float32), which is what produced the shipped golden tables."""
    ns, l = scores.shape[0], scores.shape[1]
    d = np.float32(delta)
    for n in range(ns):
        s = scores[n]
        for i in range(1, l):
            s[0, i] = 20
            s[1, i] = 20
            s[2, i] = 20
            s[i, i] = 0
        for j in range(1, l):                      # more insertions -> more penalty
            for i in range(j - 1, -1, -1):
                s[i, j] = max(s[i, j], np.float32(s[i + 1, j] + d), np.float32(s[i, j - 1] + d))
        for i in range(4, l):                      # more deletions -> more penalty
            for j in range(i - 1, -1, -1):
                s[i, j] = max(s[i, j], np.float32(s[i, j + 1] + d), np.float32(s[i - 1, j] + d))
        for i in range(4, l):                      # prefer INDELs in longer n-polymers
            for j in range(1, l):
                if i != j:
                    s[i, j] = min(s[i, j], np.float32(s[i - 1, j - 1] - d))
    return scores


def calc_score_matrices(subs, nps, inss, dels, eps=0.01):
    """Reference src/aln.pyx:62-96: count matrices -> -log penalty tables."""
    max_n, max_l = int(cfg.args.max_n), int(cfg.args.max_l)
    np_scores = np.zeros_like(nps, dtype=np.float32)
    for n in range(max_n):
        for ref_len in range(max_l):
            total = np.sum(nps[n, ref_len])
            counts = nps[n, ref_len, :max_l].astype(np.int64)
            frac = (counts + eps) / (total + eps)
            np_scores[n, ref_len, :max_l] = -np.log(frac)
    np_scores = fix_matrix_properties(np_scores)

    sub_scores = np.zeros((cfg.nbases, cfg.nbases), dtype=np.float32)
    for i in range(1, cfg.nbases):
        for j in range(1, cfg.nbases):
            if i != j:
                sub_scores[i, j] = -np.log((subs[i, j] + eps) / (np.sum(subs[i]) + eps))
            else:
                sub_scores[i, j] = 0

    ins_scores = np.zeros_like(inss, dtype=np.float32)
    total = np.sum(inss)
    ins_scores[:max_l] = -np.log((inss[:max_l] + eps) / (total + eps))
    del_scores = np.zeros_like(dels, dtype=np.float32)
    total = np.sum(dels)
    del_scores[:max_l] = -np.log((dels[:max_l] + eps) / (total + eps))
    return sub_scores, np_scores, ins_scores, del_scores


def load_default_tables(stats_dir=None):
    """np.load of the four count matrices (reference src/bam.pyx:173-176) from the
    shipped guppy5_stats, then calc_score_matrices."""
    import os
    d = stats_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "guppy5_stats")
    subs, nps, inss, dels = (np.load(os.path.join(d, f"{k}_cm.npy")) for k in ("subs", "nps", "inss", "dels"))
    return calc_score_matrices(subs, nps, inss, dels)
