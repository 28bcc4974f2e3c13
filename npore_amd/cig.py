"""CIGAR utilities either side of align() -- host glue of the realignment path.

Counterparts of reference src/cig.pyx: expand_cigar / collapse_cigar (13-57),
bases_to_int (212-229), push_indels_left (102-159), push_inss_thru_dels
(164-192), and the one-pass standardisation that realign_read applies to the
string align() returns (src/bam.pyx:65-78).  Plain Python/numpy: sequential byte
scans, O(length) per read.
"""
import numpy as np

from . import cfg

_BASE_LUT = np.zeros(256, dtype=np.uint8)
for _c, _v in (("N", 0), ("A", 1), ("C", 2), ("G", 3), ("T", 4), ("-", 5)):
    _BASE_LUT[ord(_c)] = _v          # like the reference, lower case is NOT mapped (callers .upper())


def expand_cigar(cigar):
    """'1D3M2I' -> 'DMMMII' (src/cig.pyx:42-57)."""
    if len(cigar) > 4096:
        return _expand_cigar_np(cigar)       # chromosome-length strings (realign_haps): same result, vectorised
    out, count = [], 0
    for ch in cigar:
        if "0" <= ch <= "9":
            count = count * 10 + ord(ch) - 48
        else:
            out.append(ch * count)
            count = 0
    return "".join(out)


def _expand_cigar_np(cigar):
    b = np.frombuffer(cigar.encode(), dtype=np.uint8)
    is_digit = (b >= 48) & (b <= 57)
    op_pos = np.flatnonzero(~is_digit)
    if len(op_pos) == 0:
        return ""
    # the digits in front of every op, least significant first
    counts = np.zeros(len(op_pos), np.int64)
    alive = np.ones(len(op_pos), bool)
    mult = 1
    for k in range(1, 19):
        idx = op_pos - k
        d = b[np.maximum(idx, 0)]
        alive &= (idx >= 0) & (d >= 48) & (d <= 57)
        if not alive.any():
            break
        counts += np.where(alive, (d.astype(np.int64) - 48) * mult, 0)
        mult *= 10
    return np.repeat(b[op_pos], counts).tobytes().decode()


def collapse_cigar(extended_cigar, return_groups=False):
    """'DMMMII' -> '1D3M2I' (run-length encoding; src/cig.pyx:13-38)."""
    from itertools import groupby
    groups = [(sum(1 for _ in run), op) for op, run in groupby(extended_cigar)]
    return groups if return_groups else "".join(f"{n}{op}" for n, op in groups)


def bases_to_int(seq):
    """'NACGT-' -> uint8 codes 0..5 (src/cig.pyx:212-229); anything else -> 0."""
    return _BASE_LUT[np.frombuffer(seq.encode("ascii", "replace"), dtype=np.uint8)]


def int_to_bases(int_seq):
    return "".join(cfg.bases[i] for i in int_seq)


def cig_to_int(cig):
    return np.array([cfg.cigar_dict[c] for c in cig], dtype=np.uint8)


def int_to_cig(int_cig):
    return "".join(cfg.cigars[i] for i in int_cig)


def to_runs(ops):
    """[(op, length)] of an iterable of op codes (maximal runs)."""
    runs = []
    for op in ops:
        if runs and runs[-1][0] == op:
            runs[-1][1] += 1
        else:
            runs.append([op, 1])
    return runs


def _push(runs, op, n):
    if n > 0:
        if runs and runs[-1][0] == op:
            runs[-1][1] += n
        else:
            runs.append([op, n])


def push_indels_left_runs(runs, seq, push_op):
    """push_indels_left (reference src/cig.pyx:102-159) on RUNS [(op, length)] over M=0, I=1, D=2.
    The reference moves a run of `push_op` one op at a time past the match ops on its left while the sequence it
    consumes stays the same (seq[x] == seq[x + k] for a run of k); a run can only ever pass ops of the match run right
    before it, so on runs this is: split that match run where the k-periodicity of `seq` ends, and put the indel run
    in between.  Runs are handled left to right and each one sees the list as the previous ones left it, like the
    in-place scan of the reference.  O(runs + positions moved)."""
    out = []
    p = 0                                    # position in seq of the next op (M and push_op consume it)
    for op, k in runs:
        if op != push_op:
            _push(out, op, k)
            if op == 0:
                p += k
            continue
        m = out[-1][1] if out and out[-1][0] == 0 else 0
        s = 0
        while s < m and seq[p - s - 1] == seq[p - s - 1 + k]:
            s += 1
        if s:
            out[-1][1] -= s
            if out[-1][1] == 0:
                out.pop()
        _push(out, push_op, k)
        _push(out, 0, s)
        p += k
    return out


def inss_before_dels_runs(runs):
    """push_inss_thru_dels (reference src/cig.pyx:164-192) on runs: its left-to-right scan swaps every 'D..D I..I'
    it meets, which cascades until each maximal block of I / D ops reads 'I..I D..D'."""
    out = []
    k = 0
    while k < len(runs):
        if runs[k][0] == 0:
            _push(out, 0, runs[k][1])
            k += 1
            continue
        ni = nd = 0
        while k < len(runs) and runs[k][0] != 0:
            if runs[k][0] == 1:
                ni += runs[k][1]
            else:
                nd += runs[k][1]
            k += 1
        _push(out, 1, ni)
        _push(out, 2, nd)
    return out


def standardize_runs(aln, int_ref, int_seq):
    """Runs [(op char, length)] of the final CIGAR: what realign_read does with align()'s string (src/bam.pyx:65-78)
    -- X,= -> M, ONE pass of push D left / I through D / push I left / I through D (the reference's `while True`
    always stops after one pass: its `old_cig = int_cig[:]` is a numpy view of the array the push functions modify in
    place, so same_cigar is trivially true), then 'ID' -> 'M' (str.replace: left to right, non-overlapping; in a block
    'I..I D..D' that is exactly one pair)."""
    runs = to_runs(0 if c in "X=M" else (1 if c == "I" else 2) for c in aln)
    ref = np.asarray(int_ref).tolist()
    seq = np.asarray(int_seq).tolist()
    runs = inss_before_dels_runs(push_indels_left_runs(runs, ref, 2))
    runs = inss_before_dels_runs(push_indels_left_runs(runs, seq, 1))
    out = []
    k = 0
    while k < len(runs):
        op, n = runs[k]
        if op == 1 and k + 1 < len(runs) and runs[k + 1][0] == 2:
            _push(out, 1, n - 1)
            _push(out, 0, 1)
            _push(out, 2, runs[k + 1][1] - 1)
            k += 2
        else:
            _push(out, op, n)
            k += 1
    return [("MID"[op], n) for op, n in out]


def standardize(aln, int_ref, int_seq):
    """The expanded op string over 'MID' of standardize_runs()."""
    return "".join(c * n for c, n in standardize_runs(aln, int_ref, int_seq))


def standardize_batch(alns, int_refs, int_seqs, threads=0, expanded=False):
    """Collapsed final CIGARs for a batch: `collapse_cigar(standardize(...))` per read, done by
    the library's C++ glue (npore_standardize_batch) on all host cores.  expanded=True: the op strings
    over 'MID' themselves (npore_standardize_ops_batch; what realign_hap returns)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    n = len(alns)
    if n == 0:
        return []
    ab = [a.encode() for a in alns]
    refs = [np.ascontiguousarray(x, dtype=np.uint8) for x in int_refs]
    seqs = [np.ascontiguousarray(x, dtype=np.uint8) for x in int_seqs]

    def pack(parts, lens):
        off = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        return off

    ao = pack(ab, [len(a) for a in ab])
    ro = pack(refs, [len(x) for x in refs])
    so = pack(seqs, [len(x) for x in seqs])
    oo = pack(ab, [(1 if expanded else 2) * len(a) + 16 for a in ab])
    abuf = np.frombuffer(b"".join(ab) + b"\0", dtype=np.uint8)
    rbuf = np.concatenate(refs + [np.zeros(1, np.uint8)])
    sbuf = np.concatenate(seqs + [np.zeros(1, np.uint8)])
    out = np.empty(int(oo[-1]) + 1, np.uint8)
    olen = np.zeros(n, np.int64)
    fn = lib.npore_standardize_ops_batch if expanded else lib.npore_standardize_batch
    rc = fn(n, abuf.ctypes.data, ao.ctypes.data, rbuf.ctypes.data, ro.ctypes.data,
            sbuf.ctypes.data, so.ctypes.data, out.ctypes.data, oo.ctypes.data, olen.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError(f"npore_standardize_batch: {rc} {_lib.last_error()}")
    return [out[oo[i]:oo[i] + olen[i]].tobytes().decode() for i in range(n)]
