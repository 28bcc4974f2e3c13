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


def push_indels_left(cigar, seq, push_op):
    """Push runs of `push_op` (1 = I, 2 = D) as far left as the sequence allows
    (src/cig.pyx:102-159).  `cigar` (list/array of op codes M=0, I=1, D=2, '='=7,
    X=8) is modified in place and returned; `seq` is the read (for I) or the
    reference (for D) as codes."""
    M, E, X = 0, 7, 8
    n = len(cigar)
    seq_ptr = cig_ptr = 0
    while cig_ptr < n:
        op = cigar[cig_ptr]
        if op == push_op:
            indel_len = 1
            while cig_ptr + indel_len < n and cigar[cig_ptr + indel_len] == push_op:
                indel_len += 1
        else:
            cig_ptr += 1
            if op == M or op == X or op == E:
                seq_ptr += 1
            continue
        nshifts = 0
        while (cig_ptr - nshifts > 0 and seq_ptr - nshifts > 0 and
               seq[seq_ptr - nshifts - 1] == seq[seq_ptr - nshifts - 1 + indel_len] and
               (cigar[cig_ptr - nshifts - 1] == E or cigar[cig_ptr - nshifts - 1] == M)):
            nshifts += 1
        if nshifts:
            moved = list(cigar[cig_ptr - nshifts:cig_ptr])
            indel = list(cigar[cig_ptr:cig_ptr + indel_len])
            cigar[cig_ptr - nshifts:cig_ptr - nshifts + indel_len] = indel
            cigar[cig_ptr - nshifts + indel_len:cig_ptr + indel_len] = moved
        cig_ptr += indel_len
        # (reference: `op == push_op` here, so the pointer of the pushed sequence advances)
        seq_ptr += indel_len
    return cigar


def push_inss_thru_dels(cigar):
    """Let insertions move left through adjacent deletions: 'DDII' -> 'IIDD'
    (src/cig.pyx:164-192); in place."""
    I, D = 1, 2
    n = len(cigar)
    for i in range(n - 1):
        if cigar[i] == D and cigar[i + 1] == I:
            del_idx = i - 1
            while del_idx >= 0 and cigar[del_idx] == D:
                del_idx -= 1
            dels = i - del_idx
            ins_idx = i + 1
            while ins_idx < n and cigar[ins_idx] == I:
                ins_idx += 1
            inss = ins_idx - i - 1
            for j in range(inss):
                cigar[del_idx + 1 + j] = I
            for j in range(dels):
                cigar[del_idx + 1 + inss + j] = D
    return cigar


def standardize(aln, int_ref, int_seq):
    """What realign_read does with align()'s string (src/bam.pyx:65-78): X,= -> M, ONE
    pass of push D left / I through D / push I left / I through D (the reference's
    `while True` always stops after one pass: its `old_cig = int_cig[:]` is a numpy view
    of the array the push functions modify in place, so same_cigar is trivially true),
    then 'ID' -> 'M'.  Returns the expanded op string over 'MID'."""
    cig = [0 if c in "X=M" else (1 if c == "I" else 2) for c in aln]
    ref = np.asarray(int_ref).tolist()
    seq = np.asarray(int_seq).tolist()
    push_indels_left(cig, ref, 2)
    push_inss_thru_dels(cig)
    push_indels_left(cig, seq, 1)
    push_inss_thru_dels(cig)
    return "".join("MID"[c] for c in cig).replace("ID", "M")


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
