"""Deterministic synthetic ONT-like reads (SURVEY.md section 8(d) spec).

One RNG stream per read: numpy Generator(PCG64(base_seed*1_000_003 + index)).
Reference sequence: built left to right; with probability p_np the next
element is an n-polymer (period n~U{1..6}, unit uniform over ACGT^n, copies
l~U{3..14}), else one uniform base; truncated to ref_len.  Read: walk the
reference with the error model of reference test/generate_bam.py:78-98 (5 %
insertion before a base, then 3 % substitution, else 3 % deletion, else
copy); in addition at each n-polymer start, with probability p_cnv, one whole
unit is added ('I'*n) or dropped ('D'*n) so the LEN/SHR states are exercised.
The CIGAR handed to align() is the true edit script over '=XID'.

All draws are bulk numpy draws in a fixed order, so the output is a pure
function of (base_seed, index, ref_len, p_np, p_cnv).
"""
import numpy as np

_MIXED_PNP = (0.0, 0.02, 0.05, 0.15)


def make_ref(rng, ref_len, p_np):
    """Returns (codes uint8[ref_len] in 1..4, np_starts int64[], np_period int64[])."""
    k = ref_len  # at most ref_len elements are needed
    is_np = rng.random(k) < p_np
    period = rng.integers(1, 7, size=k)
    copies = rng.integers(3, 15, size=k)
    units = rng.integers(1, 5, size=(k, 6)).astype(np.uint8)
    single = rng.integers(1, 5, size=k).astype(np.uint8)
    seg_len = np.where(is_np, period * copies, 1)
    ends = np.cumsum(seg_len)
    n_el = int(np.searchsorted(ends, ref_len, side="left")) + 1
    out = np.empty(int(ends[n_el - 1]), dtype=np.uint8)
    starts = ends[:n_el] - seg_len[:n_el]
    # singles first (vectorised), then the (few) n-polymers
    sing_idx = np.nonzero(~is_np[:n_el])[0]
    out[starts[sing_idx]] = single[sing_idx]
    np_idx = np.nonzero(is_np[:n_el])[0]
    for e in np_idx:
        n, c, s = int(period[e]), int(copies[e]), int(starts[e])
        out[s:s + n * c] = np.tile(units[e, :n], c)
    keep = starts[np_idx] + period[np_idx] * 3 <= ref_len  # at least 3 copies survive truncation
    return out[:ref_len], starts[np_idx][keep], period[np_idx][keep]


def make_read(rng, ref, np_starts, np_period, p_cnv):
    """Returns (seq codes uint8[], cigar bytes over b'=XID')."""
    R = len(ref)
    u_ins = rng.random(R) < 0.05
    ins_base = rng.integers(1, 5, size=R).astype(np.uint8)
    u_sub = rng.random(R) < 0.03
    sub_off = rng.integers(1, 4, size=R)
    u_del = rng.random(R) < 0.03
    u_cnv = rng.random(len(np_starts)) < p_cnv
    cnv_add = rng.random(len(np_starts)) < 0.5

    # main op per reference base: 0 '=', 1 'X', 2 'D'
    main = np.where(u_sub, 1, np.where(u_del, 2, 0)).astype(np.int8)
    pre_ins = u_ins.astype(np.int64)          # number of inserted bases before ref base j
    cnv_ins_at = {}                            # j -> unit to insert (before the 1-base insertion)
    for s, n, do, add in zip(np_starts, np_period, u_cnv, cnv_add):
        if not do:
            continue
        s, n = int(s), int(n)
        if add:
            cnv_ins_at[s] = ref[s:s + n].copy()
        else:
            main[s:s + n] = 2
            pre_ins[s:s + n] = 0
    sub_base = ((ref.astype(np.int64) - 1 + sub_off) % 4 + 1).astype(np.uint8)

    # assemble: per ref base j -> [cnv unit 'I'*n] ['I'] [main op]
    cnv_len = np.zeros(R, dtype=np.int64)
    for j, unit in cnv_ins_at.items():
        cnv_len[j] = len(unit)
    ops_per = cnv_len + pre_ins + 1
    op_end = np.cumsum(ops_per)
    total = int(op_end[-1]) if R else 0
    cig = np.full(total, ord("I"), dtype=np.uint8)
    main_pos = op_end - 1
    cig[main_pos] = np.array([ord("="), ord("X"), ord("D")], dtype=np.uint8)[main]

    # bases emitted by each op ('D' emits none)
    seq_full = np.zeros(total, dtype=np.uint8)
    seq_full[main_pos] = np.where(main == 0, ref, np.where(main == 1, sub_base, 0))
    one_ins = np.nonzero(pre_ins)[0]
    seq_full[main_pos[one_ins] - 1] = ins_base[one_ins]
    for j, unit in cnv_ins_at.items():
        e = int(main_pos[j] - pre_ins[j])
        seq_full[e - len(unit):e] = unit
    seq = seq_full[cig != ord("D")]
    return np.ascontiguousarray(seq), cig.tobytes()


def make_pair(base_seed, index, ref_len=10_000, p_np=0.05, p_cnv=0.3, mixed=False):
    """One (ref, seq, cigar) triple.  mixed=True draws p_np per read from
    {0, 0.02, 0.05, 0.15} (configs C3/C4)."""
    rng = np.random.Generator(np.random.PCG64(base_seed * 1_000_003 + index))
    if mixed:
        p_np = _MIXED_PNP[int(rng.integers(0, 4))]
    ref, starts, period = make_ref(rng, ref_len, p_np)
    seq, cig = make_read(rng, ref, starts, period, p_cnv)
    return ref, seq, cig


def make_batch(base_seed, n_reads, ref_len=10_000, p_np=0.05, p_cnv=0.3, mixed=False,
               first=0, stride=1):
    """Reads first, first+stride, ... (n_reads of them): lists of refs, seqs, cigars."""
    refs, seqs, cigs = [], [], []
    for k in range(n_reads):
        r, s, c = make_pair(base_seed, first + k * stride, ref_len, p_np, p_cnv, mixed)
        refs.append(r); seqs.append(s); cigs.append(c)
    return refs, seqs, cigs


def make_span(a):
    """make_batch for worker pools: a = (base_seed, n_reads, ref_len, mixed, first, stride)."""
    seed, cnt, ref_len, mixed, first, stride = a
    return make_batch(seed, cnt, ref_len=ref_len, mixed=mixed, first=first, stride=stride)
