"""CPU model (numpy) of the WAVE-LOCAL n-polymer annotation of csrc/annot_wave.hpp -- test infrastructure.

get_np_info (reference src/aln.pyx:179-251) restated so that one wavefront can annotate a sequence window by window
(64 positions per window, one position per lane) for all periods, with nothing but registers:

  e_n[p]   = seq[p] == seq[p+n]  (p + n < len)                       "periodicity indicator"
  kf(p)    = length of the run of e_n starting at p                   q = kf / n
  own(p)   = p is an ELIGIBLE start on its own: base != N, q >= 2 (l = q + 1 >= 3 repeats) and
             (q + 1) n > L_n2[p] n2 for every shorter period n2 (final values of the SAME position)
  R(p)     = (L, L_IDX) of period n at p, by a stride-n recurrence inside a run of e_n:
               base != N and q + 1 > max_l and q + 1 >= 3  -> (max_l, 0)       starts with more than max_l repeats
                                                                                overwrite each other in turn
               e_n[p-n .. p-1] all set and L(p-n) != 0      -> (L(p-n), L_IDX(p-n) + 1)
               own(p)                                       -> (q + 1, 0)
               otherwise                                    -> (0, 0)
`recurrence()` is that statement, position by position.  `windows()` is the closed form the kernel evaluates per
window: the recurrence's chain is resolved inside the window from the window's own ballot of own(p), and enters from
the previous window through the (L, L_IDX) of that window's last n lanes (the carry) -- no look-back beyond that, and
the forward look-ahead only as far as (max_l + 2) n positions.
Both are checked against the oracle's literal loop in tests/test_model_vs_oracle.py."""
import numpy as np


def _e(seq, n):
    ln = len(seq)
    e = np.zeros(ln, bool)
    if ln > n:
        e[:ln - n] = seq[:ln - n] == seq[n:]
    return e


def _kf(e):
    """run of ones starting at p"""
    kf = np.zeros(len(e) + 1, np.int64)
    for p in range(len(e) - 1, -1, -1):
        kf[p] = kf[p + 1] + 1 if e[p] else 0
    return kf[:-1]


def recurrence(seq, max_n=6, max_l=100):
    seq = np.asarray(seq, np.uint8)
    ln = len(seq)
    out = np.zeros((ln, 2, max_n), np.int32)
    for n in range(1, max_n + 1):
        e = _e(seq, n)
        kf = _kf(e)
        conn = 0                        # run of e_n ending at p-1
        for p in range(ln):
            q = int(kf[p]) // n
            nz = seq[p] != 0
            mx = max([int(out[p, 0, n2 - 1]) * n2 for n2 in range(1, n)], default=0)
            if nz and q + 1 > max_l and q + 1 >= 3:
                out[p, 0, n - 1], out[p, 1, n - 1] = max_l, 0
            elif conn >= n and out[p - n, 0, n - 1] != 0:
                out[p, 0, n - 1], out[p, 1, n - 1] = out[p - n, 0, n - 1], out[p - n, 1, n - 1] + 1
            elif nz and q >= 2 and (q + 1) * n > mx:
                out[p, 0, n - 1], out[p, 1, n - 1] = q + 1, 0
            conn = conn + 1 if e[p] else 0
    return out


def windows(seq, max_n=6, max_l=100, W=64):
    """the kernel's per-window closed form (W lanes)"""
    seq = np.asarray(seq, np.uint8)
    ln = len(seq)
    out = np.zeros((ln, 2, max_n), np.int32)
    nwin = (ln + W - 1) // W
    lane = np.arange(W)
    es = [None] + [np.concatenate([_e(seq, n), np.zeros(2 * W, bool)]) for n in range(1, max_n + 1)]
    kfs = [None] + [_kf(es[n]) for n in range(1, max_n + 1)]
    prev = np.zeros((max_n + 1, 2, W), np.int64)          # previous window's (L, idx) per period and lane
    for w in range(nwin):
        base = w * W
        pos = base + lane
        valid = pos < ln
        b = np.where(valid, seq[np.minimum(pos, ln - 1)], 0)
        mx = np.zeros(W, np.int64)
        cur = np.zeros((max_n + 1, 2, W), np.int64)
        for n in range(1, max_n + 1):
            cap = (max_l + 2) * n
            M = es[n][base:base + W]
            # forward run, looked ahead no further than the kernel does
            kf = np.minimum(kfs[n][base:base + W], (W - lane) + cap + W)
            q = kf // n
            own = valid & (b != 0) & (q >= 2) & ((q + 1) * n > mx)
            # backward run inside the window
            kb = np.zeros(W, np.int64)
            for l in range(1, W):
                kb[l] = kb[l - 1] + 1 if M[l - 1] else 0
            J = kb // n
            # ones at the top of the previous window's mask
            B = 0
            if w > 0:
                Mp = es[n][base - W:base]
                while B < W and Mp[W - 1 - B]:
                    B += 1
            phi = lane % n
            entering = (kb == lane) & (B >= n - phi)
            cL, cI = prev[n, 0, W - n + phi], prev[n, 1, W - n + phi]
            big = valid & (b != 0) & (q + 1 > max_l) & (q + 1 >= 3)
            jcap = np.maximum(np.maximum(max_l - q, 2 - q), 0)
            capwin = valid & (b != 0) & (jcap <= J)
            L = np.zeros(W, np.int64)
            I = np.zeros(W, np.int64)
            # C: earliest set bit of the window's own ballot among pos - j n, j <= J
            for l in range(W):
                if not valid[l]:
                    continue
                if capwin[l]:
                    L[l], I[l] = max_l, jcap[l]
                elif entering[l] and cL[l] != 0:
                    L[l], I[l] = cL[l], cI[l] + l // n + 1
                else:
                    for j in range(int(J[l]), -1, -1):
                        if own[l - j * n]:
                            L[l], I[l] = j + q[l] + 1, j
                            break
            assert not (big & ~capwin).any()
            cur[n, 0], cur[n, 1] = L, I
            mx = np.maximum(mx, L * n)
            out[base:base + W, 0, n - 1][valid[:min(W, ln - base)]] = L[valid]
            out[base:base + W, 1, n - 1][valid[:min(W, ln - base)]] = I[valid]
        prev = cur
    return out
