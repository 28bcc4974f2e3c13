"""TEST INFRASTRUCTURE ONLY (see oracle/npore_oracle.c header): the per-op restatement of the reference's CIGAR
standardisation -- push_indels_left (reference src/cig.pyx:102-159), push_inss_thru_dels (src/cig.pyx:164-192) and the
one-pass glue of realign_read (src/bam.pyx:65-78) -- that the product's run-based formulation (npore_amd/cig.py,
npore_amd/csrc/glue.hpp) is checked against, itself pinned by the final CIGARs of the reference's golden SAM (G4).
Nothing under npore_amd/ may import this."""
import numpy as np


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
