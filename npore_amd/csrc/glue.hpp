// glue.hpp -- host-side CIGAR standardisation that realign_read applies to the string
// align() returns (reference src/bam.pyx:65-78), in C++ because the reference's is
// compiled Cython (src/cig.pyx:102-192).  Sequential byte scans, one read per task.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace npore {

// op codes of the reference (src/cfg.py:28-32): M=0 I=1 D=2 '='=7 X=8
enum : uint8_t { OP_M = 0, OP_I = 1, OP_D = 2, OP_E = 7, OP_X = 8 };

// push_indels_left, src/cig.pyx:102-159 (in place)
inline void push_indels_left(std::vector<uint8_t> &cigar, const uint8_t *seq, int64_t seq_len, uint8_t push_op)
{
    const int64_t n = (int64_t)cigar.size();
    int64_t seq_ptr = 0, cig_ptr = 0;
    std::vector<uint8_t> moved;
    while (cig_ptr < n) {
        const uint8_t op = cigar[cig_ptr];
        int64_t indel_len;
        if (op == push_op) {
            indel_len = 1;
            while (cig_ptr + indel_len < n && cigar[cig_ptr + indel_len] == push_op) indel_len++;
        } else {
            cig_ptr++;
            if (op == OP_M || op == OP_X || op == OP_E) seq_ptr++;
            continue;
        }
        int64_t nshifts = 0;
        while (cig_ptr - nshifts > 0 && seq_ptr - nshifts > 0 &&
               seq_ptr - nshifts - 1 + indel_len < seq_len &&          // (the reference indexes unchecked)
               seq[seq_ptr - nshifts - 1] == seq[seq_ptr - nshifts - 1 + indel_len] &&
               (cigar[cig_ptr - nshifts - 1] == OP_E || cigar[cig_ptr - nshifts - 1] == OP_M))
            nshifts++;
        if (nshifts) {
            moved.assign(cigar.begin() + (cig_ptr - nshifts), cigar.begin() + cig_ptr);
            for (int64_t i = 0; i < indel_len; i++) cigar[cig_ptr - nshifts + i] = push_op;
            for (int64_t i = 0; i < nshifts; i++) cigar[cig_ptr - nshifts + indel_len + i] = moved[i];
        }
        cig_ptr += indel_len;
        seq_ptr += indel_len;     // op == push_op here
    }
}

// push_inss_thru_dels, src/cig.pyx:164-192 (in place)
inline void push_inss_thru_dels(std::vector<uint8_t> &cigar)
{
    const int64_t n = (int64_t)cigar.size();
    for (int64_t i = 0; i + 1 < n; i++) {
        if (cigar[i] == OP_D && cigar[i + 1] == OP_I) {
            int64_t del_idx = i - 1;
            while (del_idx >= 0 && cigar[del_idx] == OP_D) del_idx--;
            const int64_t dels = i - del_idx;
            int64_t ins_idx = i + 1;
            while (ins_idx < n && cigar[ins_idx] == OP_I) ins_idx++;
            const int64_t inss = ins_idx - i - 1;
            for (int64_t j = 0; j < inss; j++) cigar[del_idx + 1 + j] = OP_I;
            for (int64_t j = 0; j < dels; j++) cigar[del_idx + 1 + inss + j] = OP_D;
        }
    }
}

// src/bam.pyx:65-78: one pass (the reference's loop always stops after one: its `old_cig` is a view of
// the array the push functions modify), 'ID' -> 'M' (left to right, non-overlapping, like str.replace).
// Calls emit(op) for every op of the expanded result ('M', 'I' or 'D').
template <class Emit>
inline void standardize_ops(const char *aln, int64_t aln_len, const uint8_t *ref, int64_t ref_len,
                            const uint8_t *seq, int64_t seq_len, Emit emit)
{
    std::vector<uint8_t> cig((size_t)aln_len);
    for (int64_t i = 0; i < aln_len; i++) {
        const char c = aln[i];
        cig[i] = (c == 'I') ? OP_I : (c == 'D') ? OP_D : OP_M;     // X,=,M -> M
    }
    push_indels_left(cig, ref, ref_len, OP_D);
    push_inss_thru_dels(cig);
    push_indels_left(cig, seq, seq_len, OP_I);
    push_inss_thru_dels(cig);
    for (int64_t i = 0; i < aln_len;) {
        if (cig[i] == OP_I && i + 1 < aln_len && cig[i + 1] == OP_D) { emit('M'); i += 2; }
        else { emit("MID"[cig[i]]); i += 1; }
    }
}

// ... + collapse_cigar (src/cig.pyx:13-38): run-length encoded text
inline std::string standardize_collapsed(const char *aln, int64_t aln_len, const uint8_t *ref, int64_t ref_len,
                                         const uint8_t *seq, int64_t seq_len)
{
    std::string out;
    char last = 0;
    int64_t count = 0;
    auto flush = [&] {
        if (count) { out += std::to_string(count); out += last; }
    };
    standardize_ops(aln, aln_len, ref, ref_len, seq, seq_len, [&](char op) {
        if (op == last) count++;
        else { flush(); last = op; count = 1; }
    });
    flush();
    return out;
}

// the expanded op string itself (what realign_hap returns, src/bam.pyx:116); at most aln_len ops
inline int64_t standardize_expanded(const char *aln, int64_t aln_len, const uint8_t *ref, int64_t ref_len,
                                    const uint8_t *seq, int64_t seq_len, char *out)
{
    int64_t n = 0;
    standardize_ops(aln, aln_len, ref, ref_len, seq, seq_len, [&](char op) { out[n++] = op; });
    return n;
}

}  // namespace npore
