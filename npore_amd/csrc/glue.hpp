// glue.hpp -- host-side CIGAR standardisation that realign_read applies to the string
// align() returns (reference src/bam.pyx:65-78), in C++ because the reference's is
// compiled Cython (src/cig.pyx:102-192).  One read per task.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace npore {

// The standardisation works on RUNS (op, length) of the alignment instead of one byte per op:
//   * push_indels_left (src/cig.pyx:102-159) moves a run of k insertions (deletions) one op at a time past the match
//     ops on its left while the sequence it consumes stays the same, seq[x] == seq[x + k].  A run can only ever pass
//     ops of the match run right before it, so on runs: split that match run where the k-periodicity of the sequence
//     ends, and put the indel run in between.  Runs are taken left to right and each sees the list as the previous
//     ones left it, like the reference's in-place scan.
//   * push_inss_thru_dels (src/cig.pyx:164-192) swaps every 'D..D I..I' its left-to-right scan meets, which cascades
//     until each maximal block of I / D ops reads 'I..I D..D'.
//   * 'ID' -> 'M' (str.replace: left to right, non-overlapping) is then exactly one pair per such block.
// O(runs + positions moved) instead of O(ops) per pass; npore_amd/cig.py holds the same formulation in Python and
// the tests compare both with the per-op restatement of the reference (oracle/glue_literal.py) and with G4.
enum : uint8_t { OP_M = 0, OP_I = 1, OP_D = 2 };

struct OpRun {
    int64_t len : 56;
    uint64_t op : 8;
};
static_assert(sizeof(OpRun) == 8, "a run is one 8-byte word");

inline void push_run(std::vector<OpRun> &runs, uint8_t op, int64_t n)
{
    if (n <= 0) return;
    if (!runs.empty() && runs.back().op == op) runs.back().len += n;
    else runs.push_back(OpRun{n, op});
}

// src/cig.pyx:102-159 on runs; `seq` is what push_op consumes besides the match ops (reference for D, read for I)
inline void push_indels_left(const std::vector<OpRun> &in, std::vector<OpRun> &out, const uint8_t *seq, int64_t seq_len,
                             uint8_t push_op)
{
    out.clear();
    int64_t p = 0;                                   // position in seq of the next op
    for (const OpRun &r : in) {
        if (r.op != push_op) {
            push_run(out, r.op, r.len);
            if (r.op == OP_M) p += r.len;
            continue;
        }
        const int64_t k = r.len, m = (!out.empty() && out.back().op == OP_M) ? out.back().len : 0;
        int64_t s = 0;
        while (s < m && p - s - 1 + k < seq_len && seq[p - s - 1] == seq[p - s - 1 + k]) s++;     // (the reference indexes unchecked)
        if (s) {
            out.back().len -= s;
            if (out.back().len == 0) out.pop_back();
        }
        push_run(out, push_op, k);
        push_run(out, OP_M, s);
        p += k;
    }
}

// src/cig.pyx:164-192 on runs
inline void inss_before_dels(const std::vector<OpRun> &in, std::vector<OpRun> &out)
{
    out.clear();
    for (size_t k = 0; k < in.size();) {
        if (in[k].op == OP_M) { push_run(out, OP_M, in[k].len); k++; continue; }
        int64_t ni = 0, nd = 0;
        for (; k < in.size() && in[k].op != OP_M; k++) (in[k].op == OP_I ? ni : nd) += in[k].len;
        push_run(out, OP_I, ni);
        push_run(out, OP_D, nd);
    }
}

// src/bam.pyx:65-78: one pass (the reference's loop always stops after one: its `old_cig` is a view of
// the array the push functions modify), then 'ID' -> 'M'.  Calls emit(op, n) for every run of the result
// ('M', 'I' or 'D'; consecutive calls never repeat an op).
template <class Emit>
inline void standardize_runs(const char *aln, int64_t aln_len, const uint8_t *ref, int64_t ref_len,
                             const uint8_t *seq, int64_t seq_len, Emit emit)
{
    static thread_local std::vector<OpRun> a, b;      // (a worker thread does thousands of reads: no allocation per read)
    a.clear();
    b.clear();
    for (int64_t i = 0; i < aln_len;) {                // runs of the op string; X,=,M -> M
        const char c = aln[i];
        int64_t j = i + 1;
        if (c == 'I' || c == 'D') { while (j < aln_len && aln[j] == c) j++; }
        else { while (j < aln_len && aln[j] != 'I' && aln[j] != 'D') j++; }
        a.push_back(OpRun{j - i, (uint64_t)((c == 'I') ? OP_I : (c == 'D') ? OP_D : OP_M)});
        i = j;
    }
    push_indels_left(a, b, ref, ref_len, OP_D);
    inss_before_dels(b, a);
    push_indels_left(a, b, seq, seq_len, OP_I);
    inss_before_dels(b, a);
    b.clear();
    for (size_t k = 0; k < a.size(); k++) {
        if (a[k].op == OP_I && k + 1 < a.size() && a[k + 1].op == OP_D) {
            push_run(b, OP_I, a[k].len - 1);
            push_run(b, OP_M, 1);
            push_run(b, OP_D, a[k + 1].len - 1);
            k++;
        } else {
            push_run(b, a[k].op, a[k].len);
        }
    }
    for (const OpRun &r : b) emit("MID"[r.op], r.len);
}

// ... + collapse_cigar (src/cig.pyx:13-38): run-length encoded text
// ... written at `out` (2 bytes per op of the alignment + 16 always suffice); returns the length
inline int64_t standardize_collapsed_into(const char *aln, int64_t aln_len, const uint8_t *ref, int64_t ref_len,
                                          const uint8_t *seq, int64_t seq_len, char *out)
{
    char *o = out;
    standardize_runs(aln, aln_len, ref, ref_len, seq, seq_len, [&](char op, int64_t n) {
        char digits[24];
        int nd = 0;
        uint64_t v = (uint64_t)n;
        do { digits[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (nd) *o++ = digits[--nd];
        *o++ = op;
    });
    return (int64_t)(o - out);
}
inline std::string standardize_collapsed(const char *aln, int64_t aln_len, const uint8_t *ref, int64_t ref_len,
                                         const uint8_t *seq, int64_t seq_len)
{
    std::string out((size_t)(2 * aln_len + 16), '\0');
    out.resize((size_t)standardize_collapsed_into(aln, aln_len, ref, ref_len, seq, seq_len, &out[0]));
    return out;
}

// the expanded op string itself (what realign_hap returns, src/bam.pyx:116); at most aln_len ops
inline int64_t standardize_expanded(const char *aln, int64_t aln_len, const uint8_t *ref, int64_t ref_len,
                                    const uint8_t *seq, int64_t seq_len, char *out)
{
    int64_t n = 0;
    standardize_runs(aln, aln_len, ref, ref_len, seq, seq_len, [&](char op, int64_t len) {
        std::memset(out + n, op, (size_t)len);
        n += len;
    });
    return n;
}

}  // namespace npore
