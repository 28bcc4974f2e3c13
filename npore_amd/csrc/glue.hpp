// glue.hpp -- host-side CIGAR standardisation that realign_read applies to the string
// align() returns (reference src/bam.pyx:65-78), in C++ because the reference's is
// compiled Cython (src/cig.pyx:102-192).  One read per task.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "std_stream.hpp"

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
// npore_amd/cig.py states these steps as passes over run lists; std_stream.hpp (what runs here and on the device) as ONE
// streaming pass with O(1) state per step.  The tests compare both with the per-op restatement of the reference
// (oracle/glue_literal.py) and with G4.
enum : uint8_t { OP_M = 0, OP_I = 1, OP_D = 2 };

// first position >= i of an 'I' or a 'D' in aln[0, n) (n if there is none): eight bytes per look
inline int64_t next_indel(const char *aln, int64_t i, int64_t n)
{
    const uint64_t ones = 0x0101010101010101ull, high = 0x8080808080808080ull;
    while (i + 8 <= n) {
        uint64_t w;
        std::memcpy(&w, aln + i, 8);
        const uint64_t a = w ^ (ones * (uint8_t)'I'), b = w ^ (ones * (uint8_t)'D');
        const uint64_t z = ((a - ones) & ~a & high) | ((b - ones) & ~b & high);      // the lowest marked byte is the first match
        if (z) return i + (__builtin_ctzll(z) >> 3);
        i += 8;
    }
    while (i < n && aln[i] != 'I' && aln[i] != 'D') i++;
    return i;
}

// src/bam.pyx:65-78: one pass (the reference's loop always stops after one: its `old_cig` is a view of
// the array the push functions modify), then 'ID' -> 'M'.  Calls emit(op, n) for every run of the result
// ('M', 'I' or 'D'; consecutive calls never repeat an op).
// The op string is cut into runs eight bytes at a time and every run goes through the five stages of std_stream.hpp at
// once (no run arrays: 50 us per 10 kb read where five passes over arrays took 88).
template <class Emit>
inline void standardize_runs(const char *aln, int64_t aln_len, const uint8_t *ref, int64_t ref_len,
                             const uint8_t *seq, int64_t seq_len, Emit emit)
{
    struct Sink {
        Emit &e;
        void operator()(uint32_t op, int64_t n) { e("MID"[op], n); }
    } sink{emit};
    static_assert((int)SOP_M == (int)OP_M && (int)SOP_I == (int)OP_I && (int)SOP_D == (int)OP_D, "one op numbering");
    StdStream<Sink> st(sink, ref, ref_len, seq, seq_len);
    for (int64_t i = 0; i < aln_len;) {
        const char c = aln[i];
        int64_t j = i + 1;
        if (c == 'I' || c == 'D') { while (j < aln_len && aln[j] == c) j++; }
        else j = next_indel(aln, j, aln_len);
        st.feed((c == 'I') ? SOP_I : (c == 'D') ? SOP_D : SOP_M, j - i);
        i = j;
    }
    st.finish();
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
