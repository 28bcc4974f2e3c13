// confusion.hpp -- basecaller error profile from pileup text: the counting loop of the reference's
// calc_confusion_matrices (src/bam.pyx:351-499), host code.  Input is what the reference reads from
// `samtools mpileup -r ctg:start+1-end bam | cut -f5` (src/bam.pyx:301-316), upper-cased: one line per
// reference position the pileup reports, in order, consumed as positions start, start+1, ... exactly as the
// reference does (its `pos += 1` per line: a region with coverage gaps shifts the same way there).
//
// Formulation.  The reference walks every line character by character under the GIL; here the lines of a range
// are independent given their position, so they are counted in parallel (per-thread matrices, summed): the order
// of additions does not matter for integer counts.
#pragma once
#include <stdint.h>

#include <cstring>
#include <vector>

namespace npore {

struct ConfusionOut {
    int64_t *subs;    // [5][5]           subs[ref_base][read_base]
    int64_t *nps;     // [max_n][max_l+1][max_l+1]
    int64_t *inss;    // [max_l+1]
    int64_t *dels;    // [max_l+1]
};

inline int pileup_base_code(char c)     // src/bam.pyx:320-332 (base_to_int)
{
    switch (c) {
        case 'N': return 0;
        case 'A': return 1;
        case 'C': return 2;
        case 'G': return 3;
        case 'T': return 4;
        default: return -1;
    }
}

// One pileup line at region position `pos`.  ref_codes: base codes of refs[ctg][start:end] (n_ref of them);
// ref_text: the upper-cased contig text from `start` on (ref_text_len characters: the reference slices the whole
// contig, src/bam.pyx:459-460); np_info: get_np_info of refs[ctg][start:end+1], int32 [np_len][2][max_n].
// Returns false on a character the reference does not know (it prints an error and leaves the line, :473-476).
inline bool confusion_count_line(const char *reads, int64_t len, int64_t pos, const uint8_t *ref_codes, int64_t n_ref,
                                 const char *ref_text, int64_t ref_text_len, const int32_t *np_info, int64_t np_len,
                                 int max_n, int max_l, const ConfusionOut &o)
{
    const int dim = max_l + 1;
    auto L_at = [&](int n_idx, int which) -> int {       // np_info[pos+1, which, n_idx]; zeros past the slice
        const int64_t p = pos + 1;
        return p < np_len ? np_info[(p * 2 + which) * max_n + n_idx] : 0;
    };
    auto nps_add = [&](int n_idx, int a, int b) {
        if (a >= 0 && a < dim && b >= 0 && b < dim) o.nps[((int64_t)n_idx * dim + a) * dim + b] += 1;
    };
    auto unchanged = [&]() {                              // :397-402, :481-486
        for (int n_idx = 0; n_idx < max_n; n_idx++) {
            const int l = L_at(n_idx, 0), l_idx = L_at(n_idx, 1);
            if (l != 0 && l_idx == 0) nps_add(n_idx, l, l);
        }
    };
    auto number = [&](int64_t &i) -> int64_t {            // :409-417 / :440-448
        int64_t v = 0;
        i++;
        while (i < len && reads[i] >= '0' && reads[i] <= '9') { v = v * 10 + (reads[i] - '0'); i++; }
        return v;
    };
    bool was_del = true, was_ins = true, ok = true;
    const int ref_base = pos < n_ref ? ref_codes[pos] : 0;
    int64_t i = 0;
    while (i < len) {
        const char c = reads[i];
        if (c == '^') i += 2;                             // start of a read + its mapping quality
        else if (c == '$' || c == '*') i += 1;            // end of a read / deleted base
        else if (pileup_base_code(c) >= 0) {
            o.subs[ref_base * 5 + pileup_base_code(c)] += 1;
            i += 1;
            if (!was_ins) o.inss[0] += 1;
            if (!was_del) o.dels[0] += 1;
            if (!was_ins && !was_del) unchanged();
            was_ins = was_del = false;
        } else if (c == '-') {
            was_del = true;
            const int64_t indel = number(i);
            bool cnv = false;
            for (int n = 1; n <= max_n; n++) {
                const int l = L_at(n - 1, 0), l_idx = L_at(n - 1, 1);
                if (l != 0 && l_idx == 0 && indel % n == 0 && indel <= (int64_t)l * n) {
                    cnv = true;
                    nps_add(n - 1, l, (int)(l - indel / n));
                } else if (l != 0 && l_idx == 0) {
                    nps_add(n - 1, l, l);
                }
            }
            if (!cnv) o.dels[indel < max_l ? indel : max_l] += 1;
            i += indel;
        } else if (c == '+') {
            was_ins = true;
            const int64_t indel = number(i);
            bool cnv = false;
            for (int n = 1; n <= max_n; n++) {
                const int l = L_at(n - 1, 0), l_idx = L_at(n - 1, 1);
                bool same = false;
                if (l != 0 && l_idx == 0 && indel % n == 0) {
                    // refs[ctg][start+pos+1 : start+pos+n+1] * (indel // n) == reads[i : i+indel]  (Python slices clip)
                    const int64_t u0 = pos + 1, ulen = u0 >= ref_text_len ? 0 : (u0 + n <= ref_text_len ? n : ref_text_len - u0);
                    const int64_t have = i >= len ? 0 : (i + indel <= len ? indel : len - i);
                    const int64_t reps = indel / n;
                    same = (ulen * reps == have);
                    for (int64_t q = 0; same && q < have; q++) same = reads[i + q] == ref_text[u0 + q % ulen];
                }
                if (same) {
                    cnv = true;
                    const int64_t to = l + indel / n;
                    nps_add(n - 1, l, (int)(to < max_l ? to : max_l));
                } else if (l != 0 && l_idx == 0) {
                    nps_add(n - 1, l, l);
                }
            }
            if (!cnv) o.inss[indel < max_l ? indel : max_l] += 1;
            i += indel;
        } else {
            ok = false;                                   // "ERROR: unexpected character"; the rest of the line is dropped
            break;
        }
    }
    if (!was_ins) o.inss[0] += 1;                         // the last read at this position, :478-486
    if (!was_del) o.dels[0] += 1;
    if (!was_ins && !was_del) unchanged();
    return ok;
}

}  // namespace npore
