// host_prep.hpp -- TEST INFRASTRUCTURE: host twin of npore_amd/csrc/prep_kernels.hpp
// (CIGAR conversion, insertion prefix counts, chunk boundaries, n-polymer
// annotation, word packing).  The CPU cell model consumes these arrays, and the
// GPU tests compare the device-prepared arrays with them word for word.  The
// product library does not include this file.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../npore_amd/csrc/layout.hpp"

namespace npore {

// ---------------------------------------------------------------------------
// n-polymer annotation, reference src/aln.pyx:179-251, restated per position so
// that every (position, period) result is computed independently from run
// lengths of the periodicity indicator e_n[p] = (seq[p] == seq[p+n]):
//   kf[p] = length of the run of e_n starting at p, kb[p] = run ending at p-1.
// A start s = pos - j*n covers pos iff j <= kb[pos]/n; its repeat count is
// l = j + kf[pos]/n + 1 (0 if fewer than one full repeat).  The reference visits
// starts in ascending order and overwrites when l exceeds the stored (capped)
// value, so: the earliest eligible start wins, except that while l > max_l later
// eligible starts keep overwriting L_IDX (the stored L is capped at max_l).
// "Eligible": base != N, l > 2, and for every shorter period n2 the already
// final L[s][n2] satisfies l*n > L[s][n2]*n2 (src/aln.pyx:238-242).
// Lout / Iout: int32 [len][max_n]; both zero where no n-polymer.
// ---------------------------------------------------------------------------
inline void np_info_compute(const uint8_t *seq, int64_t len, int max_n, int max_l,
                            int32_t *Lout, int32_t *Iout)
{
    std::fill(Lout, Lout + len * max_n, 0);
    std::fill(Iout, Iout + len * max_n, 0);
    if (len <= 0) return;
    std::vector<int32_t> kf(len + 1), kb(len + 1);
    for (int n = 1; n <= max_n; n++) {
        kf[len] = 0;
        for (int64_t p = len - 1; p >= 0; p--)
            kf[p] = (p + n < len && seq[p] == seq[p + n]) ? kf[p + 1] + 1 : 0;
        kb[0] = 0;
        for (int64_t p = 1; p <= len; p++)
            kb[p] = (p - 1 + n < len && seq[p - 1] == seq[p - 1 + n]) ? kb[p - 1] + 1 : 0;
        for (int64_t pos = 0; pos < len; pos++) {
            const int q = kf[pos] / n;
            const int J = kb[pos] / n;
            int stored = 0, idx = 0;
            for (int j = J; j >= 0; j--) {
                const int l = (j == 0) ? (q >= 1 ? q + 1 : 0) : j + q + 1;
                if (stored && l <= max_l) break;   // nothing later can exceed the stored value
                if (l < 3) continue;
                const int64_t s = pos - (int64_t)j * n;
                if (!seq[s]) continue;
                bool longest = true;
                for (int n2 = 1; n2 < n; n2++)
                    if ((int64_t)l * n <= (int64_t)Lout[s * max_n + (n2 - 1)] * n2) longest = false;
                if (!longest) continue;
                if (l > stored) { stored = std::min(max_l, l); idx = j; }
            }
            Lout[pos * max_n + (n - 1)] = stored;
            Iout[pos * max_n + (n - 1)] = idx;
        }
    }
}

// ---------------------------------------------------------------------------
// One read: converted path (reference src/aln.pyx:386-392) and chunk list.
// steps[b] = 1 if converted-CIGAR op b is 'I' (row+1), 0 if 'D' (col+1);
// inss[b] = number of 'I' among ops [0,b)  (src/aln.pyx:279-292).
// Returns false if the CIGAR is malformed or disagrees with the lengths.
// ---------------------------------------------------------------------------
struct ReadPath {
    std::vector<uint8_t> steps;
    std::vector<int32_t> inss;
    std::vector<int64_t> breaks;
};

inline bool build_path(const char *cig, int64_t cig_len, int64_t seq_len, int64_t ref_len,
                       int max_b_rows, ReadPath &out)
{
    int64_t n = 0;
    for (int64_t i = 0; i < cig_len; i++) {
        char c = cig[i];
        if (c == 'X' || c == '=' || c == 'M') n += 2;
        else if (c == 'I' || c == 'D') n += 1;
        else return false;
    }
    out.steps.resize(n);
    out.inss.resize(n + 1);
    int64_t k = 0;
    for (int64_t i = 0; i < cig_len; i++) {
        char c = cig[i];
        if (c == 'I') out.steps[k++] = 1;
        else if (c == 'D') out.steps[k++] = 0;
        else { out.steps[k++] = 0; out.steps[k++] = 1; }   // X,=,M -> "DI"
    }
    out.inss[0] = 0;
    for (int64_t b = 0; b < n; b++) out.inss[b + 1] = out.inss[b] + out.steps[b];
    if (out.inss[n] != seq_len || n - out.inss[n] != ref_len) return false;

    // get_breaks, src/aln.pyx:344-358, array_size = S+R+1
    const int64_t array_size = seq_len + ref_len + 1;
    const int64_t cm1 = (int64_t)max_b_rows - 1;
    const int64_t nbrk = 1 + ((array_size - 1) + cm1 - 1) / cm1;
    out.breaks.assign(nbrk, 0);
    for (int64_t i = 0; i < nbrk - 1; i++) {
        int64_t bk = i * cm1;
        // don't split a "DI" pair: op[bk] == 'I' and op[bk-1] == 'D'
        if (i > 0 && out.steps[bk] == 1 && out.steps[bk - 1] == 0) bk -= 1;
        out.breaks[i] = bk;
    }
    out.breaks[nbrk - 1] = array_size - 1;
    return true;
}

// Pack the annotation words of one chunk (layout.hpp).  seq/ref point at the
// chunk slices (length slen/rlen, i.e. including the look-ahead base when there
// is one, src/aln.pyx:453-454).  Outputs have drows+1 / dcols+1 entries.
inline void pack_chunk_words(const uint8_t *seq, int slen, int drows,
                             const uint8_t *ref, int rlen, int dcols, int max_n, int max_l,
                             uint32_t *seqw, uint32_t *refw /* interleaved x,y,z,w */, uint8_t *refl /* 8/entry */,
                             std::vector<int32_t> &scratch)
{
    scratch.resize((size_t)2 * max_n * (size_t)(std::max(slen, rlen) + 1));
    int32_t *L = scratch.data();
    int32_t *I = L + (size_t)max_n * (std::max(slen, rlen) + 1);

    np_info_compute(seq, slen, max_n, max_l, L, I);
    for (int i = 0; i <= drows; i++) {
        uint32_t w = 0;
        for (int k = 0; k < 6; k++) {
            int p = i - 6 + k;
            uint32_t code = (p < 0) ? 7u : (uint32_t)seq[p];
            w |= code << (MER_SHIFT + 3 * k);
        }
        for (int n = 1; n <= max_n; n++) {
            int p = i - n;
            if (p >= 0 && p < slen && L[(size_t)p * max_n + (n - 1)] != 0) {
                w |= 1u << (FLAG_SHIFT + n - 1);
                if (I[(size_t)p * max_n + (n - 1)] == 0) w |= 1u << (n - 1);
            }
        }
        seqw[i] = w;
    }

    np_info_compute(ref, rlen, max_n, max_l, L, I);
    for (int j = 0; j <= dcols; j++) {
        uint32_t x = 0, y = 0;
        for (int k = 0; k < 6; k++) {
            int p = j + k;
            uint32_t code = (p >= rlen) ? 6u : (uint32_t)ref[p];
            x |= code << (MER_SHIFT + 3 * k);
        }
        uint8_t *lb = refl + (size_t)j * 8;
        std::memset(lb, 0, 8);
        for (int n = 1; n <= max_n; n++) {
            if (j < rlen) {
                int32_t l = L[(size_t)j * max_n + (n - 1)];
                lb[n - 1] = (uint8_t)l;
                if (l != 0 && I[(size_t)j * max_n + (n - 1)] == 0) x |= 1u << (FLAG_SHIFT + n - 1);
            }
            int p = j - n;
            if (p >= 0 && p < rlen && L[(size_t)p * max_n + (n - 1)] != 0) {
                y |= 1u << (n - 1);
                if (I[(size_t)p * max_n + (n - 1)] == 0) y |= 1u << (6 + n - 1);
            }
        }
        if (j >= 1) x |= (uint32_t)ref[j - 1];
        // pre-decoded SHR candidates: the two highest periods with y's "inside an n-polymer" bit
        uint32_t dsc[2] = {0u, 0u};
        int nd = 0;
        for (int n = max_n; n >= 1; n--) {
            if (!((y >> (n - 1)) & 1u)) continue;
            if (nd < 2) {
                const uint32_t l = (uint32_t)L[(size_t)(j - n) * max_n + (n - 1)];
                dsc[nd] = make_shr_desc(n, ((y >> (6 + n - 1)) & 1u) != 0u, l, max_l);
            } else {
                dsc[1] |= DSC_MORE;
            }
            nd++;
        }
        if (dsc[1] != 0u) dsc[0] |= DSC_HAS2;
        if (((dsc[0] | dsc[1]) & DSC_BIGL) || (dsc[1] & DSC_MORE)) dsc[0] |= DSC_RARE;
        refw[4 * j] = x;
        refw[4 * j + 1] = y;
        refw[4 * j + 2] = dsc[0];
        refw[4 * j + 3] = dsc[1];
    }
}

}  // namespace npore
