// annot_wave.hpp -- n-polymer annotation and word packing of a chunk slice by ONE wavefront, in registers.
//
// get_np_info (reference src/aln.pyx:179-251) on the chunk slices of src/aln.pyx:453-456, and the packed words of
// layout.hpp, for the batch path (annotate_wave_kernel: one wave per (chunk, sequence), no LDS, no barrier, no
// scratch memory -- it runs as well beside a fill kernel, which holds the CUs' LDS, as on an empty GPU).
//
// Formulation (tests/model/annot_wave_model.py is its CPU statement, checked against the oracle's literal loop):
// with e_n[p] = (seq[p] == seq[p+n]), kf(p) = run of e_n from p on, q = kf / n, a position is an eligible start ON ITS
// OWN when its base is not N, q >= 2 (l = q + 1 >= 3 repeats) and (q + 1) n > L_n2[p] n2 for every shorter period
// (final values of the SAME position: periods are done in ascending order) -- one ballot per window and period.  The
// (L, L_IDX) of period n then follow a stride-n recurrence inside a run of e_n:
//     base != N and q + 1 > max_l            -> (max_l, 0)          (starts with more than max_l repeats overwrite
//                                                                    each other in turn: src/aln.pyx:245-249)
//     e_n[p-n .. p-1] set and L(p-n) != 0    -> (L(p-n), L_IDX(p-n) + 1)
//     own(p)                                 -> (q + 1, 0)
// which a window of 64 positions resolves in closed form: the latest start at or before the lane with more than max_l
// repeats (an index computation), else what enters from the previous window through the (L, L_IDX) of ITS last n
// lanes (one cross-lane read), else the earliest set bit of the window's own ballot on the lane's phase (two shifts,
// a stride mask, count-trailing-zeros).  Look-ahead: the masks of the next window, and -- only where a run covers
// all of it -- further windows, up to (max_l + 2) n positions (beyond that every answer is the same).
// Windows are visited in order, the masks of window w + 1 are formed while window w is annotated.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "layout.hpp"
#include "prep_kernels.hpp"

namespace npore {

__device__ __forceinline__ int ctz64_or64(unsigned long long x) { return x ? __builtin_ctzll(x) : 64; }
__device__ __forceinline__ int clz64_or64(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }

// value of `v` in lane `src` (mod 64): ds_bpermute takes the byte address 4 * lane and ignores the bits above
__device__ __forceinline__ uint32_t lane_read(uint32_t v, int src)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v);
}

// bits at multiples of n
template <int n>
__host__ __device__ constexpr unsigned long long stride_mask()
{
    unsigned long long m = 0ull;
    for (int k = 0; k < 64; k += n) m |= 1ull << k;
    return m;
}

// x / n for 0 <= x < 4096 (run lengths inside a window and capped look-ahead)
template <int n>
__device__ __forceinline__ int div_small(int x)
{
    if constexpr (n == 1) return x;
    else if constexpr (n == 2) return x >> 1;
    else if constexpr (n == 4) return x >> 2;
    else return (int)(((uint32_t)x * (uint32_t)((65536 + n - 1) / n)) >> 16);      // exact for x < 13107 (cell.hpp recip16)
}

// One period of one window, in two parts.
// annot_forward: q = (run of e_n from the lane's position on) / n.  MC: e_n of the window's positions (bit = lane);
// F: ones of e_n from the next window's first position on.
template <int n>
__device__ __forceinline__ int annot_forward(int lane, unsigned long long MC, int F)
{
    const unsigned long long sh = MC >> lane;
    int kf = ctz64_or64(~sh);                                   // <= 64 - lane (zeros are shifted in)
    kf += (kf == 64 - lane) ? F : 0;
    return div_small<n>(kf);
}
// annot_resolve: B: ones of e_n at the END of the previous window (<= 64); nzmask / startable: the lanes that hold a
// base of the slice that is not N / ... and have q >= 2 (three or more repeats from here), as lane masks; mx: max over the
// shorter periods of L n2 at the lane's own position; carry: the previous window's result of lane 64 - n + lane % n;
// jdiv = lane / n.  Returns L | L_IDX << 8.
// (Only called when some lane of the window is startable or the previous window left a result: otherwise all is 0.)
template <int n>
__device__ __forceinline__ uint32_t annot_resolve(int lane, unsigned long long MC, int q, int B, unsigned long long nzmask,
                                                  unsigned long long startable, uint32_t mx, uint32_t carry, int jdiv, int max_l)
{
    const unsigned long long E = startable & __builtin_amdgcn_ballot_w64((uint32_t)((q + 1) * n) > mx);
    const int kb = clz64_or64(~((MC << (63 - lane)) << 1));      // ones ending at lane - 1, <= lane
    const int J = div_small<n>(kb);
    const int phi = lane - jdiv * n;
    const int jcap = max(max(max_l - q, 2 - q), 0);
    uint32_t res = 0u;
    {
        // earliest set bit of the window's own ballot among pos - j n, j <= J: position pos - J n to bit 0
        const unsigned long long y = ((E << (63 - lane)) >> (63 - J * n)) & stride_mask<n>();
        const int j = J - div_small<n>(ctz64_or64(y));
        res = y ? (uint32_t)(j + q + 1) | ((uint32_t)j << 8) : 0u;
    }
    const bool inherit = (kb == lane) & (B >= n - phi) & ((carry & 0xFFu) != 0u);
    res = inherit ? carry + ((uint32_t)(jdiv + 1) << 8) : res;
    const bool capped = ((nzmask >> lane) & 1ull) != 0ull && jcap <= J;
    res = capped ? (uint32_t)max_l | ((uint32_t)jcap << 8) : res;
    return res;
}

// ones of e_n from position `from` on, looked at for at most `cap` positions (rare: a run that covers a whole window)
__device__ __noinline__ int annot_far_ones(const uint8_t *g, int len, int n, int from, int cap)
{
    const int lane = threadIdx.x & 63;
    int total = 0;
    for (int base = from; base < len && total < cap; base += 64) {
        const int pos = base + lane;
        const bool e = pos + n < len && g[pos] == g[pos + n];
        const int t = ctz64_or64(~__builtin_amdgcn_ballot_w64(e));
        total += t;
        if (t != 64) break;
    }
    return total;
}

// What a slice's annotation is written as: the packed words of a read slice (seqw) or of a reference slice (refw / refl)
// for the fill kernel, the (L, L_IDX) values themselves in get_np_info()'s [position][2][max_n] int32 layout, or one byte
// plane per period (L | start flag << 7) for the region kernels.
enum AnnotMode { ANNOT_SEQW = 0, ANNOT_REFW = 1, ANNOT_RAW = 2, ANNOT_PLANES = 3 };

// the annotation of one slice g[0, len).  ANNOT_SEQW / ANNOT_REFW: words 0 ... span are written.  ANNOT_RAW / ANNOT_PLANES:
// positions w_from <= pos < span are written (the windows in front of w_from are the warm-up of a segment that does not
// start at its sequence's first base: see np_info_wave_kernel); raw: out32 + pos * 2 * max_n; planes: plane n - 1 at
// planes + (n - 1) * pstride.
template <int MODE, bool ALLN>
__device__ __forceinline__ void annotate_slice(const uint8_t *g, const int len, const int span, const int max_n_, const int max_l,
                                               uint32_t *seqw, uint4 *refw, uint2 *refl, const int w_from = 0, int32_t *out32 = nullptr,
                                               uint8_t *planes = nullptr, const int64_t pstride = 0)
{
    constexpr bool IS_REF = MODE == ANNOT_REFW;
    const int lane = threadIdx.x;
    const int max_n = ALLN ? MAX_PERIOD : max_n_;
    const int nwin = MODE <= ANNOT_REFW ? (span + 1 + 63) >> 6 : (span + 63) >> 6;      // words i = 0 ... span are written
    // windows that hold bases: a chunk slice ends with its words, a segment of a longer sequence (ANNOT_RAW / ANNOT_PLANES)
    // is followed by more of it, and its last window looks ahead into that like every other
    const int nwin_have = MODE <= ANNOT_REFW ? nwin : (len + 63) >> 6;

    // bases around every position of window v: own code c (6 past the slice), the six before it (7 in front of the
    // slice) and the six behind it as 3-bit fields (layout.hpp), and the window's indicator masks
    auto load_window = [&](int v, uint32_t &c, uint32_t &kp, uint32_t &kn, unsigned long long (&M)[MAX_PERIOD]) {
        const int base = v << 6, pos = base + lane;
        uint32_t b[13];
        unsigned long long inside = ~0ull;
        if (base >= 6 && base + 63 + 6 < len) {                 // (wave-uniform) every byte lies in the slice
#pragma unroll
            for (int t = 0; t < 13; t++) b[t] = g[pos - 6 + t];
        } else {
#pragma unroll
            for (int t = 0; t < 13; t++) {
                const int idx = pos - 6 + t;
                b[t] = idx < 0 ? 7u : idx >= len ? 6u : (uint32_t)g[idx];
            }
            inside = __builtin_amdgcn_ballot_w64(pos < len);    // (a base past the slice equals no base: code 6 on both sides)
        }
        c = b[6];
        kp = b[0] | (b[1] << 3) | (b[2] << 6) | (b[3] << 9) | (b[4] << 12) | (b[5] << 15);
        kn = b[6] | (b[7] << 3) | (b[8] << 6) | (b[9] << 9) | (b[10] << 12) | (b[11] << 15);
#pragma unroll
        for (int n = 1; n <= MAX_PERIOD; n++) M[n - 1] = n <= max_n ? (__builtin_amdgcn_ballot_w64(b[6] == b[6 + n]) & inside) : 0ull;
    };

    uint32_t cC, kpC, knC, cN = 6u, kpN = 0u, knN = 0u;
    unsigned long long MC[MAX_PERIOD], MN[MAX_PERIOD];
    load_window(0, cC, kpC, knC, MC);
    uint32_t Rp[MAX_PERIOD] = {0u, 0u, 0u, 0u, 0u, 0u};         // the previous window's results, per period
    int B[MAX_PERIOD] = {0, 0, 0, 0, 0, 0};                     // ones at the end of the previous window's masks
    uint32_t left = 0u;                                         // bit n-1: the previous window's last n lanes hold a result

    for (int w = 0; w < nwin; w++) {
        // (the lane number as a value the compiler cannot move out of the loop: what it derives from it per period --
        // lane / n, lane % n, cross-lane addresses, lane masks -- would otherwise be kept in two dozen registers for
        // the whole loop, and the kernel is held to 64)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int base = w << 6, pos = base + ln;
        if (w + 1 < nwin_have) load_window(w + 1, cN, kpN, knN, MN);
        else {
#pragma unroll
            for (int n = 0; n < MAX_PERIOD; n++) MN[n] = 0ull;
        }
        const unsigned long long nzmask = __builtin_amdgcn_ballot_w64((cC - 1u) < 4u);      // bases of the slice that are not N (codes 1 ... 4)
        uint32_t R[MAX_PERIOD];
        uint32_t mx = 0u;
        uint32_t left_now = 0u;
        auto one = [&](auto tag) __attribute__((always_inline)) {
            constexpr int n = decltype(tag)::value;
            R[n - 1] = 0u;
            if (n > max_n) return;
            const unsigned long long M = MC[n - 1];
            int F = ctz64_or64(~MN[n - 1]);
            if (F == 64 && (M >> 63)) F += annot_far_ones(g, len, n, base + 128, (max_l + 2) * n);      // (rare: a run over all of the next window)
            const int q = annot_forward<n>(ln, M, F);
            const unsigned long long startable = nzmask & __builtin_amdgcn_ballot_w64(q >= 2);
            // a position of the window can only get a result from a startable position of the window or through what
            // the previous window left in its last lanes
            if (startable != 0ull || ((left >> (n - 1)) & 1u)) {
                const int jdiv = div_small<n>(ln);
                const uint32_t carry = lane_read(Rp[n - 1], 64 - n + ln - jdiv * n);
                const uint32_t r = annot_resolve<n>(ln, M, q, B[n - 1], nzmask, startable, mx, carry, jdiv, max_l);
                R[n - 1] = r;
                mx = max(mx, (r & 0xFFu) * (uint32_t)n);
                left_now |= ((__builtin_amdgcn_ballot_w64(r != 0u) >> (64 - n)) != 0ull ? 1u : 0u) << (n - 1);
            }
            B[n - 1] = clz64_or64(~M);
        };
        one(std::integral_constant<int, 1>{});
        one(std::integral_constant<int, 2>{});
        one(std::integral_constant<int, 3>{});
        one(std::integral_constant<int, 4>{});
        one(std::integral_constant<int, 5>{});
        one(std::integral_constant<int, 6>{});
        left = left_now;

        // ---- the words of positions base ... base + 63 (layout.hpp).  Period n's result at position pos - n is this
        // window's lane - n, or the previous window's lane 64 - n + lane
        if constexpr (MODE == ANNOT_RAW) {
            if (pos >= w_from && pos < span) {
                int32_t *o = out32 + (size_t)pos * (2 * max_n);
                if (ALLN) {                                      // 12 consecutive int32 per position: three 16-byte stores
                    int4 *o4 = reinterpret_cast<int4 *>(o);
                    o4[0] = make_int4((int)(R[0] & 0xFFu), (int)(R[1] & 0xFFu), (int)(R[2] & 0xFFu), (int)(R[3] & 0xFFu));
                    o4[1] = make_int4((int)(R[4] & 0xFFu), (int)(R[5] & 0xFFu), (int)(R[0] >> 8), (int)(R[1] >> 8));
                    o4[2] = make_int4((int)(R[2] >> 8), (int)(R[3] >> 8), (int)(R[4] >> 8), (int)(R[5] >> 8));
                } else {
#pragma unroll
                    for (int n = 0; n < MAX_PERIOD; n++)
                        if (n < max_n) { o[n] = (int)(R[n] & 0xFFu); o[max_n + n] = (int)(R[n] >> 8); }
                }
            }
        } else if constexpr (MODE == ANNOT_PLANES) {
            if (pos >= w_from && pos < span) {
#pragma unroll
                for (int n = 0; n < MAX_PERIOD; n++)
                    if (n < max_n) planes[(size_t)n * pstride + pos] = (uint8_t)((R[n] & 0x7Fu) | ((R[n] - 1u) < 255u ? 128u : 0u));
            }
        } else if constexpr (!IS_REF) {
            uint32_t wd = kpC << MER_SHIFT;
#pragma unroll
            for (int n = 1; n <= MAX_PERIOD; n++) {
                const uint32_t S = lane_read(ln >= 64 - n ? Rp[n - 1] : R[n - 1], ln - n);
                wd |= (S != 0u ? 1u : 0u) << (FLAG_SHIFT + n - 1);
                wd |= ((S - 1u) < 255u ? 1u : 0u) << (n - 1);                  // L != 0 and L_IDX == 0
            }
            if (pos <= span) seqw[pos] = wd;
        } else {
            uint32_t x = knC << MER_SHIFT, y = 0u;
            uint32_t sl03 = 0u, sl45 = 0u;                                     // L of period n at position pos - n, byte n - 1
#pragma unroll
            for (int n = 1; n <= MAX_PERIOD; n++) {
                const uint32_t S = lane_read(ln >= 64 - n ? Rp[n - 1] : R[n - 1], ln - n);
                x |= ((R[n - 1] - 1u) < 255u ? 1u : 0u) << (FLAG_SHIFT + n - 1);
                y |= (S != 0u ? 1u : 0u) << (n - 1);
                y |= ((S - 1u) < 255u ? 1u : 0u) << (6 + n - 1);
                if (n <= 4) sl03 |= (S & 0xFFu) << (8 * (n - 1)); else sl45 |= (S & 0xFFu) << (8 * (n - 5));
            }
            if (pos >= 1) x |= kpC >> 15;                        // the cell's own reference base ref[j-1]
            const uint32_t l03 = (R[0] & 0xFFu) | ((R[1] & 0xFFu) << 8) | ((R[2] & 0xFFu) << 16) | ((R[3] & 0xFFu) << 24);
            const uint32_t l45 = (R[4] & 0xFFu) | ((R[5] & 0xFFu) << 8);
            uint32_t dsc0 = 0u, dsc1 = 0u;
            if (__builtin_amdgcn_ballot_w64((y & 63u) != 0u)) {
                // pre-decoded SHR candidates: the two highest periods flagged in y (layout.hpp)
                const unsigned long long sl = (unsigned long long)sl03 | ((unsigned long long)sl45 << 32);
                uint32_t yl = y & 63u;
                const int na = yl ? 32 - __builtin_clz(yl) : 0;
                if (na) {
                    dsc0 = make_shr_desc(na, ((y >> (6 + na - 1)) & 1u) != 0u, (uint32_t)(sl >> (8 * (na - 1))) & 0xFFu, max_l);
                    yl &= ~(1u << (na - 1));
                    const int nb = yl ? 32 - __builtin_clz(yl) : 0;
                    if (nb) {
                        dsc1 = make_shr_desc(nb, ((y >> (6 + nb - 1)) & 1u) != 0u, (uint32_t)(sl >> (8 * (nb - 1))) & 0xFFu, max_l);
                        yl &= ~(1u << (nb - 1));
                        if (yl) dsc1 |= DSC_MORE;
                    }
                }
                if (dsc1 != 0u) dsc0 |= DSC_HAS2;
                if (((dsc0 | dsc1) & DSC_BIGL) || (dsc1 & DSC_MORE)) dsc0 |= DSC_RARE;
            }
            if (pos <= span) {
                refw[pos] = make_uint4(x, y, dsc0, dsc1);
                refl[pos] = make_uint2(l03, l45);               // bytes 0..5 = L for n = 1..6 (0 past the slice)
            }
        }
        // next window
#pragma unroll
        for (int n = 0; n < MAX_PERIOD; n++) { Rp[n] = R[n]; MC[n] = MN[n]; }
        cC = cN; kpC = kpN; knC = knN;
    }
}

// One wave per (chunk, sequence).  ALLN: the context's max_n is MAX_PERIOD (the tool's default): no "n <= max_n" tests.
// Held to 64 vector registers: a wave then fits beside the four waves a fill kernel keeps on every SIMD (kernels.hpp).
template <bool ALLN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void annotate_wave_kernel(PrepParams p)
{
    const int k = (int)(blockIdx.x >> 1);
    const bool is_ref = blockIdx.x & 1;
    if (k >= p.counters[0]) return;
    const ChunkDesc d = p.descs[k];
    const int64_t rd = d.read_id;
    const int64_t T = is_ref ? p.ref_off[rd + 1] - p.ref_off[rd] : p.seq_off[rd + 1] - p.seq_off[rd];
    const int start = is_ref ? d.col0 : d.row0, span = is_ref ? d.dcols : d.drows;
    const int len = (int)(((int64_t)start + span + 1 < T ? (int64_t)start + span + 1 : T) - start);   // src/aln.pyx:453-454
    const uint8_t *g = (is_ref ? p.refs + p.ref_off[rd] : p.seqs + p.seq_off[rd]) + start;
    if (is_ref) annotate_slice<ANNOT_REFW, ALLN>(g, len, span, p.max_n, p.max_l, nullptr, p.refw + d.refw_off, p.refl + d.refw_off);
    else annotate_slice<ANNOT_SEQW, ALLN>(g, len, span, p.max_n, p.max_l, p.seqw + d.seqw_off, nullptr, nullptr);
}

// ---------------------------------------------------------------------------
// get_np_info() of WHOLE sequences of any length (the API, reference src/aln.pyx:179; genome-scale slices, src/bed.py:56-76)
// by the same wave-local formulation: a sequence is cut into segments of `seg` positions, one wave per segment.  The
// recurrence of a period reaches back at most (max_l + 1) n positions -- a chain starts at an own start or is re-started
// by every position with more than max_l repeats ahead of it -- and through the "longest period wins" rule a shorter
// period's values decide a longer one's starts, so a wave that begins `warm` = sum over n of (max_l + 2) n positions
// (rounded up to whole windows) in front of its segment with no state has the exact state where the segment begins
// (measured on adversarial sequences: 768 positions are needed at max_l = 100, 2 142 are given; tests/test_model_vs_oracle.py
// states it on the CPU).  Look-ahead needs no margin: the slice handed to annotate_slice runs to the sequence's end.
struct NpInfoParams {
    const uint8_t *seqs;       // base codes, sequences ("slices") back to back
    const int64_t *seq_off;    // [n_slices + 1]
    const int2 *work;          // per wave: (slice, segment within the slice)
    int n_work;
    int max_n, max_l;
    int seg, warm;             // multiples of 64
    int32_t *out32;            // ANNOT_RAW: [position][2][max_n], slices back to back
    uint8_t *planes;           // ANNOT_PLANES: slice k's planes at planes + max_n * seq_off[k], stride = its length
};

template <int MODE, bool ALLN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void np_info_wave_kernel(NpInfoParams p)
{
    if ((int)blockIdx.x >= p.n_work) return;
    const int2 wk = p.work[blockIdx.x];
    const int64_t off = p.seq_off[wk.x];
    const int64_t slen = p.seq_off[wk.x + 1] - off;
    const int64_t s0 = (int64_t)wk.y * p.seg;                      // the segment: positions [s0, s1) of the slice
    const int64_t s1 = s0 + p.seg < slen ? s0 + p.seg : slen;
    const int64_t g0 = s0 > p.warm ? s0 - p.warm : 0;              // where this wave starts (a window boundary of the slice)
    const uint8_t *g = p.seqs + off + g0;
    // (lengths relative to g0 fit 32 bits: a slice is shorter than 2^30)
    annotate_slice<MODE, ALLN>(g, (int)(slen - g0), (int)(s1 - g0), p.max_n, p.max_l, nullptr, nullptr, nullptr, (int)(s0 - g0),
                               MODE == ANNOT_RAW ? p.out32 + (size_t)(off + g0) * (2 * p.max_n) : nullptr,
                               MODE == ANNOT_PLANES ? p.planes + (size_t)p.max_n * off + g0 : nullptr, slen);
}

}  // namespace npore
