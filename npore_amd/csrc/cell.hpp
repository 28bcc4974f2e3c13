// cell.hpp -- the per-cell recurrence of the banded 5-state DP, written once and
// compiled both into the gfx950 kernels (kernels.hpp) and into the host-side
// cell-level model that the CPU tests run against the oracle
// (tests/model/pull_model.cpp).  It restates reference src/aln.pyx:481-667.
//
// Formulation.  The reference *pushes* LEN/SHR (lengthen / shorten an
// n-polymer) candidates from a cell to the cells n rows below / n columns to the
// right, which live n anti-diagonals ahead.  Here every cell *pulls* its <= 6
// LEN and <= 6 SHR candidates from the n = 6..1 previous anti-diagonals instead:
// the reference processes anti-diagonals in ascending order and only replaces a
// stored candidate by a strictly smaller one (src/aln.pyx:616,630,651,664), so
// scanning n = 6 -> 1 with strict '<' reproduces its tie-breaking exactly.
// "Continue" candidates read MAT.VAL at the start of the run, `run` cells back
// (src/aln.pyx:623-629, 657-663); that value is carried along with the LEN/SHR
// state (lenstart / shrstart) so no look-back is needed: a run is only ever
// started from, and extended through, interior in-chunk cells, which makes the
// reference's `a_row-run >= inss[brk]` / `b_runup_col < 2r` guards vacuous, and
// the first-row / first-column overrides (src/aln.pyx:596-599, 637-640) only
// matter through the RUN they leave behind (always failing those guards), which
// is stored as 0 in the history.
//
// SIMT shape.  A lane rarely has more than one or two live candidates, but which
// period n is live differs from lane to lane, so instead of twelve
// (n, LEN/SHR) blocks the candidates are consumed in passes in which every lane
// takes its own highest remaining n (per-lane n); a pass is skipped when no lane
// of the wave has a candidate left (wave-uniform `any`).  Everything that depends
// only on (n, anti-diagonal) -- where the source row sits in the history ring and
// how far the band moved -- is a per-step table indexed by the lane's n
// (Env::step_tables; one ds_bpermute per use on the device).
//
// History convention.  "No candidate can come from here" is stored in the history
// itself instead of being tested per candidate: records outside band columns
// 1..2r-1 (the band edges and HIST_PAD pad records either side of a row) are
// never written and hold MAT.VAL = +inf, and a cell that is not inside a LEN (SHR)
// run stores lenstart (shrstart) = +inf; an infinite candidate never wins a strict '<'.
//
// Arithmetic is IEEE fp32: one add per candidate, strict '<' compares, in the
// reference's order.  No FMA contraction is possible (adds only).
#pragma once
#include <stdint.h>

#include "layout.hpp"

namespace npore {

}  // namespace npore
// measurement builds only (scripts/ab_fill.py, step_stats.py): ablation switches, instruction pads, path counters
#if defined(NPORE_EXPERIMENTS)
#include "experiments.hpp"
#else
#define NPORE_COUNT(k) ((void)0)
namespace npore { namespace xp { constexpr bool NOLEN = false, NOSHR = false, NOPOLL = false, CHUNKMAJOR = false, NOASM = false, NOPRETEST = false, DBGMAT = false; constexpr int POLLSLEEP = 0, ANN = 0, PRIO = 0, ANNT = 1024, NOASM_ROLES = 0, ZEROLDS = 0; } }
#endif
namespace npore {

// Wave-uniform quantities of one anti-diagonal (b-row) of one chunk.
struct StepInfo {
    int b_local;     // 0-based b-row inside the chunk
    float init_f;    // (float)(100 * b_local): what LEN / SHR start from (src/aln.pyx:473,476)
    int ins_l;       // inss[b] - row0 : local row of the input path on this anti-diagonal
    int del_l;       // dels[b] - col0 = b_local - ins_l
    int r;
    int drows, dcols;
    uint32_t hist6;  // bit k: input-path step (b-1-k) -> (b-k) was an 'I'; so
                     // inss[b]-inss[b-n] = popcount(hist6 & ((1<<n)-1))
    float indel_start, indel_extend;
};

// Values a cell reads from its three neighbours (previous two anti-diagonals).
struct CellIn {
    float topM, topI;      // MAT.VAL / INS.VAL of (a_row-1, a_col)
    float leftM, leftD;    // MAT.VAL / DEL.VAL of (a_row, a_col-1)
    float diagM;           // MAT.VAL of (a_row-1, a_col-1)
    int topIrun, leftDrun; // INS.RUN / DEL.RUN of those
    int diagMrun;          // MAT.RUN of the diagonal cell if its MAT.TYP == MAT else 0
    uint32_t seqw, refx;
    uint32_t sc0, sc1;     // refw.z/.w: pre-decoded SHR candidates of this column (layout.hpp)
    int c;                 // band column 0..2r
};

struct CellOut {
    float matv, insv, delv;
    float lenstart, shrstart;  // MAT.VAL at the start of the current LEN / SHR run (+inf outside a run)
    int matrun;                // MAT.RUN if MAT.TYP == MAT else 0
    int insrun, delrun;
    int lenrun_h, shrrun_h;    // LEN.RUN / SHR.RUN as later "continue" moves may use them
    uint32_t tb;               // tb_word(MAT.TYP, MAT.RUN)
};

// What later LEN/SHR "start"/"continue" moves need from a finished cell.
struct alignas(16) HistCell {
    float matv;       // MAT.VAL
    float lenstart;   // MAT.VAL at the start of its LEN run, +inf if it is in none
    float shrstart;   // MAT.VAL at the start of its SHR run, +inf if it is in none
    uint32_t runs;    // LEN.RUN | SHR.RUN << 16
};

NPORE_HD float huge_f() { return __builtin_inff(); }
NPORE_HD HistCell hist_none() { return HistCell{huge_f(), huge_f(), huge_f(), 0u}; }

NPORE_HD int popc32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}
// 0-based index of the highest set bit; 32 if m == 0 (v_ffbh_u32 returns -1 for 0, and the
// plain instruction is wanted here, not a clz with a zero fix-up)
NPORE_HD int top_index(uint32_t m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int lead;
    asm("v_ffbh_u32 %0, %1" : "=v"(lead) : "v"(m));
    return 31 - lead;
#else
    return m ? 31 - __builtin_clz(m) : 32;
#endif
}
// keep the low `width & 31` bits (v_bfe_u32)
NPORE_HD uint32_t low_bits(uint32_t x, int width)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(x, 0u, (uint32_t)width);
#else
    return x & ((1u << (width & 31)) - 1u);
#endif
}

// run / n by a 16.16 reciprocal (RECIP16[n] = ceil(65536 / n)): exact for run < 13107; beyond
// that the estimate is q or q + 1 with q >= 2184, far above every length clamp of np_score, so
// the two are indistinguishable to the recurrence (checked exhaustively by the GPU tests).
NPORE_HD uint32_t recip16(int n) { return n ? (65536u + (uint32_t)n - 1u) / (uint32_t)n : 0u; }
NPORE_HD int div_recip(int run, uint32_t m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)((uint32_t)__umul24((unsigned)run, m) >> 16);   // run < 2^16, m <= 2^16: fits 32 bits
#else
    return (int)(((uint64_t)(uint32_t)run * m) >> 16);
#endif
}

// Env supplies (all const, all inlined):
//   float sub(uint32_t seqw, uint32_t refx)                    sub_scores[seq[i-1]][ref[j-1]] from the two words
//   float np_full(int n_idx, int a, int b, bool active)        np_scores[n_idx][a][b] (a, b already clamped)
//   float np_small(uint32_t dsc, int q)     np_score(L, -(q+1)) for a descriptor with L < NP_LT:
//                                           np_scores[n-1][L][L-1-q], or INF_F if that call length is < 0;
//                                           +infinity for the empty descriptor 0 (a column without a candidate)
//   int   clamp()                                              max_l - 1 (see np_score_index below)
//   int   refl(int j, int n_idx)                               L of local ref position j
//   uint32_t refy(int j)                                       refw[j].y, 0 outside the chunk's columns
//   Tab   step_tables(const StepInfo &)     whatever the lookups below need per anti-diagonal
//   HistCell h_shr(Tab, uint32_t n4, int c) what the cell (i, j-n) left behind: band column
//                                           c - (inss[b]-inss[b-n]) of anti-diagonal b-n   (n4 = 4n)
//   auto  h_off(Tab, uint32_t n4) / HistCell h_shr_at(Tab, off, uint32_t n4, int c)   the same in two steps: where
//                                           the record lies (a lane-table read on the device), then the record
//   T     opaque(T x)                       x; device: through an empty asm, so that what is computed from it stays
//                                           where it is written (not hoisted out of a rarely taken block)
//   void  pin(T &...)                       device: the values are complete here and nothing that produces them is
//                                           moved below (an empty asm with in/out operands); host: nothing
//   HistCell h_len(Tab, uint32_t n4, int c) likewise (i-n, j): column c + n - (inss[b]-inss[b-n])
//   uint32_t recip(Tab, uint32_t n4)        RECIP16[n]
//   int   mer_shift(Tab, uint32_t n4)       32 - 3n (0 for n = 0)
//   uint32_t mer_mask(Tab, uint32_t n4)     (1 << 3n) - 1
//   static constexpr bool LEN_ARITH         how the LEN filter forms its n-mer shift (see there)
//   static constexpr bool MIN3, float min3(a, b, c)   MAT by 3-way minima on the plain path (see there); min3 returns
//                                           one of its operands
//   bool  any(bool), any2(bool a, bool b)   wave-level "any lane" of x / of a && b (identity on the host)
// Lanes without a candidate call these with n4 = 0 (or, in the LEN filter, 4*33) and ignore the result.
// The table lookups are cross-lane reads on the device: call them where all lanes are active
// (never behind a per-lane `&&` / `?:`).

// np_score, reference src/aln.pyx:257-274, split into index formation and lookup.
// Callers pass max_l where the signature says max_n, so lengths clamp to max_l-1
// and the `n > max_n` test reads n > max_l.  Returns true if the score is the constant 100.
NPORE_HD bool np_score_index(int clampv, int n, int ref_np_len, int indel_len, int &a, int &call)
{
    call = ref_np_len + indel_len;
    // (n > max_l: the `n > max_n` test of src/aln.pyx:265 with max_l in max_n's place -- alive when max_l < max_n)
    const bool invalid = (ref_np_len <= 0) || (call < 0) || (n > clampv + 1);
    a = ref_np_len < 0 ? 0 : ref_np_len;
    if (a > clampv) a = clampv;
    if (call < 0) call = 0;
    if (call > clampv) call = clampv;
    return invalid;
}

// True when every band cell of this anti-diagonal is an ordinary one: inside the chunk
// rectangle, not on its first two rows / columns, and with all of its <= 6-back
// LEN/SHR source cells inside the rectangle too.  That is the case for all but the
// first and last ~2r anti-diagonals of a chunk, and lets the wave run the FAST
// instantiation below, which drops the first-row / first-column / rectangle selects.
NPORE_HD bool step_is_plain(const StepInfo &st)
{
    return (st.ins_l - st.r >= MAX_PERIOD) && (st.del_l - st.r >= MAX_PERIOD) &&
           (st.ins_l + st.r <= st.drows) && (st.del_l + st.r <= st.dcols);
}

// SHR candidate(s) straight from a column's descriptor(s) whose repeat count is < NP_LT (all lanes; wave-uniform
// precondition): the score comes from the LDS table at an address the descriptor already holds, plus the number of
// copies deleted so far (src/aln.pyx:642-667).  TWO = the column's second candidate in the same block.  On the GPU
// a candidate is a chain of three dependent LDS round trips (lane tables -> history record -> score); the block is
// written in phases -- all lane-table reads, all history reads, all score reads, then the compares in the
// reference's order -- with env.pin() after each, which on the device keeps the compiler from sinking a read to
// its first use (one candidate after the other: twice the chain).
template <bool FAST, bool TWO, class Env, class Tab>
NPORE_HD void shr_small(const Env &env, const Tab &tab, const CellIn &in, int j, bool act0, bool act1,
                        float &shrv, int &shrrun, float &shrstart)
{
    // the descriptor itself serves as the lane-table address: the tables repeat every 8 lanes and only
    // address bits 2-7 select a lane, so the flag bits above the period do not matter
    const uint32_t d0 = in.sc0, d1 = TWO ? in.sc1 : 0u;
    auto o0 = env.h_off(tab, d0);
    uint32_t m0 = env.recip(tab, d0);
    auto o1 = o0;
    uint32_t m1 = m0;
    if constexpr (TWO) {
        o1 = env.h_off(tab, d1);
        m1 = env.recip(tab, d1);
        env.pin(o0, m0, o1, m1);
    } else {
        env.pin(o0, m0);
    }
    HistCell h0 = env.h_shr_at(tab, o0, d0, in.c), h1 = h0;
    if constexpr (TWO) {
        h1 = env.h_shr_at(tab, o1, d1, in.c);
        env.pin(h0.matv, h0.shrstart, h0.runs, h1.matv, h1.shrstart, h1.runs);
    } else {
        env.pin(h0.matv, h0.shrstart, h0.runs);
    }
    const int n0 = (int)((d0 >> 2) & 7u), n1 = (int)((d1 >> 2) & 7u);
    const bool start0 = (d0 & DSC_START) != 0u, start1 = (d1 & DSC_START) != 0u;
    const float cstart0 = start0 ? h0.matv : h0.shrstart;             // :649 / :662
    const float cstart1 = start1 ? h1.matv : h1.shrstart;
    const int run0 = start0 ? 0 : (int)(h0.runs >> 16), run1 = start1 ? 0 : (int)(h1.runs >> 16);
    float s0 = env.np_small(d0, div_recip(run0, m0)), s1 = s0;        // indel = -(q + 1), :650 / :663
    if constexpr (TWO) {
        s1 = env.np_small(d1, div_recip(run1, m1));
        env.pin(s0, s1);
    }
    const float cand0 = cstart0 + s0;
    // (FAST: a lane without a candidate holds an empty descriptor, whose score is +infinity -- np_small -- so its
    // candidate never passes the compare; what the band-edge and out-of-band lanes pick up is never read)
    const bool take0 = (FAST || (act0 && j - n0 >= 0)) && cand0 < shrv;
    shrv = take0 ? cand0 : shrv;
    shrrun = take0 ? run0 + n0 : shrrun;                              // :654 / :667
    shrstart = take0 ? cstart0 : shrstart;
    if constexpr (TWO) {
        const float cand1 = cstart1 + s1;
        const bool take1 = (FAST || (act1 && j - n1 >= 0)) && cand1 < shrv;
        shrv = take1 ? cand1 : shrv;
        shrrun = take1 ? run1 + n1 : shrrun;
        shrstart = take1 ? cstart1 : shrstart;
    }
}

// One SHR candidate of period n (n4 = 4n; 0 = this cell has none), repeat count L of reference
// position j-n, any L (pull form of src/aln.pyx:642-667).
template <bool FAST, class Env, class Tab>
NPORE_HD void shr_generic(const Env &env, const Tab &tab, const CellIn &in, int j, uint32_t n4, int L, bool start,
                          bool act, float &shrv, int &shrrun, float &shrstart)
{
    const int n = (int)(n4 >> 2);
    const HistCell h = env.h_shr(tab, n4, in.c);
    const float cstart = start ? h.matv : h.shrstart;
    const int run = start ? 0 : (int)(h.runs >> 16);
    const int indel = -div_recip(run, env.recip(tab, n4)) - 1;
    int a, call;
    const bool inval = np_score_index(env.clamp(), n, L, indel, a, call);
    const bool ok = act && (FAST || j - n >= 0);
    const float score = env.np_full(n ? n - 1 : 0, a, call, ok);
    const float cand = cstart + (inval ? INF_F : score);
    const bool take = ok && cand < shrv;
    shrv = take ? cand : shrv;
    shrrun = take ? run + n : shrrun;
    shrstart = take ? cstart : shrstart;
}

// EDGES = false: the caller patches the two band-edge cells itself (kernels.hpp patches only the
// three values their one in-band neighbour reads).  MID = true: the caller guarantees 1 <= c <= 2r-1 for every
// cell it passes (a middle wave of a chunk holds band-interior columns only), which drops the column tests.
template <bool FAST, bool EDGES = true, bool MID = false, class Env>
NPORE_HD void cell_update(const Env &env, const StepInfo &st, const CellIn &in, CellOut &q)
{
    const int r2 = 2 * st.r;
    const int c = in.c;
    const int i = st.ins_l + st.r - c;   // local a_row
    const int j = st.del_l - st.r + c;   // local a_col
    const float init = st.init_f;                    // src/aln.pyx:473,476
    float insv, delv;
    int insrun, delrun;
    // ---- INS, src/aln.pyx:525-543 (branch-free: selects only)
    {
        const float v1 = in.topM + st.indel_start;
        const float v2 = in.topI + st.indel_extend;
        const bool ext = v2 < v1;
        const int erun = (!FAST && i == 1) ? 1 : in.topIrun + 1;
        const float v = ext ? v2 : v1;
        const int rr = ext ? erun : 1;
        insv = (!FAST && i == 0) ? (float)(100 * (j + 1)) : v;
        insrun = (!FAST && i == 0) ? j : rr;
    }
    // ---- DEL, src/aln.pyx:547-565
    {
        const float v1 = in.leftM + st.indel_start;
        const float v2 = in.leftD + st.indel_extend;
        const bool ext = v2 < v1;
        const int erun = (!FAST && j == 1) ? 1 : in.leftDrun + 1;
        const float v = ext ? v2 : v1;
        const int rr = ext ? erun : 1;
        delv = (!FAST && j == 0) ? (float)(100 * (i + 1)) : v;
        delrun = (!FAST && j == 0) ? i : rr;
    }
    float lenv = init, shrv = init, lenstart = huge_f(), shrstart = huge_f();
    int lenrun = 0, shrrun = 0;
    // candidate periods: LEN needs "ref position j starts an n-polymer" and
    // "read position i-n inside one"; SHR needs "ref position j-n inside one"
    const bool interior = (MID || ((c >= 1) && (c <= r2 - 1))) &&
                          (FAST || ((i >= 0) && (j >= 0) && (i <= st.drows) && (j <= st.dcols)));
    const uint32_t imask = interior ? 0xFFFFFFFFu : 0u;    // loop-invariant in the plain case
    uint32_t lm = ((in.refx & in.seqw & imask) >> FLAG_SHIFT) & 63u;
    if constexpr (xp::NOLEN) lm = 0u;
    // period of the column's first SHR candidate + summary bits
    const uint32_t sm = xp::NOSHR ? 0u : in.sc0 & imask & (DSC_N4 | DSC_HAS2 | DSC_RARE);

    // ---- LEN / SHR candidates (pull form of src/aln.pyx:601-633, 642-667)
    // SHR of a cell comes from X = (i, j-n) at band column c - dI; LEN from
    // X = (i-n, j) at band column c + (n - dI), dI = inss[b] - inss[b-n].
    // SHR candidates are dense inside reference n-polymers, LEN candidates are rare
    // (they also need the read to repeat the same unit), so each has its own loop.
    NPORE_COUNT(0);
    // (almost every anti-diagonal of a wave has an SHR candidate somewhere -- 94 % at r=100, 99.8 % at r=30 -- and two
    // thirds of them nothing else: the tests are ordered for that)
    {
        const auto tab = env.step_tables(st);
        // (a middle wave -- 64 band-interior columns -- has one practically always: no test)
        if ((FAST && MID) || env.any(sm != 0u)) {
            NPORE_COUNT(2);
            // the column's two highest periods come pre-decoded with the reference words
            // (evaluation order = the reference's: higher period first)
            const bool act = sm != 0u;
            if (!env.any(sm > DSC_N4)) {
                // no column of the wave has a second candidate or needs the generic path
                NPORE_COUNT(3);
                shr_small<FAST, false>(env, tab, in, j, act, false, shrv, shrrun, shrstart);
            } else if (!env.any(sm >= DSC_RARE)) {
                // (both in one block: the second candidate's history / table reads are independent of the first's
                // outcome and overlap with them)
                NPORE_COUNT(3);
                NPORE_COUNT(4);
                shr_small<FAST, true>(env, tab, in, j, act, (sm & DSC_HAS2) != 0u, shrv, shrrun, shrstart);
            } else {
                // a long n-polymer (L >= NP_LT) or three or more periods in one column somewhere in the wave
                const bool has2 = (sm & DSC_HAS2) != 0u, act2 = has2;
                shr_generic<FAST>(env, tab, in, j, in.sc0 & DSC_N4, (int)((in.sc0 >> 8) & 127u),
                                  (in.sc0 & DSC_START) != 0u, act, shrv, shrrun, shrstart);
                if (env.any(has2))
                    shr_generic<FAST>(env, tab, in, j, in.sc1 & DSC_N4, (int)((in.sc1 >> 8) & 127u),
                                      (in.sc1 & DSC_START) != 0u, act2, shrv, shrrun, shrstart);
                const bool more = act2 && (in.sc1 & DSC_MORE) != 0u;
                if (env.any(more)) {
                const uint32_t ry = env.refy(j);      // refw[j].y (layout.hpp); not carried through the lanes
                const uint32_t done = (1u << ((in.sc0 >> 2) & 7u)) | (1u << ((in.sc1 >> 2) & 7u));   // bit n
                uint32_t rest = more ? ((ry & 63u) << 1) & ~done : 0u;                                // bit n
                while (env.any(rest != 0u)) {
                    const bool a3 = rest != 0u;
                    const int n = a3 ? top_index(rest) : 0;
                    rest &= ~(1u << n);
                    const bool startf = ((ry >> ((6 + n - 1) & 31)) & 1u) != 0u;
                    const int L = env.refl(j - n, n ? n - 1 : 0);
                    shr_generic<FAST>(env, tab, in, j, (uint32_t)n << 2, L, startf, a3, shrv, shrrun, shrstart);
                }
                }
            }
        }

        while (env.any(lm != 0u)) {
            NPORE_COUNT(5);
            // (opaque: formed here, in the one anti-diagonal in three that gets this far, not hoisted in front of the loop)
            const uint32_t refm = env.opaque(in.refx) >> MER_SHIFT;
            const bool valid = lm != 0u;
            const int nm1 = top_index(lm);                 // period - 1 (32 if this lane has none left)
            lm = low_bits(lm, nm1);
            const uint32_t n4 = (uint32_t)(nm1 + 1) << 2;  // (4*33 where none: tables are 8-periodic)
            // match(), :606-607: the n most recent read bases against the next n reference bases.  The shift
            // count 32 - 3n either comes from the lane tables (two cross-lane reads, to be made BEFORE any per-lane
            // condition: they only work while every lane of the wave is executing) or is computed per lane; on the
            // GPU the second is faster for chunks of several waves (-1.5...2 % fill at r = 70...200) and slightly
            // slower for one wave per chunk (+0.4 % at r = 30), so the Env says which
            bool match;
            if constexpr (Env::LEN_ARITH) {
                const uint32_t sh = (uint32_t)(29 - 3 * nm1) & 31u;
                match = (((in.seqw >> sh) ^ refm) << sh) == 0u;
            } else {
                const uint32_t smer = in.seqw >> env.mer_shift(tab, n4);
                match = ((smer ^ refm) & env.mer_mask(tab, n4)) == 0u;
            }
            const bool inside = FAST || i - (nm1 + 1) >= 0;
            const bool good = valid && inside && match;
            if (!(FAST ? env.any2(valid, match) : env.any(good))) continue;
            NPORE_COUNT(6);
            const int n = nm1 + 1;
            const bool start = ((in.seqw >> (nm1 & 31)) & 1u) != 0u;
            const int L = env.refl(st.del_l - st.r + env.opaque(c), nm1 & 7);      // (j, formed here: see refm)
            const HistCell h = env.h_len(tab, n4, c);
            const float cstart = start ? h.matv : h.lenstart;                  // :614 / :628
            const int run = start ? 0 : (int)(h.runs & 0xFFFFu);
            const int indel = div_recip(run, env.recip(tab, n4)) + 1;          // :615 / :629
            int a, call;
            const bool inval = np_score_index(env.clamp(), n, L, indel, a, call);
            const float score = env.np_full(nm1 & 7, a, call, good);
            const float cand = cstart + (inval ? INF_F : score);
            const bool take = good && cand < lenv;
            lenv = take ? cand : lenv;
            lenrun = take ? run + n : lenrun;                                  // :619 / :633
            lenstart = take ? cstart : lenstart;
        }
    }

    // ---- MAT, src/aln.pyx:569-592 (selects; candidate order INS, LEN, DEL, SHR, strict '<')
    const bool diag_ok = FAST || ((i > 0) && (j > 0));
    const float vdiag = in.diagM + env.sub(in.seqw, in.refx);
    float v = diag_ok ? vdiag : delv + 100.0f;     // else-branch: "ensure val1 isn't chosen"
    const uint32_t tr_diag = tb_word(T_MAT, (uint32_t)(in.diagMrun + 1));
    uint32_t tr = diag_ok ? tr_diag : tb_word(T_MAT, 0u);
    bool t1 = false, t2 = false, t3 = false, t4 = false;
    if constexpr (FAST && Env::MIN3) {
        // The chain of strict '<' below picks the FIRST candidate, in the order MAT, INS, LEN, DEL, SHR, that attains
        // the minimum.  The same from two 3-way minima (which return one of their operands bit for bit: no NaNs
        // here, denormals are kept) and one equality test per candidate, applied last to first -- 2 selects fewer
        const float vmin = env.min3(env.min3(vdiag, insv, lenv), delv, shrv);
        tr = tb_word(T_SHR, (uint32_t)shrrun);
        tr = (delv == vmin) ? tb_word(T_DEL, (uint32_t)delrun) : tr;
        tr = (lenv == vmin) ? tb_word(T_LEN, (uint32_t)lenrun) : tr;
        tr = (insv == vmin) ? tb_word(T_INS, (uint32_t)insrun) : tr;
        tr = (vdiag == vmin) ? tr_diag : tr;
        v = vmin;
    } else {
        t1 = insv < v;
        v = t1 ? insv : v;
        tr = t1 ? tb_word(T_INS, (uint32_t)insrun) : tr;
        t2 = lenv < v;
        v = t2 ? lenv : v;
        tr = t2 ? tb_word(T_LEN, (uint32_t)lenrun) : tr;
        t3 = delv < v;
        v = t3 ? delv : v;
        tr = t3 ? tb_word(T_DEL, (uint32_t)delrun) : tr;
        t4 = shrv < v;
        v = t4 ? shrv : v;
        tr = t4 ? tb_word(T_SHR, (uint32_t)shrrun) : tr;
    }
    // "no INDEL state won": every other candidate carries a non-zero TYP, so the word itself says so (one vector
    // compare; or-ing the four compare masks costs three scalar instructions, which are the dearer ones here)
    static_assert(T_MAT == 0, "tr == tr_diag <=> MAT.TYP == MAT");
    const bool any_taken = FAST ? tr != tr_diag : (t1 || t2 || t3 || t4);
    // band edge, src/aln.pyx:502-507: all five states = 100*(b_row+1), TYP = MAT, RUN = 0
    const bool edge = EDGES && ((c == 0) || (c == r2));
    const bool inrect = FAST || ((i >= 0) && (j >= 0) && (i <= st.drows) && (j <= st.dcols));
    const float e = (float)(100 * (st.b_local + 1));
    q.matv = edge ? e : v;
    q.insv = edge ? e : insv;
    q.delv = edge ? e : delv;
    // MAT.RUN while MAT.TYP == MAT (no INDEL state won), else 0 -- from the compare masks, not from tr
    q.matrun = (edge || any_taken || !diag_ok) ? 0 : in.diagMrun + 1;
    q.insrun = edge ? 0 : insrun;
    q.delrun = edge ? 0 : delrun;
    // src/aln.pyx:596-599 / :637-640 leave RUN = j / i on the first row / column, which no continue
    // move can use: such a cell is "in no run" for its successors
    const bool no_len = !FAST && i == 0, no_shr = !FAST && j == 0;
    q.lenstart = no_len ? huge_f() : lenstart;
    q.shrstart = no_shr ? huge_f() : shrstart;
    q.lenrun_h = no_len ? 0 : lenrun;
    q.shrrun_h = no_shr ? 0 : shrrun;
    // cells outside the chunk rectangle are never read by cells inside it
    q.tb = (edge || !inrect) ? 0u : tr;
}

}  // namespace npore
