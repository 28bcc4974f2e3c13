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
// (n, LEN/SHR) blocks the candidates are consumed by a loop in which every lane
// takes its own highest remaining n (per-lane n), for LEN and SHR and for all NG
// cells of the lane at once, branch-free inside the loop so that the LDS reads
// of the 2*NG evaluations overlap.  The loop runs max-over-lanes(#candidates)
// times (wave-uniform `any`).
//
// Arithmetic is IEEE fp32: one add per candidate, strict '<' compares, in the
// reference's order.  No FMA contraction is possible (adds only).
#pragma once
#include <stdint.h>

#include "layout.hpp"

#if defined(__HIPCC__)
#define NPORE_HD __host__ __device__ __forceinline__
#else
#define NPORE_HD inline
#endif

namespace npore {

// Wave-uniform quantities of one anti-diagonal (b-row) of one chunk.
struct StepInfo {
    int b_local;     // 0-based b-row inside the chunk
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
    uint32_t seqw, refx, refy;
    uint32_t sc0, sc1;     // refw.z/.w: pre-decoded SHR candidates of this column (layout.hpp)
    int c;                 // band column 0..2r
};

struct CellOut {
    float matv, insv, delv;
    float lenstart, shrstart;  // MAT.VAL at the start of the current LEN / SHR run
    int matrun;                // MAT.RUN if MAT.TYP == MAT else 0
    int insrun, delrun;
    int lenrun_h, shrrun_h;    // LEN.RUN / SHR.RUN as later "continue" moves may use them
    uint32_t tb;               // MAT.TYP | MAT.RUN << 3
};

// What later LEN/SHR "start"/"continue" moves need from a finished cell.
struct alignas(16) HistCell {
    float matv;       // MAT.VAL
    float lenstart;   // MAT.VAL at the start of its LEN run
    float shrstart;   // MAT.VAL at the start of its SHR run
    uint32_t runs;    // LEN.RUN | SHR.RUN << 16 (0 where a continue must not happen)
};

NPORE_HD int popc32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}
NPORE_HD int top_bit(uint32_t m)   // 1-based index of the highest set bit, 0 if none
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 32 - __clz((int)m);
#else
    return m ? 32 - __builtin_clz(m) : 0;
#endif
}
// run / n for 0 <= run < 65536, 1 <= n <= 6.  On the device a float reciprocal
// plus a bias is exact on that domain (checked exhaustively by the GPU tests).
NPORE_HD int div_small(int run, int n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)((float)run * __builtin_amdgcn_rcpf((float)n) + 0.03f);   // v_rcp_f32: 1 ulp
#else
    return run / n;
#endif
}

// Env supplies (all const, all inlined):
//   float sub(uint32_t seq_base, uint32_t ref_base)            sub_scores[s][r]
//   template<int K> void np_many(const int (&n_idx)[K], const int (&a)[K], const int (&b)[K],
//                                const bool (&active)[K], float (&out)[K])
//                                     out[k] = np_scores[n_idx][a][b] (a, b already clamped)
//   float np_lds(int row, int call)       np_scores[row / 32][row % 32][call], row < 6*32, call < 64
//   int   clamp()                                              max_l - 1 (see np_score_index below)
//   int   refl(int j, int n_idx)                               L of local ref position j
//   HistCell h_cell(int n, int col)        what the cell at band column col of
//                                          anti-diagonal b-n left behind
//   bool  any(bool)                        wave-level "any lane" (identity on the host)

// np_score, reference src/aln.pyx:257-274, split into index formation and lookup.
// Callers pass max_l where the signature says max_n, so lengths clamp to max_l-1
// and the `n > max_n` test is dead.  Returns true if the score is the constant 100.
NPORE_HD bool np_score_index(int clampv, int ref_np_len, int indel_len, int &a, int &call)
{
    call = ref_np_len + indel_len;
    const bool invalid = (ref_np_len <= 0) || (call < 0);
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

// SHR candidate straight from a column descriptor whose repeat count is < 32 (all lanes;
// wave-uniform precondition): then the score row is in the LDS table, the call length
// L - k - 1 is < 32 too, and none of np_score's clamps can trigger -- only "call < 0 -> 100".
template <int NG, bool FAST, class Env>
NPORE_HD void shr_pass_small(const Env &env, const StepInfo &st, const CellIn (&in)[NG], const int (&jj)[NG],
                             const uint32_t (&dsc)[NG], const bool (&act)[NG],
                             float (&shrv)[NG], int (&shrrun)[NG], float (&shrstart)[NG])
{
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int g = 0; g < NG; g++) {
        const int nb = (int)(dsc[g] & 7u);
        const int n = nb ? nb : 1;
        const int dI = popc32(st.hist6 & ((1u << n) - 1u));
        const int cx = in[g].c - dI;
        const bool good = act[g] && (FAST || jj[g] - n >= 0) && (cx >= 1);
        const bool start = (dsc[g] & 8u) != 0u;
        const int L = (int)((dsc[g] >> 4) & 127u);
        const HistCell h = env.h_cell(n, cx);
        const float cstart = start ? h.matv : h.shrstart;             // :649 / :662
        const int run = start ? 0 : (int)(h.runs >> 16);
        const int call = L - (start ? 1 : div_small(run, n) + 1);     // L + indel, :650 / :663
        const int row = (int)((dsc[g] >> 12) & 255u);                 // (n-1)*32 + L
        const float score = env.np_lds(row, call < 0 ? 0 : call);
        const float cand = cstart + (call < 0 ? 100.0f : score);
        const bool take = good && (start || run > 0) && cand < shrv[g];
        shrv[g] = take ? cand : shrv[g];
        shrrun[g] = take ? run + n : shrrun[g];                        // :654 / :667
        shrstart[g] = take ? cstart : shrstart[g];
    }
}

// One SHR candidate per cell (pull form of src/aln.pyx:642-667): period nn[g] (>= 1),
// repeat count LL[g] of reference position j-n, start-vs-continue, and whether the cell
// has such a candidate at all.  X = (i, j-n) sits at band column c - dI of anti-diagonal b-n.
template <int NG, bool FAST, class Env>
NPORE_HD void shr_pass(const Env &env, const StepInfo &st, const CellIn (&in)[NG], const int (&jj)[NG],
                       const int (&nn)[NG], const int (&LL)[NG], const bool (&startf)[NG], const bool (&act)[NG],
                       float (&shrv)[NG], int (&shrrun)[NG], float (&shrstart)[NG])
{
    const int clampv = env.clamp();
    int nidx[NG], ta[NG], tb_[NG], crun[NG];
    bool ok[NG], inval[NG];
    float cstart[NG], score[NG];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int g = 0; g < NG; g++) {
        const int n = nn[g];
        const int dI = popc32(st.hist6 & ((1u << n) - 1u));
        const int cx = in[g].c - dI;
        const bool good = act[g] && (FAST || jj[g] - n >= 0) && (cx >= 1);
        // (lanes that are not `good` read some in-LDS garbage below and ignore it)
        const HistCell h = env.h_cell(n, cx);
        cstart[g] = startf[g] ? h.matv : h.shrstart;                  // :649 / :662
        const int run = startf[g] ? 0 : (int)(h.runs >> 16);
        const int indel = startf[g] ? -1 : -div_small(run, n) - 1;   // :650 / :663
        inval[g] = np_score_index(clampv, LL[g], indel, ta[g], tb_[g]);
        nidx[g] = n - 1;
        crun[g] = run + n;                                            // :654 / :667
        ok[g] = good && (startf[g] || run > 0);
    }
    env.template np_many<NG>(nidx, ta, tb_, ok, score);
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int g = 0; g < NG; g++) {
        const float cand = cstart[g] + (inval[g] ? 100.0f : score[g]);
        const bool take = ok[g] && cand < shrv[g];
        shrv[g] = take ? cand : shrv[g];
        shrrun[g] = take ? crun[g] : shrrun[g];
        shrstart[g] = take ? cstart[g] : shrstart[g];
    }
}

// EDGES = false: the caller patches the two band-edge cells itself (kernels.hpp patches only the
// three values their one in-band neighbour reads).
template <int NG, bool FAST, bool EDGES = true, class Env>
NPORE_HD void cells_update(const Env &env, const StepInfo &st, const CellIn (&in)[NG], CellOut (&o)[NG])
{
    const int r2 = 2 * st.r;
    float insv[NG], delv[NG], lenv[NG], shrv[NG], lenstart[NG], shrstart[NG];
    int insrun[NG], delrun[NG], lenrun[NG], shrrun[NG], ii[NG], jj[NG];
    uint32_t lm[NG], sm[NG];
    const float init = (float)(100 * st.b_local);    // src/aln.pyx:473,476
    uint32_t pend = 0;

#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int g = 0; g < NG; g++) {
        const int c = in[g].c;
        const int i = st.ins_l + st.r - c;   // local a_row
        const int j = st.del_l - st.r + c;   // local a_col
        ii[g] = i;
        jj[g] = j;
        // ---- INS, src/aln.pyx:525-543 (branch-free: selects only)
        {
            const float v1 = in[g].topM + st.indel_start;
            const float v2 = in[g].topI + st.indel_extend;
            const bool ext = v2 < v1;
            const int erun = (!FAST && i == 1) ? 1 : in[g].topIrun + 1;
            const float v = ext ? v2 : v1;
            const int rr = ext ? erun : 1;
            insv[g] = (!FAST && i == 0) ? (float)(100 * (j + 1)) : v;
            insrun[g] = (!FAST && i == 0) ? j : rr;
        }
        // ---- DEL, src/aln.pyx:547-565
        {
            const float v1 = in[g].leftM + st.indel_start;
            const float v2 = in[g].leftD + st.indel_extend;
            const bool ext = v2 < v1;
            const int erun = (!FAST && j == 1) ? 1 : in[g].leftDrun + 1;
            const float v = ext ? v2 : v1;
            const int rr = ext ? erun : 1;
            delv[g] = (!FAST && j == 0) ? (float)(100 * (i + 1)) : v;
            delrun[g] = (!FAST && j == 0) ? i : rr;
        }
        lenv[g] = shrv[g] = init;
        lenstart[g] = shrstart[g] = 0.0f;
        lenrun[g] = shrrun[g] = 0;
        // candidate periods: LEN needs "ref position j starts an n-polymer" and
        // "read position i-n inside one"; SHR needs "ref position j-n inside one"
        const bool interior = (c >= 1) && (c <= r2 - 1) &&
                              (FAST || ((i >= 0) && (j >= 0) && (i <= st.drows) && (j <= st.dcols)));
        const uint32_t imask = interior ? 0xFFFFFFFFu : 0u;    // loop-invariant in the plain case
        lm[g] = ((in[g].refx & in[g].seqw & imask) >> 18) & 63u;
        sm[g] = in[g].sc0 & imask & 7u;                 // period of the column's first SHR candidate (0 = none)
        pend |= lm[g] | sm[g];
    }

    // ---- LEN / SHR candidates (pull form of src/aln.pyx:601-633, 642-667)
    // SHR of cell g comes from X = (i, j-n) at band column c - dI; LEN from
    // X = (i-n, j) at band column c + (n - dI), dI = inss[b] - inss[b-n].
    // SHR candidates are dense inside reference n-polymers, LEN candidates are rare
    // (they also need the read to repeat the same unit), so each has its own loop.
    const int clampv = env.clamp();
    (void)clampv;
    if (env.any(pend != 0u)) {
        uint32_t pendS = 0u, pendL = 0u;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int g = 0; g < NG; g++) { pendS |= sm[g]; pendL |= lm[g]; }

        if (env.any(pendS != 0u)) {
            // the column's two highest periods come pre-decoded with the reference words
            // (evaluation order = the reference's: higher period first)
            int nn[NG], LL[NG];
            uint32_t dsc[NG];
            bool startf[NG], act[NG], more = false, second = false, bigL = false;
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int g = 0; g < NG; g++) {
                act[g] = sm[g] != 0u;
                dsc[g] = in[g].sc0;
                second |= act[g] && (in[g].sc1 & 7u) != 0u;
                more |= act[g] && (in[g].sc1 & 0x800u) != 0u;
                bigL |= act[g] && (((in[g].sc0 | in[g].sc1) >> 20) & 1u) != 0u;
            }
            const bool small_l = !env.any(bigL);     // wave-uniform: every live descriptor has L < 32
            auto decode = [&]() {
#if defined(__HIPCC__)
#pragma unroll
#endif
                for (int g = 0; g < NG; g++) {
                    nn[g] = act[g] ? (int)(dsc[g] & 7u) : 1;
                    startf[g] = (dsc[g] & 8u) != 0u;
                    LL[g] = (int)((dsc[g] >> 4) & 127u);
                }
            };
            if (small_l) shr_pass_small<NG, FAST>(env, st, in, jj, dsc, act, shrv, shrrun, shrstart);
            else { decode(); shr_pass<NG, FAST>(env, st, in, jj, nn, LL, startf, act, shrv, shrrun, shrstart); }
            if (env.any(second)) {
#if defined(__HIPCC__)
#pragma unroll
#endif
                for (int g = 0; g < NG; g++) {
                    dsc[g] = in[g].sc1;
                    act[g] = (sm[g] != 0u) && (dsc[g] & 7u) != 0u;
                }
                if (small_l) shr_pass_small<NG, FAST>(env, st, in, jj, dsc, act, shrv, shrrun, shrstart);
                else { decode(); shr_pass<NG, FAST>(env, st, in, jj, nn, LL, startf, act, shrv, shrrun, shrstart); }
            }
            if (env.any(more)) {     // rare: three or more periods in one column -> decode the rest generically
                uint32_t rest[NG], pr = 0u;
#if defined(__HIPCC__)
#pragma unroll
#endif
                for (int g = 0; g < NG; g++) {
                    const uint32_t done = (1u << ((in[g].sc0 & 7u) - 1u)) | (1u << ((in[g].sc1 & 7u) - 1u));
                    rest[g] = (sm[g] != 0u && (in[g].sc1 & 0x800u)) ? (in[g].refy & 63u & ~done) : 0u;
                    pr |= rest[g];
                }
                while (env.any(pr != 0u)) {
                    pr = 0u;
#if defined(__HIPCC__)
#pragma unroll
#endif
                    for (int g = 0; g < NG; g++) {
                        const int nb = top_bit(rest[g]);
                        act[g] = nb != 0;
                        nn[g] = nb ? nb : 1;
                        rest[g] &= ~(1u << (nn[g] - 1));
                        pr |= rest[g];
                        startf[g] = ((in[g].refy >> (6 + nn[g] - 1)) & 1u) != 0u;
                        LL[g] = env.refl(jj[g] - nn[g], nn[g] - 1);
                    }
                    shr_pass<NG, FAST>(env, st, in, jj, nn, LL, startf, act, shrv, shrrun, shrstart);
                }
            }
        }

        while (env.any(pendL != 0u)) {
            pendL = 0u;
            int nn[NG], cxx[NG];
            bool good[NG], anygood = false;
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int g = 0; g < NG; g++) {
                const int c = in[g].c;
                const int nb = top_bit(lm[g]);
                const int n = nb ? nb : 1;
                lm[g] &= ~(1u << (n - 1));
                pendL |= lm[g];
                const int dI = popc32(st.hist6 & ((1u << n) - 1u));
                const int cx = c + (n - dI);
                const uint32_t smer = (in[g].seqw & 0x3FFFFu) >> (3 * (MAX_PERIOD - n));
                const uint32_t rmer = in[g].refx & ((1u << (3 * n)) - 1u);
                good[g] = (nb != 0) && (FAST || ii[g] - n >= 0) && (cx <= r2 - 1) && (smer == rmer);   // match(), :606-607
                nn[g] = n;
                cxx[g] = cx;
                anygood |= good[g];
            }
            if (!env.any(anygood)) continue;
            int nidx[NG], ta[NG], tb_[NG], crun[NG];
            bool ok[NG], inval[NG];
            float cstart[NG], score[NG];
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int g = 0; g < NG; g++) {
                const int n = nn[g];
                const bool start = ((in[g].seqw >> (24 + n - 1)) & 1u) != 0u;
                const int L = env.refl(jj[g], n - 1);
                const HistCell h = env.h_cell(n, cxx[g]);
                cstart[g] = start ? h.matv : h.lenstart;                  // :614 / :628
                const int run = start ? 0 : (int)(h.runs & 0xFFFFu);
                const int indel = start ? 1 : div_small(run, n) + 1;     // :615 / :629
                inval[g] = np_score_index(clampv, L, indel, ta[g], tb_[g]);
                nidx[g] = n - 1;
                crun[g] = run + n;                                        // :619 / :633
                ok[g] = good[g] && (start || run > 0);
            }
            env.template np_many<NG>(nidx, ta, tb_, ok, score);
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int g = 0; g < NG; g++) {
                const float cand = cstart[g] + (inval[g] ? 100.0f : score[g]);
                const bool take = ok[g] && cand < lenv[g];
                lenv[g] = take ? cand : lenv[g];
                lenrun[g] = take ? crun[g] : lenrun[g];
                lenstart[g] = take ? cstart[g] : lenstart[g];
            }
        }
    }

#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int g = 0; g < NG; g++) {
        const int c = in[g].c, i = ii[g], j = jj[g];
        // ---- MAT, src/aln.pyx:569-592 (selects; candidate order INS, LEN, DEL, SHR, strict '<')
        const bool diag_ok = FAST || ((i > 0) && (j > 0));
        const float vdiag = in[g].diagM + env.sub((in[g].seqw >> 15) & 7u, (in[g].refx >> 24) & 7u);
        float v = diag_ok ? vdiag : delv[g] + 100.0f;     // else-branch: "ensure val1 isn't chosen"
        uint32_t tr = diag_ok ? ((uint32_t)T_MAT | ((uint32_t)(in[g].diagMrun + 1) << 3)) : (uint32_t)T_MAT;  // typ | run<<3
        bool any_taken;
        {
            const bool t1 = insv[g] < v;
            v = t1 ? insv[g] : v;
            tr = t1 ? ((uint32_t)T_INS | ((uint32_t)insrun[g] << 3)) : tr;
            const bool t2 = lenv[g] < v;
            v = t2 ? lenv[g] : v;
            tr = t2 ? ((uint32_t)T_LEN | ((uint32_t)lenrun[g] << 3)) : tr;
            const bool t3 = delv[g] < v;
            v = t3 ? delv[g] : v;
            tr = t3 ? ((uint32_t)T_DEL | ((uint32_t)delrun[g] << 3)) : tr;
            const bool t4 = shrv[g] < v;
            v = t4 ? shrv[g] : v;
            tr = t4 ? ((uint32_t)T_SHR | ((uint32_t)shrrun[g] << 3)) : tr;
            any_taken = t1 || t2 || t3 || t4;
        }
        // band edge, src/aln.pyx:502-507: all five states = 100*(b_row+1), TYP = MAT, RUN = 0
        const bool edge = EDGES && ((c == 0) || (c == r2));
        const bool inrect = FAST || ((i >= 0) && (j >= 0) && (i <= st.drows) && (j <= st.dcols));
        const float e = (float)(100 * (st.b_local + 1));
        CellOut &q = o[g];
        q.matv = edge ? e : v;
        q.insv = edge ? e : insv[g];
        q.delv = edge ? e : delv[g];
        // MAT.RUN while MAT.TYP == MAT (no INDEL state won), else 0 -- from the compare masks, not from tr
        q.matrun = (edge || any_taken || !diag_ok) ? 0 : in[g].diagMrun + 1;
        q.insrun = edge ? 0 : insrun[g];
        q.delrun = edge ? 0 : delrun[g];
        q.lenstart = lenstart[g];
        q.shrstart = shrstart[g];
        q.lenrun_h = (edge || (!FAST && i == 0)) ? 0 : lenrun[g];   // src/aln.pyx:596-599 leaves RUN = j, never usable
        q.shrrun_h = (edge || (!FAST && j == 0)) ? 0 : shrrun[g];   // src/aln.pyx:637-640 likewise
        // cells outside the chunk rectangle are never read by cells inside it
        q.tb = (edge || !inrect) ? 0u : tr;
    }
}

}  // namespace npore
