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
    int b_local;   // 0-based b-row inside the chunk
    int ins_l;     // inss[b] - row0 : local row of the input path on this anti-diagonal
    int del_l;     // dels[b] - col0 = b_local - ins_l
    int r;
    int drows, dcols;
    int dI[MAX_PERIOD + 1];  // dI[n] = inss[b] - inss[b-n], n = 1..6 (0 where b-n < chunk start)
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

// Env supplies (all const, all inlined):
//   float sub(uint32_t seq_base, uint32_t ref_base)            sub_scores[s][r]
//   float np(int n_idx, int ref_len_clamped, int call_len_clamped)  np_scores[n_idx][.][.]
//   int   clamp()                                              max_l - 1 (see np_score below)
//   int   refl(int j, int n_idx)                               L of local ref position j
//   float h_mat(int n, int col), h_lens(int n, int col), h_shrs(int n, int col)
//   uint32_t h_runs(int n, int col)      history of anti-diagonal b-n at band column col
//   bool  any(bool)                      wave-level "any lane" (identity on the host)
template <class Env>
NPORE_HD float np_score(const Env &env, int n, int ref_np_len, int indel_len)
{
    // reference src/aln.pyx:257-274; callers pass max_l where the signature says
    // max_n, so lengths clamp to max_l-1 and the `n > max_n` test is dead.
    if (ref_np_len <= 0) return 100.0f;
    if (ref_np_len + indel_len < 0) return 100.0f;
    int call = ref_np_len + indel_len;
    const int cl = env.clamp();
    if (ref_np_len > cl) ref_np_len = cl;
    if (call > cl) call = cl;
    return env.np(n - 1, ref_np_len, call);
}

template <class Env>
NPORE_HD void cell_update(const Env &env, const StepInfo &st, const CellIn &in, CellOut &o)
{
    const int c = in.c;
    const int i = st.ins_l + st.r - c;   // local a_row
    const int j = st.del_l - st.r + c;   // local a_col
    const bool first_row = (i == 0), first_col = (j == 0);

    // ---- INS, src/aln.pyx:525-543
    float insv;
    int insrun;
    if (first_row) {
        insv = (float)(100 * (j + 1));
        insrun = j;
    } else {
        const float v1 = in.topM + st.indel_start;
        const float v2 = in.topI + st.indel_extend;
        if (v2 < v1) { insv = v2; insrun = (i == 1) ? 1 : in.topIrun + 1; }
        else { insv = v1; insrun = 1; }
    }
    // ---- DEL, src/aln.pyx:547-565
    float delv;
    int delrun;
    if (first_col) {
        delv = (float)(100 * (i + 1));
        delrun = i;
    } else {
        const float v1 = in.leftM + st.indel_start;
        const float v2 = in.leftD + st.indel_extend;
        if (v2 < v1) { delv = v2; delrun = (j == 1) ? 1 : in.leftDrun + 1; }
        else { delv = v1; delrun = 1; }
    }

    // ---- LEN / SHR as seen by this cell (pull form of src/aln.pyx:601-633, 642-667)
    const float init = (float)(100 * st.b_local);    // src/aln.pyx:473,476
    float lenv = init, shrv = init, lenstart = 0.0f, shrstart = 0.0f;
    int lenrun = 0, shrrun = 0;

    const uint32_t len_flags = (in.refx >> 18) & (in.seqw >> 18) & 63u;  // ref starts n-polymer & read pos i-n inside one
    const uint32_t shr_flags = in.refy & 63u;                            // ref pos j-n inside an n-polymer
    const bool interior = (c >= 1) && (c <= 2 * st.r - 1) && (i >= 0) && (j >= 0) && (i <= st.drows) && (j <= st.dcols);
    if (env.any(interior && (len_flags | shr_flags))) {
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int n = MAX_PERIOD; n >= 1; n--) {
            const int dI = st.dI[n];
            // LEN from X = (i-n, j): band column c + (n - dI)
            {
                const int cx = c + (n - dI);
                bool ok = interior && ((len_flags >> (n - 1)) & 1u) && (i - n >= 0) && (cx <= 2 * st.r - 1);
                if (ok) {
                    const uint32_t smer = ((in.seqw & 0x3FFFFu) >> (3 * (MAX_PERIOD - n)));
                    const uint32_t rmer = in.refx & ((1u << (3 * n)) - 1u);
                    ok = (smer == rmer);                 // match(), src/aln.pyx:606-607
                }
                if (env.any(ok)) {
                    if (ok) {
                        const int L = env.refl(j, n - 1);
                        float cand, cstart;
                        int crun;
                        bool have = true;
                        if ((in.seqw >> (24 + n - 1)) & 1u) {          // start insertion :613-619
                            cstart = env.h_mat(n, cx);
                            cand = cstart + np_score(env, n, L, 1);
                            crun = n;
                        } else {                                         // continue :621-633
                            const int run = (int)(env.h_runs(n, cx) & 0xFFFFu);
                            have = run > 0;
                            cstart = env.h_lens(n, cx);
                            cand = cstart + np_score(env, n, L, run / n + 1);
                            crun = run + n;
                        }
                        if (have && cand < lenv) { lenv = cand; lenrun = crun; lenstart = cstart; }
                    }
                }
            }
            // SHR from X = (i, j-n): band column c - dI
            {
                const int cx = c - dI;
                const bool ok = interior && ((shr_flags >> (n - 1)) & 1u) && (j - n >= 0) && (cx >= 1);
                if (env.any(ok)) {
                    if (ok) {
                        const int L = env.refl(j - n, n - 1);
                        float cand, cstart;
                        int crun;
                        bool have = true;
                        if ((in.refy >> (6 + n - 1)) & 1u) {           // start deletion :648-654
                            cstart = env.h_mat(n, cx);
                            cand = cstart + np_score(env, n, L, -1);
                            crun = n;
                        } else {                                         // continue :656-667
                            const int run = (int)(env.h_runs(n, cx) >> 16);
                            have = run > 0;
                            cstart = env.h_shrs(n, cx);
                            cand = cstart + np_score(env, n, L, -(run / n) - 1);
                            crun = run + n;
                        }
                        if (have && cand < shrv) { shrv = cand; shrrun = crun; shrstart = cstart; }
                    }
                }
            }
        }
    }

    // ---- MAT, src/aln.pyx:569-592
    float v;
    int typ, run;
    if (i > 0 && j > 0) {
        run = in.diagMrun + 1;
        v = in.diagM + env.sub((in.seqw >> 15) & 7u, (in.refx >> 24) & 7u);
        typ = T_MAT;
    } else {
        v = delv + 100.0f;   // "ensure val1 isn't chosen"
        typ = T_MAT;
        run = 0;
    }
    if (insv < v) { v = insv; typ = T_INS; run = insrun; }
    if (lenv < v) { v = lenv; typ = T_LEN; run = lenrun; }
    if (delv < v) { v = delv; typ = T_DEL; run = delrun; }
    if (shrv < v) { v = shrv; typ = T_SHR; run = shrrun; }

    o.matv = v;
    o.insv = insv;
    o.delv = delv;
    o.matrun = (typ == T_MAT) ? run : 0;
    o.insrun = insrun;
    o.delrun = delrun;
    o.lenstart = lenstart;
    o.shrstart = shrstart;
    o.lenrun_h = first_row ? 0 : lenrun;   // src/aln.pyx:596-599 leaves RUN = j, never usable
    o.shrrun_h = first_col ? 0 : shrrun;   // src/aln.pyx:637-640 likewise
    o.tb = (uint32_t)typ | ((uint32_t)run << 3);

    // ---- band edge, src/aln.pyx:502-507 (all five states, TYP = MAT, RUN = 0)
    if (c == 0 || c == 2 * st.r) {
        const float e = (float)(100 * (st.b_local + 1));
        o.matv = e; o.insv = e; o.delv = e;
        o.matrun = 0; o.insrun = 0; o.delrun = 0;
        o.lenrun_h = 0; o.shrrun_h = 0;
        o.tb = 0;
    }
    // cells outside the chunk rectangle are never read by cells inside it
    if (!((i >= 0) && (j >= 0) && (i <= st.drows) && (j <= st.dcols))) o.tb = 0;
}

}  // namespace npore
