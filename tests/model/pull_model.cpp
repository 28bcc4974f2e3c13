// tests/model/pull_model.cpp -- TEST INFRASTRUCTURE.
// Host-side, cell-by-cell execution of the *product's* recurrence (cell.hpp) on
// the *product's* packed inputs (tests/model/host_prep.hpp, the host twin of prep_kernels.hpp), sweeping anti-diagonals exactly as
// the gfx950 kernel does (same neighbour selection, same 8-row history ring,
// same traceback words), but one cell at a time.  The CPU tests compare its
// output with the oracle: this proves the pull reformulation, the carried
// run-start value and the word packing independently of the GPU mechanics
// (lane shifts, LDS, barriers), which only the -m gpu tests can exercise.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../npore_amd/csrc/cell.hpp"
#include "host_prep.hpp"

using namespace npore;

namespace {
constexpr int NS = 6;   // as in the kernel: row b overwrites row b-6 after it was read

struct ModelEnv {
    const float *sub_scores, *np_scores;
    int max_l;
    const uint8_t *refl_p;
    const uint32_t *refw_p;
    const float *hv[3];
    const uint32_t *hr;
    int W, slot;   // slot of the current row; row b-n is (slot - n) mod NS
    int refl_n;    // entries in refl_p / refw_p
    static constexpr bool LEN_ARITH = true;
    static constexpr bool MIN3 = true;      // (the model takes the kernel's route)
    float min3(float a, float b, float c) const { const float m = b < a ? b : a; return c < m ? c : m; }
    struct Tab { uint32_t hist6; };
    Tab step_tables(const StepInfo &st) const { return Tab{st.hist6}; }
    float sub(uint32_t seqw, uint32_t refx) const
    {
        const uint32_t s = seqw >> 29, r = refx & 7u;
        return (s < 5 && r < 5) ? sub_scores[s * 5 + r] : 0.0f;
    }
    float np_full(int n_idx, int a, int b, bool active) const
    {
        return active ? np_scores[((size_t)n_idx * (max_l + 1) + a) * (max_l + 1) + b] : 0.0f;
    }
    float np_small(uint32_t dsc, int q) const
    {
        const int n = (int)((dsc >> 2) & 7u), L = (int)((dsc >> 8) & 127u);
        const int call = L - 1 - q;
        if (n == 0) return huge_f();      // empty descriptor: never wins
        if (call < 0) return INF_F;
        const int row = L < max_l ? L : max_l - 1;          // np_score clamps the row to max_l - 1
        return np_scores[((size_t)(n - 1) * (max_l + 1) + row) * (max_l + 1) + call];
    }
    int clamp() const { return max_l - 1; }
    int refl(int j, int n_idx) const { return (j >= 0 && j < refl_n) ? refl_p[(size_t)j * 8 + n_idx] : 0; }
    uint32_t refy(int j) const { return (j >= 0 && j < refl_n) ? refw_p[4 * (size_t)j + 1] : 0u; }
    // history: band edges and everything outside the band hold "no candidate can come from here"
    HistCell h_at(int n, int col) const
    {
        if (col < 1 || col > W - 2) return hist_none();
        const size_t k = (size_t)((slot - n + NS) % NS) * W + col;
        return HistCell{hv[0][k], hv[1][k], hv[2][k], hr[k]};
    }
    static int period(uint32_t n4) { return (int)((n4 >> 2) & 7u); }   // tables are 8-periodic over the lanes
    HistCell h_shr(const Tab &t, uint32_t n4, int c) const
    {
        const int n = period(n4);
        return h_at(n, c - popc32(t.hist6 & ((1u << n) - 1u)));
    }
    uint32_t h_off(const Tab &, uint32_t n4) const { return n4; }
    HistCell h_shr_at(const Tab &t, uint32_t, uint32_t n4, int c) const { return h_shr(t, n4, c); }
    template <class... T> void pin(T &...) const {}
    template <class T> T opaque(T x) const { return x; }
    HistCell h_len(const Tab &t, uint32_t n4, int c) const
    {
        const int n = period(n4);
        return h_at(n, c + n - popc32(t.hist6 & ((1u << n) - 1u)));
    }
    uint32_t recip(const Tab &, uint32_t n4) const { return recip16(period(n4)); }
    int mer_shift(const Tab &, uint32_t n4) const { const int n = period(n4); return n ? 32 - 3 * n : 0; }
    uint32_t mer_mask(const Tab &, uint32_t n4) const { return (1u << (3 * period(n4))) - 1u; }
    bool any(bool x) const { return x; }
    bool any2(bool a, bool b) const { return a && b; }
};
}  // namespace

extern "C" int64_t pull_model_align(const uint8_t *full_ref, int64_t ref_len, const uint8_t *full_seq,
                                    int64_t seq_len, const char *cigar, int64_t cig_len,
                                    const float *sub_scores, const float *np_scores, int max_n, int max_l,
                                    float indel_start, float indel_extend, int max_b_rows, int r, char *out,
                                    int64_t out_cap, int32_t *status)
{
    *status = 0;
    ReadPath path;
    if (max_n > MAX_PERIOD || !build_path(cigar, cig_len, seq_len, ref_len, max_b_rows, path)) {
        *status = 32;
        return -1;
    }
    const int W = 2 * r + 1;
    int64_t out_len = 0;
    std::vector<int32_t> scratch;
    for (size_t k = 0; k + 1 < path.breaks.size(); k++) {
        const int64_t brk = path.breaks[k], nxt = path.breaks[k + 1];
        const int nrows = (int)(nxt - brk + 1);
        const int row0 = path.inss[brk], col0 = (int)(brk - path.inss[brk]);
        const int rowN = path.inss[nxt], colN = (int)(nxt - path.inss[nxt]);
        const int drows = rowN - row0, dcols = colN - col0;
        const int slen = (int)(std::min<int64_t>(rowN + 1, seq_len) - row0);
        const int rlen = (int)(std::min<int64_t>(colN + 1, ref_len) - col0);
        std::vector<uint32_t> seqw(drows + 1), refw(4 * (size_t)(dcols + 1));
        std::vector<uint8_t> refl(8 * (size_t)(dcols + 1));
        pack_chunk_words(full_seq + row0, slen, drows, full_ref + col0, rlen, dcols, max_n, max_l, seqw.data(),
                         refw.data(), refl.data(), scratch);

        std::vector<float> matv(W, 0.f), insv(W, 0.f), delv(W, 0.f), LMv(W, 0.f), TMv(W, 0.f);
        std::vector<int> matrun(W, 0), insrun(W, 0), delrun(W, 0), LMrun(W, 0), TMrun(W, 0);
        std::vector<float> hm((size_t)NS * W, 0.f), hl((size_t)NS * W, 0.f), hs((size_t)NS * W, 0.f);
        std::vector<uint32_t> hr((size_t)NS * W, 0u);
        std::vector<uint32_t> tb((size_t)nrows * W, 0u);
        std::vector<CellOut> cur(W);

        ModelEnv env{sub_scores, np_scores, max_l, refl.data(), refw.data(), {hm.data(), hl.data(), hs.data()}, hr.data(), W, 0, dcols + 1};
        for (int bl = 0; bl < nrows; bl++) {
            const int64_t b = brk + bl;
            StepInfo st;
            st.b_local = bl;
            st.init_f = (float)(100 * bl);
            st.ins_l = path.inss[b] - row0;
            st.del_l = bl - st.ins_l;
            st.r = r;
            st.drows = drows;
            st.dcols = dcols;
            st.indel_start = indel_start;
            st.indel_extend = indel_extend;
            st.hist6 = 0;
            for (int k = 0; k < MAX_PERIOD; k++)
                if (b - 1 - k >= brk && path.steps[b - 1 - k]) st.hist6 |= 1u << k;
            const int I = (bl > 0) ? path.steps[b - 1] : 0;
            env.slot = bl % NS;
            for (int c = 0; c < W; c++) {
                CellIn in;
                std::memset(&in, 0, sizeof in);
                in.c = c;
                const int tc = I ? c : c + 1, lc = I ? c - 1 : c;
                if (tc >= 0 && tc < W) { in.topM = matv[tc]; in.topI = insv[tc]; in.topIrun = insrun[tc]; }
                if (lc >= 0 && lc < W) { in.leftM = matv[lc]; in.leftD = delv[lc]; in.leftDrun = delrun[lc]; }
                in.diagM = I ? LMv[c] : TMv[c];
                in.diagMrun = I ? LMrun[c] : TMrun[c];
                const int i = st.ins_l + r - c, j = st.del_l - r + c;
                in.seqw = (i >= 0 && i <= drows) ? seqw[i] : SEQW_SENTINEL;
                in.refx = (j >= 0 && j <= dcols) ? refw[4 * (size_t)j] : REFW_SENTINEL;
                in.sc0 = (j >= 0 && j <= dcols) ? refw[4 * (size_t)j + 2] : 0u;
                in.sc1 = (j >= 0 && j <= dcols) ? refw[4 * (size_t)j + 3] : 0u;
                if (step_is_plain(st)) cell_update<true>(env, st, in, cur[c]);   // same dispatch as the kernel
                else cell_update<false>(env, st, in, cur[c]);
            }
            // commit the row: neighbour-of-neighbour values for the next diagonal, history, traceback
            std::vector<float> nLMv(W), nTMv(W);
            std::vector<int> nLMrun(W), nTMrun(W);
            for (int c = 0; c < W; c++) {
                const int tc = I ? c : c + 1, lc = I ? c - 1 : c;
                nTMv[c] = (tc >= 0 && tc < W) ? matv[tc] : 0.f;
                nTMrun[c] = (tc >= 0 && tc < W) ? matrun[tc] : 0;
                nLMv[c] = (lc >= 0 && lc < W) ? matv[lc] : 0.f;
                nLMrun[c] = (lc >= 0 && lc < W) ? matrun[lc] : 0;
            }
            for (int c = 0; c < W; c++) {
                matv[c] = cur[c].matv; insv[c] = cur[c].insv; delv[c] = cur[c].delv;
                matrun[c] = cur[c].matrun; insrun[c] = cur[c].insrun; delrun[c] = cur[c].delrun;
                LMv[c] = nLMv[c]; LMrun[c] = nLMrun[c]; TMv[c] = nTMv[c]; TMrun[c] = nTMrun[c];
                const size_t h = (size_t)env.slot * W + c;
                hm[h] = cur[c].matv; hl[h] = cur[c].lenstart; hs[h] = cur[c].shrstart;
                hr[h] = (uint32_t)cur[c].lenrun_h | ((uint32_t)cur[c].shrrun_h << 16);
                tb[(size_t)bl * W + c] = cur[c].tb;
            }
        }

        // traceback, reference src/aln.pyx:670-742
        int64_t a_row = rowN, a_col = colN;
        std::string aln;
        while (a_row > row0 || a_col > col0) {
            const int64_t bl = a_row + a_col - brk;
            if (a_row < row0 || a_col < col0 || bl < 0 || bl >= nrows) { *status |= 16; break; }
            const int64_t bc = (int64_t)path.inss[a_row + a_col] - a_row + r;
            if (bc < 0 || bc >= W) { *status |= 16; break; }
            const uint32_t w = tb[(size_t)bl * W + bc];
            const int typ = npore::tb_typ(w), run = npore::tb_run(w);
            if (run < 1) { *status |= 4; break; }
            if (typ == T_LEN || typ == T_INS) { aln.append(run, 'I'); a_row -= run; }
            else if (typ == T_SHR || typ == T_DEL) { aln.append(run, 'D'); a_col -= run; }
            else if (typ == T_MAT) {
                bool bad = false;
                for (int q = 0; q < run; q++) {
                    a_row--; a_col--;
                    if (a_row < row0 || a_col < col0) { bad = true; break; }
                    aln.push_back(full_ref[a_col] == full_seq[a_row] ? '=' : 'X');
                }
                if (bad) { *status |= 16; break; }
            } else { *status |= 8; break; }
        }
        if (out_len + (int64_t)aln.size() > out_cap) { *status |= 64; return -1; }
        for (size_t q = 0; q < aln.size(); q++) out[out_len + q] = aln[aln.size() - 1 - q];
        out_len += (int64_t)aln.size();
    }
    return out_len;
}

// Host-prepared arrays of one read, chunk after chunk, for word-by-word comparison with
// the device prep kernels.  Returns the number of chunks (or -1); *_n receive element counts.
extern "C" int64_t pull_model_prep(const uint8_t *full_ref, int64_t ref_len, const uint8_t *full_seq, int64_t seq_len,
                                   const char *cigar, int64_t cig_len, int max_n, int max_l, int max_b_rows,
                                   uint8_t *steps, int64_t *steps_n, int32_t *inss, int64_t *inss_n,
                                   int32_t *chunk_geom /* 7 per chunk: brk nrows row0 col0 drows dcols out_cap */,
                                   uint32_t *seqw, int64_t *seqw_n, uint32_t *refw, uint8_t *refl, int64_t *refw_n)
{
    ReadPath path;
    if (!build_path(cigar, cig_len, seq_len, ref_len, max_b_rows, path)) return -1;
    std::memcpy(steps, path.steps.data(), path.steps.size());
    *steps_n = (int64_t)path.steps.size();
    std::memcpy(inss, path.inss.data(), path.inss.size() * 4);
    *inss_n = (int64_t)path.inss.size();
    int64_t so = 0, ro = 0;
    std::vector<int32_t> scratch;
    int64_t nch = path.steps.empty() ? 0 : (int64_t)path.breaks.size() - 1;
    for (int64_t k = 0; k < nch; k++) {
        const int64_t brk = path.breaks[k], nxt = path.breaks[k + 1];
        const int row0 = path.inss[brk], col0 = (int)(brk - path.inss[brk]);
        const int drows = path.inss[nxt] - row0, dcols = (int)(nxt - path.inss[nxt]) - col0;
        const int slen = (int)(std::min<int64_t>((int64_t)row0 + drows + 1, seq_len) - row0);
        const int rlen = (int)(std::min<int64_t>((int64_t)col0 + dcols + 1, ref_len) - col0);
        int32_t *g = chunk_geom + 7 * k;
        g[0] = (int32_t)brk; g[1] = (int32_t)(nxt - brk + 1); g[2] = row0; g[3] = col0; g[4] = drows; g[5] = dcols;
        g[6] = drows + dcols;
        pack_chunk_words(full_seq + row0, slen, drows, full_ref + col0, rlen, dcols, max_n, max_l, seqw + so,
                         refw + 4 * ro, refl + 8 * ro, scratch);
        so += drows + 1;
        ro += dcols + 1;
    }
    *seqw_n = so;
    *refw_n = ro;
    return nch;
}

extern "C" void pull_model_np_info(const uint8_t *seq, int64_t len, int max_n, int max_l, int32_t *out)
{
    std::vector<int32_t> L((size_t)len * max_n + 1), I((size_t)len * max_n + 1);
    np_info_compute(seq, len, max_n, max_l, L.data(), I.data());
    for (int64_t p = 0; p < len; p++)
        for (int n = 0; n < max_n; n++) {
            out[(p * 2 + 0) * max_n + n] = L[p * max_n + n];
            out[(p * 2 + 1) * max_n + n] = I[p * max_n + n];
        }
}
