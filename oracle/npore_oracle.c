/*
 * oracle/npore_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded, literal restatement of the nPoRe per-read
 * realignment DP.  It exists so that tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py can check / time the HIP product path against
 * an independent CPU implementation.  Nothing under npore_amd/ may include,
 * link, import or execute this file; the product path fails loudly when the
 * HIP library is missing and never falls back to this code.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file
 * against golden vectors produced by importing the reference's own compiled
 * Cython module in the build container (tests/golden/make_golden.py): the
 * reference's docstring example, its test/get_np_info.py and test/align.py
 * cases, its test/data reads + npore_realigned.sam golden output, and seeded
 * synthetic reads over several (r, max_b_rows) settings.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference checkout).  The statement order inside the fill loop is kept
 * exactly as in the reference because tie-breaking is by evaluation order
 * with strict '<' on IEEE fp32 values.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define T_MAT 0
#define T_INS 1
#define T_LEN 2
#define T_DEL 3
#define T_SHR 4
#define NTYPS 5
#define D_VAL 0
#define D_TYP 1
#define D_RUN 2
#define NDIMS 3
#define ORC_INF 100 /* src/aln.pyx:428 */

/* status bits reported per read (the reference prints and truncates instead,
 * src/aln.pyx:689-716,737-739) */
#define ORC_ERR_ROW_NEG 1
#define ORC_ERR_COL_NEG 2
#define ORC_ERR_RUN_LT1 4
#define ORC_ERR_BAD_TYPE 8
#define ORC_ERR_OUT_OF_CHUNK 16 /* reference: undefined behaviour (bounds checks off) */
#define ORC_ERR_BAD_INPUT 32
#define ORC_ERR_OUT_CAP 64

/* ------------------------------------------------------------------------
 * get_np_info: src/aln.pyx:179-251.  out is int32 [len][2][max_n], index
 * [pos][0=L,1=L_IDX][n-1]; zeroed here like np.zeros at src/aln.pyx:209.
 * ---------------------------------------------------------------------- */
void npore_oracle_get_np_info(const uint8_t *seq, int64_t seq_len, int max_n,
                              int max_l, int32_t *np_info)
{
    memset(np_info, 0, (size_t)seq_len * 2 * (size_t)max_n * sizeof(int32_t));
#define NPI(pos, which, nidx) np_info[((pos) * 2 + (which)) * (int64_t)max_n + (nidx)]
    for (int64_t seq_idx = 0; seq_idx < seq_len; seq_idx++) {
        if (!seq[seq_idx]) continue;                       /* :221-222 */
        for (int n = 1; n <= max_n; n++) {                 /* :224 */
            int n_idx = n - 1;
            int l = 0;
            int64_t seq_ptr = seq_idx;
            while (seq_ptr + n < seq_len && seq[seq_ptr] == seq[seq_ptr + n]) { /* :230 */
                seq_ptr += 1;
                if ((seq_ptr - seq_idx) % n == 0) l += 1;
            }
            if (l) l += 1;                                 /* :234 */
            if (l > 2) {                                   /* :237 */
                int longest = 1;
                for (int n2 = 1; n2 < n; n2++)             /* :240-242 */
                    if (l * n <= NPI(seq_idx, 0, n2 - 1) * n2) longest = 0;
                for (int l_idx = 0; l_idx < l; l_idx++) {  /* :245-249 */
                    int64_t pos = seq_idx + (int64_t)l_idx * n;
                    if (longest && l > NPI(pos, 0, n_idx)) {
                        NPI(pos, 0, n_idx) = max_l < l ? max_l : l;
                        NPI(pos, 1, n_idx) = l_idx;
                    }
                }
            }
        }
    }
#undef NPI
}

/* np_score: src/aln.pyx:257-274.  `clamp` is what the reference calls max_n in
 * the signature; every caller passes max_l (src/aln.pyx:615,629,650,663). */
static float orc_np_score(int n, int ref_np_len, int indel_len,
                          const float *np_scores, int np_dim, int clamp)
{
    if (ref_np_len <= 0) return 100;
    else if (ref_np_len + indel_len < 0) return 100;
    else if (n < 1 || n > clamp) return 100;
    int call_np_len = ref_np_len + indel_len;
    if (ref_np_len > clamp - 1) ref_np_len = clamp - 1;
    if (call_np_len > clamp - 1) call_np_len = clamp - 1;
    return np_scores[((int64_t)(n - 1) * np_dim + ref_np_len) * np_dim + call_np_len];
}

/* match: src/aln.pyx:364-372 on two Python-style clipped slices
 * A = a[a0 : a0+n] of an array of length alen, B likewise. */
static int orc_match(const uint8_t *a, int64_t alen, int64_t a0,
                     const uint8_t *b, int64_t blen, int64_t b0, int n)
{
    int64_t a1 = a0 + n, b1 = b0 + n;
    if (a0 > alen) a0 = alen;
    if (a1 > alen) a1 = alen;
    if (b0 > blen) b0 = blen;
    if (b1 > blen) b1 = blen;
    if (a1 - a0 != b1 - b0) return 0;
    for (int64_t i = 0; i < a1 - a0; i++)
        if (a[a0 + i] != b[b0 + i]) return 0;
    return 1;
}

/* ------------------------------------------------------------------------
 * align: src/aln.pyx:379-787 (verbose printer 744-785 omitted).
 * cigar: expanded ops over "=XMID" (S/H already stripped by the caller,
 * src/bam.pyx:59).  np_scores is f32 [max_n][max_l+1][max_l+1].
 * Returns the length written to out (>=0), or -1 on bad input / capacity.
 * ---------------------------------------------------------------------- */
int64_t npore_oracle_align(const uint8_t *full_ref, int64_t ref_len,
                           const uint8_t *full_seq, int64_t seq_len,
                           const char *cigar_in, int64_t cigar_in_len,
                           const float *sub_scores, const float *np_scores,
                           int max_n, int max_l, float indel_start,
                           float indel_extend, int max_b_rows, int r,
                           char *out, int64_t out_cap, int32_t *status)
{
    const int np_dim = max_l + 1;
    *status = 0;
    if (max_b_rows < 2 || r < 1 || max_n < 1) { *status = ORC_ERR_BAD_INPUT; return -1; }

    /* src/aln.pyx:386  X,=,M -> "DI" */
    int64_t cig_len = 0;
    for (int64_t i = 0; i < cigar_in_len; i++) {
        char c = cigar_in[i];
        if (c == 'X' || c == '=' || c == 'M') cig_len += 2;
        else if (c == 'I' || c == 'D') cig_len += 1;
        else { *status = ORC_ERR_BAD_INPUT; return -1; }
    }
    char *cigar = (char *)malloc((size_t)cig_len + 1);
    {
        int64_t k = 0;
        for (int64_t i = 0; i < cigar_in_len; i++) {
            char c = cigar_in[i];
            if (c == 'I' || c == 'D') cigar[k++] = c;
            else { cigar[k++] = 'D'; cigar[k++] = 'I'; }
        }
    }
    /* src/aln.pyx:279-311 */
    int32_t *inss = (int32_t *)calloc((size_t)cig_len + 1, sizeof(int32_t));
    int32_t *dels = (int32_t *)calloc((size_t)cig_len + 1, sizeof(int32_t));
    for (int64_t i = 0; i < cig_len; i++) {
        inss[i + 1] = inss[i] + (cigar[i] == 'I');
        dels[i + 1] = dels[i] + (cigar[i] == 'D');
    }
    if (inss[cig_len] != seq_len || dels[cig_len] != ref_len) {
        /* reference: undefined behaviour; we refuse */
        free(cigar); free(inss); free(dels);
        *status = ORC_ERR_BAD_INPUT; return -1;
    }

    /* get_breaks: src/aln.pyx:344-358 with array_size = S+R+1 (:391-392) */
    int64_t array_size = seq_len + ref_len + 1;
    int64_t chunk_m1 = (int64_t)max_b_rows - 1;
    int64_t nbrk = 1 + ((array_size - 1) + chunk_m1 - 1) / chunk_m1; /* 1+ceil() */
    int64_t *breaks = (int64_t *)calloc((size_t)nbrk, sizeof(int64_t));
    for (int64_t i = 0; i < nbrk - 1; i++) {
        breaks[i] = i * chunk_m1;
        if (i > 0 && inss[breaks[i] + 1] == inss[breaks[i]] + 1 &&
            dels[breaks[i]] == dels[breaks[i] - 1] + 1)
            breaks[i] -= 1;
    }
    breaks[nbrk - 1] = array_size - 1;

    const int64_t a_rows = seq_len + 1, a_cols = ref_len + 1; /* :396-397 */
    const int b_cols = 2 * r + 1;                             /* :398 */
    int64_t out_len = 0;
    int32_t *zeros = (int32_t *)calloc((size_t)max_n, sizeof(int32_t));

    for (int64_t brk_idx = 0; brk_idx < nbrk - 1; brk_idx++) { /* :445 */
        const int64_t brk = breaks[brk_idx], next_brk = breaks[brk_idx + 1];
        const int64_t b_rows = next_brk - brk + 1;
        /* :424,:450 -- the reference keeps one (max_b_rows+1)-row buffer and
         * zero-fills all of it per chunk; only the first b_rows rows are ever
         * addressed, so a zeroed buffer of b_rows rows is equivalent. */
        const size_t plane = (size_t)b_rows * (size_t)b_cols * NDIMS;
        float *matrix = (float *)calloc(plane * NTYPS, sizeof(float));
#define M(typ, brow, bcol, dim) \
    matrix[(size_t)(typ) * plane + ((size_t)(brow) * (size_t)b_cols + (size_t)(bcol)) * NDIMS + (dim)]

        /* :453-456 python slices clip at the end */
        const int64_t row0 = inss[brk], col0 = dels[brk];
        const int64_t rowN = inss[next_brk], colN = dels[next_brk];
        int64_t ref_end = colN + 1 > ref_len ? ref_len : colN + 1;
        int64_t seq_end = rowN + 1 > seq_len ? seq_len : rowN + 1;
        const uint8_t *ref = full_ref + col0;
        const uint8_t *seq = full_seq + row0;
        const int64_t rlen = ref_end - col0, slen = seq_end - row0;
        int32_t *np_info = (int32_t *)malloc((size_t)(rlen > 0 ? rlen : 1) * 2 * max_n * sizeof(int32_t));
        int32_t *np_info_seq = (int32_t *)malloc((size_t)(slen > 0 ? slen : 1) * 2 * max_n * sizeof(int32_t));
        npore_oracle_get_np_info(ref, rlen, max_n, max_l, np_info);
        npore_oracle_get_np_info(seq, slen, max_n, max_l, np_info_seq);

        /* coordinate maps src/aln.pyx:317-338 */
#define B2A_ROW(brow, bcol) ((int64_t)inss[(brow)] + r - (bcol))
#define B2A_COL(brow, bcol) ((int64_t)dels[(brow)] - r + (bcol))
#define A2B_COL(arow, acol) ((int64_t)inss[(arow) + (acol)] - (arow) + r)

        /* init pass :465-478 */
        for (int64_t b_row = 0; b_row < b_rows; b_row++)
            for (int b_col = 0; b_col < b_cols; b_col++) {
                int64_t a_row = B2A_ROW(b_row + brk, b_col);
                int64_t a_col = B2A_COL(b_row + brk, b_col);
                if (a_row < row0 || a_col < col0 || a_row > rowN || a_col > colN ||
                    b_col == 0 || b_col == 2 * r)
                    continue;
                M(T_LEN, b_row, b_col, D_VAL) = (float)(ORC_INF * (a_row - row0 + a_col - col0));
                M(T_LEN, b_row, b_col, D_TYP) = T_MAT;
                M(T_LEN, b_row, b_col, D_RUN) = 0;
                M(T_SHR, b_row, b_col, D_VAL) = (float)(ORC_INF * (a_row - row0 + a_col - col0));
                M(T_SHR, b_row, b_col, D_TYP) = T_MAT;
                M(T_SHR, b_row, b_col, D_RUN) = 0;
            }

        /* fill :481-667 */
        for (int64_t b_row = 0; b_row < b_rows; b_row++)
            for (int b_col = 0; b_col < b_cols; b_col++) {
                int64_t a_row = B2A_ROW(b_row + brk, b_col);
                int64_t a_col = B2A_COL(b_row + brk, b_col);
                /* :497-499 (neighbour indices, :487-492, are only formed after
                 * this guard here: the reference forms them first and may index
                 * inss[-1], but never uses the value) */
                if (a_row < row0 || a_col < col0 || a_row > rowN || a_col > colN)
                    continue;
                if (b_col == 0 || b_col == 2 * r) {        /* :502-507 */
                    for (int typ = 0; typ < NTYPS; typ++) {
                        M(typ, b_row, b_col, D_VAL) = (float)(ORC_INF * (b_row + 1));
                        M(typ, b_row, b_col, D_TYP) = T_MAT;
                        M(typ, b_row, b_col, D_RUN) = 0;
                    }
                    continue;
                }
                int64_t b_top_row = -1, b_top_col = -1, b_left_row = -1, b_left_col = -1;
                int64_t b_diag_row = -1, b_diag_col = -1;
                if (a_row > row0) { b_top_row = (a_row - 1 + a_col) - brk; b_top_col = A2B_COL(a_row - 1, a_col); }
                if (a_col > col0) { b_left_row = (a_row + a_col - 1) - brk; b_left_col = A2B_COL(a_row, a_col - 1); }
                if (a_row > row0 && a_col > col0) { b_diag_row = (a_row + a_col - 2) - brk; b_diag_col = A2B_COL(a_row - 1, a_col - 1); }
                const int64_t ref_idx = a_col - col0 - 1;
                const int64_t seq_idx = a_row - row0 - 1;

                /* :510-521 */
                const int32_t *l, *l_idx, *l_seq, *l_idx_seq;
                if (a_col >= a_cols - 1) { l = zeros; l_idx = zeros; }
                else { l = np_info + ((ref_idx + 1) * 2 + 0) * max_n; l_idx = np_info + ((ref_idx + 1) * 2 + 1) * max_n; }
                if (a_row >= a_rows - 1) { l_seq = zeros; l_idx_seq = zeros; }
                else { l_seq = np_info_seq + ((seq_idx + 1) * 2 + 0) * max_n; l_idx_seq = np_info_seq + ((seq_idx + 1) * 2 + 1) * max_n; }

                float val1, val2;
                int run;

                /* INS :525-543 */
                if (a_row == row0) {
                    M(T_INS, b_row, b_col, D_VAL) = (float)(ORC_INF * (a_col - col0 + 1));
                    M(T_INS, b_row, b_col, D_TYP) = T_DEL;
                    M(T_INS, b_row, b_col, D_RUN) = (float)(a_col - col0);
                } else {
                    val1 = M(T_MAT, b_top_row, b_top_col, D_VAL) + indel_start;
                    M(T_INS, b_row, b_col, D_VAL) = val1;
                    M(T_INS, b_row, b_col, D_TYP) = T_INS;
                    M(T_INS, b_row, b_col, D_RUN) = 1;
                    val2 = M(T_INS, b_top_row, b_top_col, D_VAL) + indel_extend;
                    if (val2 < val1) {
                        if (a_row == row0 + 1) run = 1;
                        else run = (int)M(T_INS, b_top_row, b_top_col, D_RUN) + 1;
                        M(T_INS, b_row, b_col, D_VAL) = val2;
                        M(T_INS, b_row, b_col, D_TYP) = T_INS;
                        M(T_INS, b_row, b_col, D_RUN) = (float)run;
                    }
                }

                /* DEL :547-565 */
                if (a_col == col0) {
                    M(T_DEL, b_row, b_col, D_VAL) = (float)(ORC_INF * (a_row - row0 + 1));
                    M(T_DEL, b_row, b_col, D_TYP) = T_INS;
                    M(T_DEL, b_row, b_col, D_RUN) = (float)(a_row - row0);
                } else {
                    val1 = M(T_MAT, b_left_row, b_left_col, D_VAL) + indel_start;
                    M(T_DEL, b_row, b_col, D_VAL) = val1;
                    M(T_DEL, b_row, b_col, D_TYP) = T_DEL;
                    M(T_DEL, b_row, b_col, D_RUN) = 1;
                    val2 = M(T_DEL, b_left_row, b_left_col, D_VAL) + indel_extend;
                    if (val2 < val1) {
                        if (a_col == col0 + 1) run = 1;
                        else run = (int)M(T_DEL, b_left_row, b_left_col, D_RUN) + 1;
                        M(T_DEL, b_row, b_col, D_VAL) = val2;
                        M(T_DEL, b_row, b_col, D_TYP) = T_DEL;
                        M(T_DEL, b_row, b_col, D_RUN) = (float)run;
                    }
                }

                /* MAT :569-592 */
                if (a_row > row0 && a_col > col0) {
                    if (M(T_MAT, b_diag_row, b_diag_col, D_TYP) == T_MAT)
                        run = (int)M(T_MAT, b_diag_row, b_diag_col, D_RUN) + 1;
                    else
                        run = 1;
                    val1 = M(T_MAT, b_diag_row, b_diag_col, D_VAL) +
                           sub_scores[seq[seq_idx] * 5 + ref[ref_idx]];
                    M(T_MAT, b_row, b_col, D_VAL) = val1;
                    M(T_MAT, b_row, b_col, D_TYP) = T_MAT;
                    M(T_MAT, b_row, b_col, D_RUN) = (float)run;
                } else {
                    val1 = M(T_DEL, b_row, b_col, D_VAL) + ORC_INF;
                }
                for (int typ = 1; typ < NTYPS; typ++) {    /* :585-592 */
                    val2 = M(typ, b_row, b_col, D_VAL);
                    if (val2 < val1) {
                        val1 = val2;
                        run = (int)M(typ, b_row, b_col, D_RUN);
                        M(T_MAT, b_row, b_col, D_VAL) = val2;
                        M(T_MAT, b_row, b_col, D_TYP) = (float)typ;
                        M(T_MAT, b_row, b_col, D_RUN) = (float)run;
                    }
                }

                /* LEN :596-633 */
                if (a_row == row0) {
                    M(T_LEN, b_row, b_col, D_VAL) = (float)(ORC_INF * (a_col - col0));
                    M(T_LEN, b_row, b_col, D_TYP) = T_DEL;
                    M(T_LEN, b_row, b_col, D_RUN) = (float)(a_col - col0);
                }
                for (int n = 1; n <= max_n; n++) {
                    int n_idx = n - 1;
                    if (l[n_idx] == 0 || l_seq[n_idx] == 0 || l_idx[n_idx] != 0 ||
                        !orc_match(seq, slen, seq_idx + 1, ref, rlen, ref_idx + 1, n))
                        continue;
                    if (a_row + n <= rowN) {               /* :611, first half */
                        int64_t b_ndown_row = (a_row + n + a_col) - brk;
                        int64_t b_ndown_col = A2B_COL(a_row + n, a_col);
                        if (b_ndown_col > 0) {
                            if (l_idx_seq[n_idx] == 0) {   /* start insertion :613-619 */
                                val1 = M(T_MAT, b_row, b_col, D_VAL) +
                                       orc_np_score(n, l[n_idx], 1, np_scores, np_dim, max_l);
                                if (val1 < M(T_LEN, b_ndown_row, b_ndown_col, D_VAL)) {
                                    M(T_LEN, b_ndown_row, b_ndown_col, D_VAL) = val1;
                                    M(T_LEN, b_ndown_row, b_ndown_col, D_TYP) = T_LEN;
                                    M(T_LEN, b_ndown_row, b_ndown_col, D_RUN) = (float)n;
                                }
                            } else {                       /* continue :621-633 */
                                run = (int)M(T_LEN, b_row, b_col, D_RUN);
                                if (run > 0 && a_row - run >= row0) {
                                    int64_t b_runup_row = (a_row - run + a_col) - brk;
                                    int64_t b_runup_col = A2B_COL(a_row - run, a_col);
                                    if (b_runup_col < 2 * r) {
                                        val1 = M(T_MAT, b_runup_row, b_runup_col, D_VAL) +
                                               orc_np_score(n, l[n_idx], (int)(run / n) + 1, np_scores, np_dim, max_l);
                                        if (val1 < M(T_LEN, b_ndown_row, b_ndown_col, D_VAL)) {
                                            M(T_LEN, b_ndown_row, b_ndown_col, D_VAL) = val1;
                                            M(T_LEN, b_ndown_row, b_ndown_col, D_TYP) = T_LEN;
                                            M(T_LEN, b_ndown_row, b_ndown_col, D_RUN) = (float)(run + n);
                                        }
                                    }
                                }
                            }
                        }
                    }
                }

                /* SHR :637-667 */
                if (a_col == col0) {
                    M(T_SHR, b_row, b_col, D_VAL) = (float)(ORC_INF * (a_row - row0));
                    M(T_SHR, b_row, b_col, D_TYP) = T_INS;
                    M(T_SHR, b_row, b_col, D_RUN) = (float)(a_row - row0);
                }
                for (int n = 1; n <= max_n; n++) {
                    int n_idx = n - 1;
                    if (l[n_idx] == 0) continue;
                    if (a_col + n <= colN) {               /* :647, first half */
                        int64_t b_nright_row = (a_row + a_col + n) - brk;
                        int64_t b_nright_col = A2B_COL(a_row, a_col + n);
                        if (b_nright_col < 2 * r) {
                            if (l_idx[n_idx] == 0) {       /* start deletion :648-654 */
                                val1 = M(T_MAT, b_row, b_col, D_VAL) +
                                       orc_np_score(n, l[n_idx], -1, np_scores, np_dim, max_l);
                                if (val1 < M(T_SHR, b_nright_row, b_nright_col, D_VAL)) {
                                    M(T_SHR, b_nright_row, b_nright_col, D_VAL) = val1;
                                    M(T_SHR, b_nright_row, b_nright_col, D_TYP) = T_SHR;
                                    M(T_SHR, b_nright_row, b_nright_col, D_RUN) = (float)n;
                                }
                            } else {                       /* continue :656-667 */
                                run = (int)M(T_SHR, b_row, b_col, D_RUN);
                                if (run > 0 && a_col - run >= col0) {
                                    int64_t b_runleft_row = (a_row + a_col - run) - brk;
                                    int64_t b_runleft_col = A2B_COL(a_row, a_col - run);
                                    if (b_runleft_col > 0) {
                                        val1 = M(T_MAT, b_runleft_row, b_runleft_col, D_VAL) +
                                               orc_np_score(n, l[n_idx], (int)(-run / n) - 1, np_scores, np_dim, max_l);
                                        if (val1 < M(T_SHR, b_nright_row, b_nright_col, D_VAL)) {
                                            M(T_SHR, b_nright_row, b_nright_col, D_VAL) = val1;
                                            M(T_SHR, b_nright_row, b_nright_col, D_TYP) = T_SHR;
                                            M(T_SHR, b_nright_row, b_nright_col, D_RUN) = (float)(run + n);
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }

        /* traceback :670-742 */
        int64_t a_row = rowN, a_col = colN;
        int64_t aln_cap = (rowN - row0) + (colN - col0) + 8;
        char *aln = (char *)malloc((size_t)aln_cap);
        int64_t aln_len = 0;
        while (a_row > row0 || a_col > col0) {
            if (a_row < 0) { *status |= ORC_ERR_ROW_NEG; break; }   /* :689 */
            if (a_col < 0) { *status |= ORC_ERR_COL_NEG; break; }   /* :699 */
            int64_t b_row = (a_row + a_col) - brk;
            if (b_row < 0 || b_row >= b_rows || a_row < row0 || a_col < col0) {
                *status |= ORC_ERR_OUT_OF_CHUNK; break;
            }
            int64_t b_col = A2B_COL(a_row, a_col);
            if (b_col < 0 || b_col >= b_cols) { *status |= ORC_ERR_OUT_OF_CHUNK; break; }
            int typ = (int)M(T_MAT, b_row, b_col, D_TYP);
            int run = (int)M(T_MAT, b_row, b_col, D_RUN);
            if (run < 1) { *status |= ORC_ERR_RUN_LT1; break; }     /* :708 */
            if (aln_len + run > aln_cap) { *status |= ORC_ERR_OUT_OF_CHUNK; break; }
            if (typ == T_LEN || typ == T_INS) {                     /* :719-722 */
                for (int i = 0; i < run; i++) aln[aln_len++] = 'I';
                a_row -= run;
            } else if (typ == T_SHR || typ == T_DEL) {              /* :723-726 */
                for (int i = 0; i < run; i++) aln[aln_len++] = 'D';
                a_col -= run;
            } else if (typ == T_MAT) {                              /* :727-736 */
                int i = 0, bad = 0;
                while (i < run) {
                    a_row -= 1; a_col -= 1;
                    if (a_row < row0 || a_col < col0) { bad = 1; break; }
                    aln[aln_len++] = (ref[a_col - col0] == seq[a_row - row0]) ? '=' : 'X';
                    i += 1;
                }
                if (bad) { *status |= ORC_ERR_OUT_OF_CHUNK; break; }
            } else { *status |= ORC_ERR_BAD_TYPE; break; }          /* :737-739 */
        }
        /* :742 full_aln += aln[::-1] */
        if (out_len + aln_len > out_cap) {
            *status |= ORC_ERR_OUT_CAP;
            free(aln); free(np_info); free(np_info_seq); free(matrix);
            free(zeros); free(breaks); free(inss); free(dels); free(cigar);
            return -1;
        }
        for (int64_t i = 0; i < aln_len; i++) out[out_len + i] = aln[aln_len - 1 - i];
        out_len += aln_len;

        free(aln); free(np_info); free(np_info_seq); free(matrix);
#undef M
#undef B2A_ROW
#undef B2A_COL
#undef A2B_COL
    }
    free(zeros); free(breaks); free(inss); free(dels); free(cigar);
    return out_len;
}

/* Batch helper for timing/tests: runs reads [0,n) serially.  Offsets are int64
 * prefix arrays of length n+1 into the concatenated byte buffers.  out_off[i]
 * gives where read i's output starts (caller-sized: cap_i = out_off[i+1]-out_off[i]). */
int64_t npore_oracle_align_batch(int64_t n_reads, const uint8_t *refs, const int64_t *ref_off,
                                 const uint8_t *seqs, const int64_t *seq_off,
                                 const char *cigars, const int64_t *cig_off,
                                 const float *sub_scores, const float *np_scores,
                                 int max_n, int max_l, float indel_start, float indel_extend,
                                 int max_b_rows, int r, char *out, const int64_t *out_off,
                                 int64_t *out_len, int32_t *status)
{
    int64_t nbad = 0;
    for (int64_t i = 0; i < n_reads; i++) {
        out_len[i] = npore_oracle_align(refs + ref_off[i], ref_off[i + 1] - ref_off[i],
                                        seqs + seq_off[i], seq_off[i + 1] - seq_off[i],
                                        cigars + cig_off[i], cig_off[i + 1] - cig_off[i],
                                        sub_scores, np_scores, max_n, max_l, indel_start,
                                        indel_extend, max_b_rows, r, out + out_off[i],
                                        out_off[i + 1] - out_off[i], &status[i]);
        if (out_len[i] < 0 || status[i]) nbad++;
    }
    return nbad;
}
