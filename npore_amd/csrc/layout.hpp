// layout.hpp -- data layout shared by host orchestration and the gfx950 kernels.
//
// Work unit: a CHUNK = one independent stretch of <= max_b_rows anti-diagonals
// ("b-rows") of one read (reference src/aln.pyx:344-358, 445-456).  A chunk is a
// rectangle [row0,row0+drows] x [col0,col0+dcols] of the read's alignment
// matrix ("A" coordinates: row = bases of the read consumed, col = bases of the
// reference consumed) swept along anti-diagonals inside a band of half-width r
// around the input alignment path ("B" coordinates, src/aln.pyx:317-338):
//   b_row = a_row + a_col,  b_col = inss[b_row] - a_row + r.
//
// Per-chunk packed annotation words (one entry per local row i / local col j):
//
//   seqw[i]  (i = a_row-row0, 0..drows)     "what a cell in row i needs to know"
//     bits 14-31  the 6 read bases before row i, oldest first: code(seq[i-6+k]) << (14+3k)
//                 (code 7 = before the chunk slice); the cell's own base seq[i-1] is
//                 bits 29-31, so the n most recent bases are simply word >> (32-3n)
//     bits  8-13  bit n-1: read position i-n lies in an n-polymer  (L_seq != 0)
//     bits  0-5   bit n-1: ... and is its first copy               (L_IDX_seq == 0)
//   refw[j].x (j = a_col-col0, 0..dcols)
//     bits 14-31  the 6 reference bases from col j on: code(ref[j+k]) << (14+3k)
//                 (code 6 = past the chunk slice)
//     bits  8-13  bit n-1: reference position j starts an n-polymer (L != 0 && L_IDX == 0)
//     bits  0-2   the cell's own reference base ref[j-1]
//                 (v_alignbit(refw.x, seqw, 25) & 0x3FC is then the byte offset of
//                 sub_scores[seq[i-1]][ref[j-1]] in a [ref][seq][4] LDS table, see SUBT_*)
//   refw[j].y
//     bits  0-5   bit n-1: reference position j-n lies in an n-polymer (L != 0)
//     bits  6-11  bit n-1: ... and is its first copy                  (L_IDX == 0)
//   refw[j].z/.w  the column's two highest-period SHR candidates, pre-decoded (make_shr_desc):
//     bits 2-4   period n (0 = none), so that word & 0x1C = 4n (a ds_bpermute lane address)
//     bit  5     (.z only) the column has a second candidate (.w != 0)
//     bit  6     L >= 32: the score row is not in the LDS table
//     bit  7     .w: the column has more than two candidate periods (rare; the rest is decoded from
//                .y and the L window); .z: bit 6 of either word or bit 7 of .w is set ("rare column")
//     bits 8-14  L of reference position j-n for that period
//     bits 15-30 byte address, inside the LDS score table, of the entry for "call length L-1":
//                (((n-1)*32 + min(L, max_l-1))*33 + 32 - L)*4; a deletion of q more copies reads 4q bytes higher
//     bit  31    "first copy": start a deletion rather than continue one
//   refl[j]  8 bytes: byte n-1 = L of reference position j for period n (0..max_l)
//
// n-polymer annotation follows get_np_info (reference src/aln.pyx:179-251) on the
// chunk's own slices (src/aln.pyx:453-456); flags are stored from the point of
// view of the cell that RECEIVES a lengthen/shorten move (see kernels.hpp).
//
// Traceback word per cell (the only per-cell HBM traffic): run | typ << 29 (tb_word), the
// MAT.TYP / MAT.RUN pair the reference's traceback reads (src/aln.pyx:684-685).  The type sits in the TOP bits so
// that a run register can carry its type tag through the recurrence -- run + 1 of a tagged run is the tagged run + 1,
// the tag falls out of a 16-bit shift -- and the fill kernel never has to assemble the word.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NPORE_HD __host__ __device__ __forceinline__
#else
#define NPORE_HD inline
#endif

namespace npore {

enum : int { T_MAT = 0, T_INS = 1, T_LEN = 2, T_DEL = 3, T_SHR = 4 };

constexpr int TB_TYP_SHIFT = 29;
constexpr uint32_t TB_RUN_MASK = (1u << TB_TYP_SHIFT) - 1u;
NPORE_HD uint32_t tb_word(int typ, uint32_t run) { return ((uint32_t)typ << TB_TYP_SHIFT) | run; }
NPORE_HD int tb_typ(uint32_t w) { return (int)(w >> TB_TYP_SHIFT); }
NPORE_HD int tb_run(uint32_t w) { return (int)(w & TB_RUN_MASK); }

constexpr int MER_SHIFT = 14, FLAG_SHIFT = 8;  // positions of the 18-bit base field and of the 6 "in an n-polymer" flags
constexpr uint32_t SEQW_SENTINEL = 0x3FFFFu << MER_SHIFT;  // six code-7 bases, no flags
constexpr uint32_t REFW_SENTINEL = 0x36DB6u << MER_SHIFT;  // six code-6 bases, no flags, own base 0
// substitution-score table as the kernel indexes it: entry (ref << 5 | seq << 2 | g), g = 2 don't-care bits
constexpr int SUBT_ENTRIES = 256;
constexpr int MAX_PERIOD = 6;                  // kernels are specialised for max_n <= 6
constexpr float INF_F = 100.0f;                // reference src/aln.pyx:428
constexpr int HIST_PAD = 6;                    // never-written history records either side of a row (cell.hpp)

constexpr uint32_t DSC_N4 = 0x1Cu, DSC_BIGL = 1u << 6, DSC_MORE = 1u << 7, DSC_START = 1u << 31;
// summary bits, only in .z (the first candidate): the column has a second candidate / something in the
// column needs the generic path (L >= NP_LT in either candidate, or more than two candidates)
constexpr uint32_t DSC_HAS2 = 1u << 5, DSC_RARE = 1u << 7;
// LDS score table: [MAX_PERIOD][NP_LT][NP_CT] floats; a row holds np_scores[n][L][call] with the call length
// DEcreasing -- entry NP_LT - 1 - call for call = NP_LT-1 ... 0 -- followed by one guard entry (NP_C0) holding INF_F
// for call < 0 (np_score's "call < 0 -> 100"), which a candidate reaches by clamping "copies deleted so far" at the
// row's own L (byte 1 of its descriptor): deleting q more copies is q entries up from the descriptor's address
// (one shift-add).  25 KB; the odd row length also spreads the rows over the LDS banks.
constexpr int NP_LT = 32, NP_CT = 33, NP_C0 = 1;
// max_l: np_score clamps the table ROW to max_l - 1 (src/aln.pyx:257-274 as called); L itself is capped at max_l, so
// the call length L - 1 - q never needs the clamp
// A period n > max_l never scores: np_score returns its constant 100 (`n > max_n` with max_l passed as max_n,
// src/aln.pyx:265) whatever L and the copies deleted are.  Such a candidate keeps its period and start flag but
// carries L = 0 (the generic path's "invalid" length) and the address of a guard entry, which holds that constant and
// which the clamp at L = 0 never leaves.  (Only contexts with max_l < max_n <= 6 have such periods.)
NPORE_HD uint32_t make_shr_desc(int n, bool start, uint32_t L, int max_l)
{
    if (n > max_l)
        return ((uint32_t)n << 2) | ((((uint32_t)(n - 1) * NP_LT) * NP_CT + (uint32_t)NP_LT) * 4u << 15) | (start ? DSC_START : 0u);
    // (L <= max_l, so the clamp only bites at L == max_l: one row up.  Written as a correction of the L-only index:
    // selecting the row first made the annotate kernel 50 % slower.)
    const uint32_t over = (L >= (uint32_t)max_l) ? (uint32_t)NP_CT : 0u;
    const uint32_t addr = L < (uint32_t)NP_LT ? ((((uint32_t)(n - 1) * NP_LT + L) * NP_CT + (uint32_t)NP_LT - L) - over) * 4u : 0u;
    return ((uint32_t)n << 2) | (L >= (uint32_t)NP_LT ? DSC_BIGL : 0u) | (L << 8) | (addr << 15) | (start ? DSC_START : 0u);
}

struct alignas(16) ChunkDesc {
    int32_t read_id;
    int32_t brk;      // first global b-row of the chunk
    int32_t nrows;    // b-rows in the chunk (next_brk - brk + 1)
    int32_t row0;     // inss[brk]
    int32_t col0;     // dels[brk]
    int32_t drows;    // inss[next_brk] - row0
    int32_t dcols;    // dels[next_brk] - col0
    int32_t out_cap;  // drows + dcols: upper bound on ops emitted
    int64_t steps_off;  // index of this READ's step 0 in the steps byte array
    int64_t inss_off;   // index of this READ's inss[0] in the inss int32 array
    int64_t seqw_off;   // index of seqw[0] of this chunk
    int64_t refw_off;   // index of refw[0] / refl[0] of this chunk
    int64_t tb_off;     // index (in uint32) of this chunk's traceback block
    int64_t out_off;    // byte offset of this chunk's output slot
    int64_t seq_off;    // byte offset of the READ's first base in the seqs buffer (for '='/'X')
    int64_t ref_off;    // byte offset of the READ's first reference base in the refs buffer
    int32_t plain_lo;   // anti-diagonals [plain_lo, plain_hi) of the chunk are "plain" (cell.hpp,
    int32_t plain_hi;   // step_is_plain): both conditions are monotone along the input path
    int32_t pad_[2];
};

// traceback row stride (uint32 words) for band half-width r: 2r+1 rounded up to 4
static inline int tb_stride(int r) { return ((2 * r + 1) + 3) & ~3; }

}  // namespace npore
