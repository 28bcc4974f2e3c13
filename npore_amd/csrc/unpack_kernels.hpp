// unpack_kernels.hpp -- the inputs of align() straight from BAM records, on the device.
//
// realign_read (reference src/bam.pyx:51-61) hands align() three things per read: the reference bases under the
// alignment (`get_reference_sequence`, :45 -- here: a slice of the FASTA contig), the query bases without the soft
// clips (:42) and the expanded CIGAR without S and H (:59), bases as codes 'NACGT-' -> 0..5 (src/cig.pyx:212-229).
// The host twin is pack_records (npore_api.cpp: base_codes / nibble_codes of hostio.hpp + the op loop); the file
// pipeline used to run it on the host for every batch -- 18 us per 10 kb read and core, 40 KB per read across PCIe.
// Here the pipeline uploads the HEAD of every record as it lies in the BAM stream (fixed fields, name, CIGAR words,
// 4-bit bases: ~11 KB per 10 kb read; qualities and tags stay on the host) and the FASTA once per run; one workgroup per
// read writes the three arrays where the host-buffer path would have uploaded them.  The sizes (ref / read / CIGAR
// length per read) are still summed on the host: the planning of groups and buffers needs them before anything runs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace npore {

struct CtgEntry {
    const char *bases;     // upper-cased contig on the device; nullptr: the BAM reference has no contig in the FASTA
    int64_t len;
};

struct UnpackParams {
    const uint8_t *raw;          // record heads one after the other (block_size word first, as in the BAM stream)
    const int64_t *raw_off;      // [n + 1]
    const CtgEntry *ctg;         // by BAM reference id
    int n_ctg;
    uint8_t *refs; const int64_t *ref_off;     // outputs and their offsets ([n + 1], from 0)
    uint8_t *seqs; const int64_t *seq_off;
    char *cigs; const int64_t *cig_off;
    int64_t n_reads;
};

__device__ __forceinline__ uint32_t ld16(const uint8_t *q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8); }
__device__ __forceinline__ uint32_t ld32(const uint8_t *q) { return ld16(q) | (ld16(q + 2) << 16); }

// one workgroup of 256 per read
__global__ __launch_bounds__(256) void unpack_records_kernel(UnpackParams p)
{
    const int64_t k = blockIdx.x;
    if (k >= p.n_reads) return;
    const int t = threadIdx.x;
    const uint8_t *f = p.raw + p.raw_off[k] + 4;                 // the fixed fields (hostio.hpp RecView)
    const int32_t rid = (int32_t)ld32(f), pos = (int32_t)ld32(f + 4);
    const int l_rn = f[8], nc = (int)ld16(f + 12);
    const uint8_t *cg = f + 32 + l_rn, *sq = cg + 4 * (size_t)nc;

    // ---- reference bases: contig[pos, pos + rl), zeros (N) where the slice leaves the contig
    {
        const int64_t rl = p.ref_off[k + 1] - p.ref_off[k];
        uint8_t *ro = p.refs + p.ref_off[k];
        const CtgEntry c = (rid >= 0 && rid < p.n_ctg) ? p.ctg[rid] : CtgEntry{nullptr, 0};
        for (int64_t q = t; q < rl; q += 256) {
            const int64_t g = (int64_t)pos + q;
            uint8_t code = 0;
            if (c.bases && g >= 0 && g < c.len) {
                const char ch = c.bases[g];
                code = (uint8_t)((ch == 'A' ? 1 : 0) + (ch == 'C' ? 2 : 0) + (ch == 'G' ? 3 : 0) + (ch == 'T' ? 4 : 0) + (ch == '-' ? 5 : 0));
            }
            ro[q] = code;
        }
    }
    // ---- soft clips (hostio.hpp rec_clips): a leading / trailing S, possibly behind / in front of an H
    auto op_of = [&](int c) { return ld32(cg + 4 * (size_t)c); };
    int64_t lead = 0;
    if (nc >= 1 && (op_of(0) & 15u) == 4) lead = op_of(0) >> 4;
    if (nc > 1 && (op_of(0) & 15u) == 5 && (op_of(1) & 15u) == 4) lead = op_of(1) >> 4;
    // ---- query bases: "=ACMGRSVTWYHKDBN"[nibble] -> 'NACGT' codes: A = 1, C = 2, G = 4, T = 8
    {
        const int64_t sl = p.seq_off[k + 1] - p.seq_off[k];
        uint8_t *so = p.seqs + p.seq_off[k];
        for (int64_t q = t; q < sl; q += 256) {
            const int64_t idx = lead + q;
            const uint32_t b = sq[idx >> 1], nib = (idx & 1) ? (b & 15u) : (b >> 4);
            so[q] = (uint8_t)((nib == 1 ? 1 : 0) + (nib == 2 ? 2 : 0) + (nib == 4 ? 3 : 0) + (nib == 8 ? 4 : 0));
        }
    }
    // ---- expanded CIGAR without S and H: every thread takes a contiguous share of the operations; their places
    // follow from the sums of the shares before it
    {
        __shared__ int64_t s_sum[257];
        char *co = p.cigs + p.cig_off[k];
        const int seg = (nc + 255) / 256;
        const int c0 = min(nc, t * seg), c1 = min(nc, c0 + seg);
        int64_t mine = 0;
        for (int c = c0; c < c1; c++) {
            const uint32_t w = op_of(c), op = w & 15u;
            if (op != 4 && op != 5) mine += w >> 4;
        }
        s_sum[t + 1] = mine;
        __syncthreads();
        if (t == 0) {
            s_sum[0] = 0;
            for (int q = 1; q <= 256; q++) s_sum[q] += s_sum[q - 1];
        }
        __syncthreads();
        int64_t at = s_sum[t];
        for (int c = c0; c < c1; c++) {
            const uint32_t w = op_of(c), op = w & 15u, len = w >> 4;
            if (op == 4 || op == 5) continue;
            const char ch = op < 10 ? "MIDNSHP=XB"[op] : '?';
            for (uint32_t q = 0; q < len; q++) co[at + q] = ch;
            at += len;
        }
    }
}

// ---------------------------------------------------------------------------
// The file pipeline's results across PCIe: standardize_kernel leaves every read's CIGAR text in a slot of the worst-case
// size (2 bytes per op + 16: ~40 KB for a 10 kb read, a few KB of it used), and rounds 4 - 5a copied the whole slot range to
// the host -- 160 MB per batch of 4 000 reads for ~12 MB of text, into page-locked memory whose allocation (six batch
// slots) was most of a file run's first half second.  One wavefront per read moves the text to the front of a compact
// buffer (an atomic cursor, 16-byte granules; the order is whatever the waves make it) and notes where: the host then
// copies the used front only.
struct CompactParams {
    const uint8_t *out;            // the slots
    const int64_t *out_off;        // [reads of the batch + 1]
    const int64_t *out_len;        // bytes of text per read (<= 0: none)
    int64_t read_base, n_reads;    // this group's reads within the batch
    uint8_t *ctext;                // the batch's compact buffer
    unsigned long long *cursor;    // bytes of it in use
    int64_t *coff;                 // [n_reads] of this group: where the read's text begins (-1: no room; the host sizes the buffer for all slots + a granule per read)
    int64_t cap;
};

__global__ __launch_bounds__(64) void compact_texts_kernel(CompactParams p)
{
    const int64_t rd = blockIdx.x;
    if (rd >= p.n_reads) return;
    const int lane = threadIdx.x;
    const int64_t grd = p.read_base + rd, len = p.out_len[grd];
    if (len <= 0) {
        if (lane == 0) p.coff[rd] = 0;
        return;
    }
    unsigned long long pos = 0;
    if (lane == 0) pos = atomicAdd(p.cursor, ((unsigned long long)len + 15ull) & ~15ull);
    pos = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pos >> 32)) << 32) |
          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pos);
    if ((int64_t)pos + len > p.cap) {
        if (lane == 0) p.coff[rd] = -1;
        return;
    }
    const uint8_t *src = p.out + p.out_off[grd];
    uint8_t *dst = p.ctext + pos;
    for (int64_t k = lane; k < len; k += 64) dst[k] = src[k];
    if (lane == 0) p.coff[rd] = (int64_t)pos;
}

}  // namespace npore
