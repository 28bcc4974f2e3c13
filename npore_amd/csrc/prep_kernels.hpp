// prep_kernels.hpp -- device-side preparation of a batch, so that the whole path
// runs from the raw (ref, read, CIGAR) bytes resident in HBM without a host
// round trip:
//   tile_scan       tiles (CIGAR_TILE ops) of every read: a read of any length is
//                   scanned by as many waves as it has tiles
//   cigar_tile      per tile: op counts, input validity
//   cigar_scan      per read: converted path length, validity, number of chunks
//                   (reference src/aln.pyx:386, 391-392, 344-345); path offset of every tile
//   read_scan       exclusive scans over reads (path offsets, first chunk)
//   expand_path     per tile: step bytes and insertion prefix counts
//                   (src/aln.pyx:279-292; dels[b] = b - inss[b])
//   make_chunks     per chunk: break points incl. the "don't split DI" shift
//                   (src/aln.pyx:349-357), rectangle, sizes, size histogram
//   chunk_scan      offsets of every chunk's arrays; schedule positions
//   sched_scatter   chunk order for the fill kernel, largest first
//   (annot_wave.hpp: annotate_wave_kernel -- per chunk and sequence the n-polymer annotation, get_np_info
//                   src/aln.pyx:179-251 on the chunk slices of src/aln.pyx:453-456, and the word packing of layout.hpp)
// All launches are sized from host-known upper bounds and read the actual counts
// from device memory, so no host synchronisation is needed in between.
#pragma once
#include <hip/hip_runtime.h>

#include "cell.hpp"      // (xp:: switches of measurement builds)
#include "layout.hpp"

namespace npore {

struct PrepParams {
    int64_t n_reads;
    const uint8_t *refs;
    const int64_t *ref_off;
    const uint8_t *seqs;
    const int64_t *seq_off;
    const char *cigs;
    const int64_t *cig_off;
    int max_b_rows, r, tbstride, max_n, max_l;
    int max_chunks;            // capacity of the chunk arrays
    // per read
    int32_t *rd_nsteps;        // [n]
    int32_t *rd_nchunks;       // [n]
    int32_t *rd_status;        // [n]  NPORE_ST_BAD_INPUT or 0
    int64_t *rd_steps_off;     // [n+1] exclusive scan of nsteps
    int32_t *rd_chunk_first;   // [n+1]
    int32_t *rd_tile_first;    // [n+1] exclusive scan of the reads' tile counts
    // per CIGAR tile
    int4 *tile_cnt;            // ops of the tile: x = X/=/M, y = I, z = D, w = invalid ops / bases
    int2 *tile_base;           // steps / 'I' steps of the read before the tile
    // path
    uint8_t *steps;            // [sum nsteps + pad]
    int32_t *inss;             // [sum (nsteps+1)]  read k starts at rd_steps_off[k] + k
    // chunks
    ChunkDesc *descs;
    int32_t *sched;
    int32_t *hist;             // [max_b_rows + 2] counting sort by rows, then running positions
    int32_t *counters;         // [0] number of chunks, [1] overflow flag, [2] fill kernel's queue head
    // annotation
    uint32_t *seqw;
    uint4 *refw;
    uint2 *refl;               // per reference position: bytes 0-5 L for n=1..6, byte 6 L_IDX==0 mask
};

// ---------------------------------------------------------------------------
// exclusive scan of one value per thread over the workgroup (64 ... 1024 threads; wave scans by shuffles, the wave
// sums through LDS); *total = sum over the workgroup.  s_wave: 16 entries of shared memory, reusable afterwards.
// (The single-workgroup scan kernels split their items over blockDim.x threads: 1 024 on a free GPU, 256 when the
// launch has to find room beside a fill kernel -- one wave per SIMD fits next to the fill's four.)
template <class T>
__device__ __forceinline__ T block_scan_1024(T v, T *s_wave, T *total)
{
    const int nwv = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    T before = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) { const T w = q < nwv ? s_wave[q] : (T)0; before += q < wave ? w : (T)0; tot += w; }
    __syncthreads();
    *total = tot;
    return before + inc - v;
}

constexpr int CIGAR_TILE = 2048;   // ops per tile (a 10 kb read is a handful of tiles, a chromosome 10^5)

__device__ __forceinline__ int64_t cigar_tiles(int64_t clen) { return clen <= CIGAR_TILE ? 1 : (clen + CIGAR_TILE - 1) / CIGAR_TILE; }

// single workgroup: exclusive scan of the reads' tile counts
__global__ __launch_bounds__(1024) void tile_scan_kernel(PrepParams p)
{
    __shared__ int64_t s_w[16];
    const int t = threadIdx.x;
    const int64_t per = (p.n_reads + blockDim.x - 1) / blockDim.x;
    const int64_t a = (int64_t)t * per < p.n_reads ? (int64_t)t * per : p.n_reads, b = (a + per < p.n_reads) ? a + per : p.n_reads;
    int64_t l = 0, total;
    for (int64_t k = a; k < b; k++) l += cigar_tiles(p.cig_off[k + 1] - p.cig_off[k]);
    int64_t acc = block_scan_1024(l, s_w, &total);
    if (t == 0) p.rd_tile_first[p.n_reads] = (int32_t)total;
    for (int64_t k = a; k < b; k++) { p.rd_tile_first[k] = (int32_t)acc; acc += cigar_tiles(p.cig_off[k + 1] - p.cig_off[k]); }
}

// read owning tile t: last rd with rd_tile_first[rd] <= t  (-1: no such tile)
__device__ __forceinline__ int64_t tile_owner(const PrepParams &p, int64_t t)
{
    if (t >= p.rd_tile_first[p.n_reads]) return -1;
    int64_t lo = 0, hi = p.n_reads;   // invariant: first[lo] <= t < first[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (p.rd_tile_first[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// wave per tile: op counts; the tile also checks its share of the read's and the reference's bases
__global__ __launch_bounds__(256) void cigar_tile_kernel(PrepParams p)
{
    const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t rd = tile_owner(p, tile);
    if (rd < 0) return;
    const int64_t j = tile - p.rd_tile_first[rd], nt = p.rd_tile_first[rd + 1] - p.rd_tile_first[rd];
    const char *cig = p.cigs + p.cig_off[rd];
    const int64_t clen = p.cig_off[rd + 1] - p.cig_off[rd];
    const int64_t S = p.seq_off[rd + 1] - p.seq_off[rd], R = p.ref_off[rd + 1] - p.ref_off[rd];
    const int64_t k0 = j * CIGAR_TILE, k1 = (k0 + CIGAR_TILE < clen) ? k0 + CIGAR_TILE : clen;
    int nM = 0, nI = 0, nD = 0, nBad = 0;
    for (int64_t k = k0 + lane; k - lane < k1; k += 64) {
        const bool v = k < k1;
        const char c = v ? cig[k] : 'I';
        const bool m = v && (c == 'X' || c == '=' || c == 'M');
        const bool i = v && c == 'I';
        const bool d = v && c == 'D';
        nM += __popcll(__builtin_amdgcn_ballot_w64(m));
        nI += __popcll(__builtin_amdgcn_ballot_w64(i));
        nD += __popcll(__builtin_amdgcn_ballot_w64(d));
        nBad += __popcll(__builtin_amdgcn_ballot_w64(v && !m && !i && !d));
    }
    const uint8_t *sq = p.seqs + p.seq_off[rd], *rf = p.refs + p.ref_off[rd];
    const int64_t s0 = S * j / nt, s1 = S * (j + 1) / nt, r0 = R * j / nt, r1 = R * (j + 1) / nt;
    for (int64_t k = s0 + lane; k - lane < s1; k += 64) nBad += __popcll(__builtin_amdgcn_ballot_w64(k < s1 && sq[k] > 4));
    for (int64_t k = r0 + lane; k - lane < r1; k += 64) nBad += __popcll(__builtin_amdgcn_ballot_w64(k < r1 && rf[k] > 4));
    if (lane == 0) p.tile_cnt[tile] = make_int4(nM, nI, nD, nBad);
}

// wave per read: totals over its tiles and every tile's position in the read's path
__global__ __launch_bounds__(256) void cigar_scan_kernel(PrepParams p)
{
    const int64_t rd = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (rd >= p.n_reads) return;
    const int64_t S = p.seq_off[rd + 1] - p.seq_off[rd], R = p.ref_off[rd + 1] - p.ref_off[rd];
    const int t0 = p.rd_tile_first[rd], t1 = p.rd_tile_first[rd + 1];
    int64_t nM = 0, nI = 0, nD = 0, nBad = 0;
    for (int tb = t0; tb < t1; tb += 64) {
        const int t = tb + lane;
        const int4 c = (t < t1) ? p.tile_cnt[t] : make_int4(0, 0, 0, 0);
        // inclusive scans over the lanes of (steps, 'I' steps) = (2M + I + D, M + I): both < 2 * CIGAR_TILE * 64
        int st = 2 * c.x + c.y + c.z, is = c.x + c.y, m = c.x, i = c.y, d = c.z, bad = c.w;
        const int own_st = st, own_is = is;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int a_st = __shfl_up(st, o), a_is = __shfl_up(is, o);
            if (lane >= o) { st += a_st; is += a_is; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m += __shfl_xor(m, o); i += __shfl_xor(i, o); d += __shfl_xor(d, o); bad += __shfl_xor(bad, o); }
        const int64_t before = 2 * nM + nI + nD + (st - own_st), beforeI = nM + nI + (is - own_is);
        if (t < t1) p.tile_base[t] = make_int2((int)(before < (1ll << 30) ? before : 0), (int)(beforeI < (1ll << 30) ? beforeI : 0));
        nM += m; nI += i; nD += d; nBad += bad;
    }
    const int64_t nsteps = 2 * nM + nI + nD;
    const bool ok = nBad == 0 && nM + nI == S && nM + nD == R && nsteps < (1ll << 30);
    if (lane == 0) {
        const int64_t cm1 = (int64_t)p.max_b_rows - 1;
        p.rd_nsteps[rd] = ok ? (int32_t)nsteps : 0;
        p.rd_nchunks[rd] = (ok && nsteps > 0) ? (int32_t)((nsteps + cm1 - 1) / cm1) : 0;   // src/aln.pyx:345
        p.rd_status[rd] = ok ? 0 : 32 /* NPORE_ST_BAD_INPUT */;
    }
}

// single workgroup: exclusive scans over reads
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(32))) void read_scan_kernel(PrepParams p)
{
    __shared__ int64_t s_w[16];
    const int t = threadIdx.x;
    const int64_t per = (p.n_reads + blockDim.x - 1) / blockDim.x;
    const int64_t a = (int64_t)t * per < p.n_reads ? (int64_t)t * per : p.n_reads, b = (a + per < p.n_reads) ? a + per : p.n_reads;
    int64_t ls = 0, lc = 0, tot_s, tot_c;
    for (int64_t k = a; k < b; k++) { ls += p.rd_nsteps[k]; lc += p.rd_nchunks[k]; }
    int64_t as = block_scan_1024(ls, s_w, &tot_s), ac = block_scan_1024(lc, s_w, &tot_c);
    if (t == 0) {
        p.rd_steps_off[p.n_reads] = tot_s;
        p.rd_chunk_first[p.n_reads] = (int32_t)(tot_c > p.max_chunks ? p.max_chunks : tot_c);
        p.counters[0] = (int32_t)(tot_c > p.max_chunks ? p.max_chunks : tot_c);
        p.counters[1] = tot_c > p.max_chunks;
        p.counters[2] = 0;      // head of the fill kernel's chunk queue
    }
    for (int64_t k = a; k < b; k++) {
        p.rd_steps_off[k] = as;
        p.rd_chunk_first[k] = (int32_t)ac;
        as += p.rd_nsteps[k];
        ac += p.rd_nchunks[k];
    }
}

// wave per tile
__global__ __launch_bounds__(256) void expand_path_kernel(PrepParams p)
{
    const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t rd = tile_owner(p, tile);
    if (rd < 0 || p.rd_status[rd] != 0) return;
    const int64_t j = tile - p.rd_tile_first[rd];
    const char *cig = p.cigs + p.cig_off[rd];
    const int64_t clen = p.cig_off[rd + 1] - p.cig_off[rd];
    const int64_t k_beg = j * CIGAR_TILE, k_end = (k_beg + CIGAR_TILE < clen) ? k_beg + CIGAR_TILE : clen;
    uint8_t *steps = p.steps + p.rd_steps_off[rd];
    int32_t *inss = p.inss + p.rd_steps_off[rd] + rd;
    if (j == 0 && lane == 0) inss[0] = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int2 tbase = p.tile_base[tile];
    int64_t base = tbase.x;      // steps emitted so far
    int32_t baseI = tbase.y;     // 'I' steps emitted so far
    for (int64_t k0 = k_beg; k0 < k_end; k0 += 64) {
        const int64_t k = k0 + lane;
        const bool v = k < k_end;
        const char c = v ? cig[k] : 'D';
        const bool m = v && (c == 'X' || c == '=' || c == 'M');
        const bool i = v && c == 'I';
        const unsigned long long mv = __builtin_amdgcn_ballot_w64(v), mm = __builtin_amdgcn_ballot_w64(m),
                                 mi = __builtin_amdgcn_ballot_w64(i);
        const int64_t off = base + __popcll(mv & lt) + __popcll(mm & lt);
        const int32_t pI = baseI + __popcll((mi | mm) & lt);
        if (v) {
            if (m) {            // X,=,M -> "DI"
                steps[off] = 0; inss[off + 1] = pI;
                steps[off + 1] = 1; inss[off + 2] = pI + 1;
            } else if (i) {
                steps[off] = 1; inss[off + 1] = pI + 1;
            } else {
                steps[off] = 0; inss[off + 1] = pI;
            }
        }
        base += __popcll(mv) + __popcll(mm);
        baseI += __popcll(mi | mm);
    }
}

__global__ __launch_bounds__(256) void make_chunks_kernel(PrepParams p)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.counters[0]) return;
    // read owning chunk k: last rd with rd_chunk_first[rd] <= k
    int64_t lo = 0, hi = p.n_reads;   // invariant: first[lo] <= k < first[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (p.rd_chunk_first[mid] <= k) lo = mid; else hi = mid;
    }
    const int64_t rd = lo;
    const int ci = k - p.rd_chunk_first[rd];
    const int nch = p.rd_nchunks[rd];
    const int32_t nsteps = p.rd_nsteps[rd];
    const uint8_t *steps = p.steps + p.rd_steps_off[rd];
    const int32_t *inss = p.inss + p.rd_steps_off[rd] + rd;
    const int cm1 = p.max_b_rows - 1;
    auto brk_at = [&](int i) -> int {
        if (i >= nch) return nsteps;                                   // src/aln.pyx:357
        int bk = i * cm1;
        if (i > 0 && steps[bk] == 1 && steps[bk - 1] == 0) bk -= 1;     // src/aln.pyx:353-355
        return bk;
    };
    const int brk = brk_at(ci), nxt = brk_at(ci + 1);
    ChunkDesc d;
    d.read_id = (int32_t)rd;
    d.brk = brk;
    d.nrows = nxt - brk + 1;
    d.row0 = inss[brk];
    d.col0 = brk - inss[brk];
    d.drows = inss[nxt] - d.row0;
    d.dcols = (nxt - inss[nxt]) - d.col0;
    d.out_cap = d.drows + d.dcols;
    d.steps_off = p.rd_steps_off[rd];
    d.inss_off = p.rd_steps_off[rd] + rd;
    d.seqw_off = d.drows + 1;                       // sizes for now; chunk_scan turns them into offsets
    d.refw_off = d.dcols + 1;
    d.tb_off = (int64_t)d.nrows * p.tbstride;
    d.out_off = d.out_cap;
    d.seq_off = p.seq_off[rd];
    d.ref_off = p.ref_off[rd];
    // plain range (cell.hpp step_is_plain): ins_l, del_l are non-decreasing in the local row b, so
    //   ins_l - r >= 6 && del_l - r >= 6        holds from some row on,
    //   ins_l + r <= drows && del_l + r <= dcols holds up to some row.
    {
        auto first_ge = [&](bool use_del, int target) {   // smallest b in [0,nrows] with value(b) >= target
            int lo2 = 0, hi2 = d.nrows;
            while (lo2 < hi2) {
                const int mid = (lo2 + hi2) >> 1;
                const int ins_l = inss[brk + mid] - d.row0;
                const int v = use_del ? mid - ins_l : ins_l;
                if (v >= target) hi2 = mid; else lo2 = mid + 1;
            }
            return lo2;
        };
        const int a1 = first_ge(false, p.r + 6), a2 = first_ge(true, p.r + 6);
        const int b1 = first_ge(false, d.drows - p.r + 1), b2 = first_ge(true, d.dcols - p.r + 1);   // first row violating
        d.plain_lo = a1 > a2 ? a1 : a2;
        d.plain_hi = b1 < b2 ? b1 : b2;
        d.pad_[0] = d.pad_[1] = 0;
    }
    p.descs[k] = d;
    int key = p.max_b_rows + 1 - d.nrows;           // larger chunks first
    key = key < 0 ? 0 : key;
    atomicAdd(&p.hist[key], 1);
}

// single workgroup: size -> offset for the four per-chunk arrays; histogram -> start positions
__global__ __launch_bounds__(1024) void chunk_scan_kernel(PrepParams p)
{
    __shared__ int64_t s_w[16];
    const int t = threadIdx.x;
    const int n = p.counters[0];
    const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x;
    const int a = t * per < n ? t * per : n, b = (a + per < n) ? a + per : n;
    int64_t l[4] = {0, 0, 0, 0};
    for (int k = a; k < b; k++) {
        const ChunkDesc &d = p.descs[k];
        l[0] += d.seqw_off; l[1] += d.refw_off; l[2] += d.tb_off; l[3] += d.out_off;
    }
    int64_t acc[4], tot;
    for (int q = 0; q < 4; q++) acc[q] = block_scan_1024(l[q], s_w, &tot);
    for (int k = a; k < b; k++) {
        ChunkDesc &d = p.descs[k];
        const int64_t v0 = d.seqw_off, v1 = d.refw_off, v2 = d.tb_off, v3 = d.out_off;
        d.seqw_off = acc[0]; d.refw_off = acc[1]; d.tb_off = acc[2]; d.out_off = acc[3];
        acc[0] += v0; acc[1] += v1; acc[2] += v2; acc[3] += v3;
    }
    // histogram (max_b_rows + 2 bins) -> exclusive start positions
    __shared__ int32_t hs_w[16];
    const int nb = p.max_b_rows + 2;
    const int hper = (nb + (int)blockDim.x - 1) / (int)blockDim.x;
    const int ha = t * hper < nb ? t * hper : nb, hb = (ha + hper < nb) ? ha + hper : nb;
    int32_t hl = 0, htot;
    for (int k = ha; k < hb; k++) hl += p.hist[k];
    int32_t hacc = block_scan_1024(hl, hs_w, &htot);
    for (int k = ha; k < hb; k++) { const int32_t v = p.hist[k]; p.hist[k] = hacc; hacc += v; }
}

__global__ __launch_bounds__(256) void sched_scatter_kernel(PrepParams p)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.counters[0]) return;
    int key = p.max_b_rows + 1 - p.descs[k].nrows;
    key = key < 0 ? 0 : key;
    p.sched[atomicAdd(&p.hist[key], 1)] = k;
}

// ---------------------------------------------------------------------------
// get_np_regions (reference src/bed.py:56-76) for a batch of independent slices of a genome: the n-polymer
// starts (L != 0 and L_IDX == 0) of every slice and period, in position order.
struct RegionParams {
    const uint8_t *seqs;       // base codes, slices back to back
    const int64_t *seq_off;    // [n_slices + 1]
    int n_slices, max_n, max_l;
    const uint8_t *planes;     // max_n bytes per base: slice k's planes at planes + max_n * seq_off[k], stride = its length (L | start << 7)
    int64_t *counts;           // [max_n][n_slices] counts, then (region_scan) exclusive offsets; total at the end
    int32_t *out_pos;          // position within the slice
    int32_t *out_reps;         // repeat count L (capped at max_l)
};

// workgroup per slice: the number of starts per period, from the planes np_info_wave_kernel (annot_wave.hpp) has written
__global__ __launch_bounds__(1024) void region_count_kernel(RegionParams p)
{
    const int k = blockIdx.x;
    const int64_t off = p.seq_off[k];
    const int len = (int)(p.seq_off[k + 1] - off);
    const uint8_t *planes = p.planes + (size_t)p.max_n * off;
    __shared__ int cnt[MAX_PERIOD];
    if (threadIdx.x < MAX_PERIOD) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int n = 0; n < p.max_n; n++) {
        int c = 0;
        for (int pos = threadIdx.x; pos < len; pos += blockDim.x) c += planes[(size_t)n * len + pos] >> 7;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&cnt[n], c);
    }
    __syncthreads();
    if (threadIdx.x < p.max_n) p.counts[(size_t)threadIdx.x * p.n_slices + k] = cnt[threadIdx.x];
}

// single workgroup: counts -> exclusive offsets (order: period, slice), total behind them
__global__ __launch_bounds__(1024) void region_scan_kernel(RegionParams p)
{
    __shared__ int64_t s_w[16];
    const int t = threadIdx.x;
    const int64_t m = (int64_t)p.max_n * p.n_slices;
    const int64_t per = (m + 1023) / 1024;
    const int64_t a = (int64_t)t * per, b = (a + per < m) ? a + per : m;
    int64_t l = 0, total;
    for (int64_t k = a; k < b; k++) l += p.counts[k];
    int64_t acc = block_scan_1024(l, s_w, &total);
    if (t == 0) p.counts[m] = total;
    for (int64_t k = a; k < b; k++) { const int64_t v = p.counts[k]; p.counts[k] = acc; acc += v; }
}

// workgroup per (slice, period): ordered compaction of the starts
__global__ __launch_bounds__(256) void region_emit_kernel(RegionParams p)
{
    const int k = blockIdx.x, n = blockIdx.y;
    const int64_t off = p.seq_off[k];
    const int len = (int)(p.seq_off[k + 1] - off);
    const uint8_t *plane = p.planes + (size_t)p.max_n * off + (size_t)n * len;
    int64_t w = p.counts[(size_t)n * p.n_slices + k];
    __shared__ int s_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < len; base += 256) {
        const int pos = base + threadIdx.x;
        const uint8_t v = pos < len ? plane[pos] : 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64((v & 128) != 0);
        __syncthreads();                 // the previous round's readers are done
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
        for (int q = 0; q < 4; q++) { before += q < wave ? s_w[q] : 0; total += s_w[q]; }
        if (v & 128) {
            const int64_t at = w + before + __popcll(m & ((1ull << lane) - 1ull));
            p.out_pos[at] = pos;
            p.out_reps[at] = v & 127;
        }
        w += total;
    }
}

}  // namespace npore
