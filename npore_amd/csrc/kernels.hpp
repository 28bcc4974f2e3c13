// kernels.hpp -- gfx950 (CDNA4, wave64) kernels of the realignment path.
//
//   fill_kernel<NG>   one wavefront per chunk.  Band column c of the current
//                     anti-diagonal lives in lane c / NG, register slot c % NG
//                     (NG consecutive columns per lane), so "column +-1" is a
//                     register rename inside a lane and one DPP wave shift at the
//                     lane boundary.  Per step (anti-diagonal) each cell needs its
//                     top / left / diagonal neighbours (previous two
//                     anti-diagonals: registers) and, for the n-polymer LEN/SHR
//                     states, values from up to 6 anti-diagonals back (LDS ring of
//                     NS rows).  The read/reference annotation words travel
//                     through the lanes systolically: an 'I' step of the input
//                     path shifts the read words one column up, a 'D' step shifts
//                     the reference words one column down; the word entering at
//                     the band edge comes from a 64-entry per-wave queue register.
//                     The only per-cell HBM traffic is one 32-bit traceback word.
//   traceback_kernel  one lane per chunk: follows MAT.TYP/MAT.RUN words
//                     (reference src/aln.pyx:670-742), writes ops right-aligned
//                     into the chunk's output slot.
//   gather_kernel     one workgroup per read: concatenates its chunks' op
//                     strings into the caller's output buffer (src/aln.pyx:742).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "cell.hpp"
#include "layout.hpp"

namespace npore {

constexpr int NP_LT = 32;  // LDS copy of np_scores covers ref length < NP_LT ...
constexpr int NP_CT = 64;  // ... and call length < NP_CT (powers of two: shifts and masks); else global memory
constexpr int XCH_WORDS = 12;   // 0-4 last cell, 5-11 first cell   // per wave, per parity: boundary cells handed to the neighbour waves

// history ring rows.  One wave per chunk: row b overwrites row b-6 after this wave
// has read it (LDS ops of a wave are in order).  Several waves per chunk: a wave may
// be one anti-diagonal ahead of its neighbours (it starts b+1 once they finished b),
// so row b+1 must not land on a row (b..b-5) a neighbour may still be reading: 7 rows.
__host__ __device__ constexpr int ring_rows(int nw) { return nw > 1 ? 7 : 6; }

struct KParams {
    const ChunkDesc *descs;
    const int32_t *sched;   // chunk slot -> chunk index (largest chunks first)
    const int32_t *n_chunks;   // device-side count (kernels are launched over an upper bound)
    const uint8_t *steps;
    const int32_t *inss;
    const uint32_t *seqw;
    const uint4 *refw;      // x, y and the two pre-decoded SHR candidates (layout.hpp)
    const uint2 *refl;      // 8 bytes per reference position
    uint32_t *tb;
    const float *sub_scores;  // [5][5]
    const float *np_scores;   // [max_n][max_l+1][max_l+1]
    int max_n, max_l;
    int r;
    int tbstride;
    int lstr;               // history row stride per column group: ceil((2r+1)/NG)
    int rwin;               // reference-L window entries (power of two)
    float indel_start, indel_extend;
};

// LDS floats: shared score tables + per chunk (history ring, reference-L window, exchange)
static inline size_t chunk_lds_floats(int nw, int ng, int lstr, int rwin)
{
    return (size_t)4 * ring_rows(nw) * ng * lstr + 2 * (size_t)rwin + (nw > 1 ? 2 * nw * XCH_WORDS + 8 : 0);
}
static inline size_t fill_lds_floats(int nw, int ng, int chunks, int lstr, int rwin)
{
    return (size_t)MAX_PERIOD * NP_LT * NP_CT + 64 + (size_t)chunks * chunk_lds_floats(nw, ng, lstr, rwin);
}

// value of the previous / next lane (lane 0 / 63 get 0)
__device__ __forceinline__ uint32_t lane_prev(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t lane_next(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ float lane_prev(float v) { return __uint_as_float(lane_prev(__float_as_uint(v))); }
__device__ __forceinline__ float lane_next(float v) { return __uint_as_float(lane_next(__float_as_uint(v))); }

// tell the compiler a value is wave-uniform (it then lives in SGPRs and conditions on it
// become scalar branches instead of exec-mask juggling)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni(int64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <int NG, int NSR>
struct DevEnv {
    const float *lds_sub;     // [8][8] padded copy of sub_scores
    const float *lds_np;      // [6][NP_LT][NP_CT]
    const float *g_np;        // full table in global memory
    const uint8_t *win;       // LDS window of reference L bytes, 8 per position
    const HistCell *hist;     // LDS: [NSR][hw] 16-byte records
    int np_dim, clampv, slot, lstr, hw, wmask;

    __device__ __forceinline__ int colidx(int col) const
    {
        if constexpr (NG == 1) return col;
        else return (col % NG) * lstr + col / NG;
    }
    __device__ __forceinline__ HistCell h_cell(int n, int col) const
    {
        int s = slot - n;
        s += (s < 0) ? NSR : 0;
        return hist[__umul24((unsigned)s, (unsigned)hw) + colidx(col)];   // 24-bit multiply: full rate
    }
    __device__ __forceinline__ float sub(uint32_t s, uint32_t r) const { return lds_sub[s * 8 + r]; }
    __device__ __forceinline__ float np_lds(int row, int call) const { return lds_np[(row << 6) + call]; }
    __device__ __forceinline__ int clamp() const { return clampv; }
    __device__ __forceinline__ int refl(int j, int n_idx) const { return win[(j & wmask) * 8 + n_idx]; }
    __device__ __forceinline__ bool any(bool x) const { return __builtin_amdgcn_ballot_w64(x) != 0ull; }
    template <int K>
    __device__ __forceinline__ void np_many(const int (&n_idx)[K], const int (&a)[K], const int (&b)[K],
                                            const bool (&active)[K], float (&out)[K]) const
    {
        // Always an LDS read (ds_read); lengths beyond the LDS copy are rare and patched
        // from global memory under a wave-uniform branch.  (Selecting between an LDS and a
        // global *pointer* would turn every lookup into a flat load, whose completion
        // wait also drains the outstanding traceback stores.)
        bool oot[K], anyoot = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
            oot[k] = active[k] && (((unsigned)a[k] >= (unsigned)NP_LT) || ((unsigned)b[k] >= (unsigned)NP_CT));
            anyoot |= oot[k];
            out[k] = lds_np[((n_idx[k] * NP_LT + (a[k] & (NP_LT - 1))) << 6) + (b[k] & (NP_CT - 1))];
            asm volatile("" : "+v"(out[k]));   // keep this a ds_read: do not fold it with the global load below
        }
        static_assert(NP_CT == 64, "index uses << 6");
        if (__builtin_amdgcn_ballot_w64(anyoot) != 0ull) {
#pragma unroll
            for (int k = 0; k < K; k++)
                if (oot[k]) {
                    out[k] = g_np[((size_t)n_idx[k] * np_dim + a[k]) * np_dim + b[k]];
                    // consume the value HERE: otherwise the compiler parks its s_waitcnt vmcnt(0) at the
                    // join below, where it would run on every pass and drain the traceback stores
                    asm volatile("" : "+v"(out[k]));
                }
        }
    }
};

// NW waves per chunk (each owns 64*NG consecutive band columns), NG columns per lane.
template <int NW, int NG, int MAXT>
__global__ __launch_bounds__(MAXT) void fill_kernel(KParams p)
{
    constexpr int NSR = ring_rows(NW);
    constexpr int WPW = 64 * NG;          // columns per wave
    constexpr int WPT = NW * WPW;         // physical columns per chunk
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *lds_np = lds;
    float *lds_sub = lds + MAX_PERIOD * NP_LT * NP_CT;
    const int wave = uni((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // Role-major numbering: consecutive waves (which the hardware deals round-robin over the CU's four
    // SIMDs) belong to DIFFERENT chunks, so that each SIMD hosts a mix of roles -- the last wave of a
    // chunk usually has few live columns (r=100: 9 of 64) and would otherwise leave one SIMD idle
    // while the other three carry all the full waves.
    const int cpg = (int)(blockDim.x >> 6) / NW;   // chunks per workgroup
    const int cw = wave / cpg;            // wave within the chunk
    const int cg = wave % cpg;            // chunk within the workgroup
    const int hw = NG * p.lstr;
    float *chunk_lds = lds_sub + 64 + (size_t)cg * (4 * NSR * hw + 2 * p.rwin + (NW > 1 ? 2 * NW * XCH_WORDS + 8 : 0));
    HistCell *hist = reinterpret_cast<HistCell *>(chunk_lds);
    uint2 *win = reinterpret_cast<uint2 *>(chunk_lds + 4 * NSR * hw);
    uint32_t *xchg = reinterpret_cast<uint32_t *>(chunk_lds + 4 * NSR * hw + 2 * p.rwin);   // [2][NW][XCH_WORDS]
    int *prog = reinterpret_cast<int *>(xchg + 2 * NW * XCH_WORDS);          // [NW] anti-diagonals completed

    // workgroup-shared tables
    const int np_dim = p.max_l + 1;
    for (int idx = threadIdx.x; idx < MAX_PERIOD * NP_LT * NP_CT; idx += blockDim.x) {
        const int n = idx / (NP_LT * NP_CT), a = (idx / NP_CT) % NP_LT, b = idx % NP_CT;
        lds_np[idx] = (n < p.max_n && a < np_dim && b < np_dim) ? p.np_scores[((size_t)n * np_dim + a) * np_dim + b] : 0.0f;
    }
    if (threadIdx.x < 64)
        lds_sub[threadIdx.x] = ((threadIdx.x >> 3) < 5 && (threadIdx.x & 7) < 5)
                                   ? p.sub_scores[(threadIdx.x >> 3) * 5 + (threadIdx.x & 7)] : 0.0f;
    __syncthreads();

    const int slot_id = blockIdx.x * cpg + cg;
    if (slot_id >= *p.n_chunks) return;    // hardware barriers only count waves that are still alive
    ChunkDesc d = p.descs[uni(p.sched[slot_id])];
    d.brk = uni(d.brk); d.nrows = uni(d.nrows); d.row0 = uni(d.row0); d.col0 = uni(d.col0);
    d.drows = uni(d.drows); d.dcols = uni(d.dcols); d.plain_lo = uni(d.plain_lo); d.plain_hi = uni(d.plain_hi);
    d.steps_off = uni(d.steps_off); d.seqw_off = uni(d.seqw_off); d.refw_off = uni(d.refw_off); d.tb_off = uni(d.tb_off);
    const int r = p.r;

    DevEnv<NG, NSR> env;
    env.lds_sub = lds_sub;
    env.lds_np = lds_np;
    env.g_np = p.np_scores;
    env.win = reinterpret_cast<const uint8_t *>(win);
    env.hist = hist;
    env.np_dim = np_dim;
    env.clampv = p.max_l - 1;
    env.slot = 0;
    env.lstr = p.lstr;
    env.hw = hw;
    env.wmask = p.rwin - 1;

    const uint32_t *seqw_g = p.seqw + d.seqw_off;
    const uint4 *refw_g = p.refw + d.refw_off;
    const uint2 *refl_g = p.refl + d.refw_off;
    const uint8_t *steps_g = p.steps + d.steps_off + d.brk;   // steps_g[k] = step from local row k to k+1
    uint32_t *tb_g = p.tb + d.tb_off;
    const int col0w = cw * WPW;           // first column of this wave

    // per-cell state of the previous anti-diagonal
    float matv[NG], insv[NG], delv[NG], LMv[NG], TMv[NG];
    uint32_t R1[NG], R2[NG], LT[NG];   // matrun|insrun<<16, matrun|delrun<<16, LMrun|TMrun<<16
    uint32_t seqw[NG], refx[NG], refy[NG], rc0[NG], rc1[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        matv[g] = insv[g] = delv[g] = LMv[g] = TMv[g] = 0.0f;
        R1[g] = R2[g] = LT[g] = 0u;
        const int col = col0w + lane * NG + g;
        const int i = r - col, j = col - r;
        seqw[g] = (i >= 0 && i <= d.drows) ? seqw_g[i] : SEQW_SENTINEL;
        uint4 rw = (j >= 0 && j <= d.dcols) ? refw_g[j] : make_uint4(REFW_SENTINEL, 0u, 0u, 0u);
        refx[g] = rw.x;
        refy[g] = rw.y;
        rc0[g] = rw.z;
        rc1[g] = rw.w;
    }
    // queues of words that will enter at column 0 (read; first wave) / column WPT-1 (reference; last wave)
    int sq_base = r + 1;              // next read index entering at column 0 is ins_l + r
    int rq_base = WPT - r;            // next reference index entering at column WPT-1 is del_l + WPT-1 - r
    uint32_t seq_q = SEQW_SENTINEL;
    uint4 ref_q = make_uint4(REFW_SENTINEL, 0u, 0u, 0u);
    if (cw == 0) {
        const int i = sq_base + lane;
        seq_q = (i <= d.drows) ? seqw_g[i] : SEQW_SENTINEL;
    }
    // reference-L window: positions [0, wfill) are resident (modulo rwin); kept by the last wave
    int wfill = 0;
    if (cw == NW - 1) {
        const int j = rq_base + lane;
        ref_q = (j >= 0 && j <= d.dcols) ? refw_g[j] : make_uint4(REFW_SENTINEL, 0u, 0u, 0u);
    }
    while (r + 32 >= wfill) {
        if (cw == NW - 1) {
            const int j = wfill + lane;
            win[j & env.wmask] = (j <= d.dcols) ? refl_g[j] : make_uint2(0u, 0u);
        }
        wfill += 64;
    }
    if constexpr (NW > 1) {
        if (lane == 0) prog[cw] = 0;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    StepInfo st;
    st.r = r;
    st.drows = d.drows;
    st.dcols = d.dcols;
    st.indel_start = p.indel_start;
    st.indel_extend = p.indel_extend;
    st.b_local = 0;
    st.ins_l = 0;
    st.del_l = 0;
    st.hist6 = 0;
    // input-path steps, 64 per coalesced load, one block prefetched (the buffer is padded)
    unsigned long long stepmask = __builtin_amdgcn_ballot_w64(steps_g[lane] != 0);
    unsigned long long nextmask = __builtin_amdgcn_ballot_w64(steps_g[64 + lane] != 0);
    const int lpos = cw * 64 + lane;          // lane position across the chunk's waves
    const int tcol = col0w + lane * NG;       // first band column of this lane
    const bool hist_lane = lpos < p.lstr;     // columns beyond the band are never read back
    const bool has_hi_edge = (2 * r >= col0w) && (2 * r < col0w + WPW);

    // One anti-diagonal.  MODE 0: first row of the chunk (no neighbours), 1: the input path
    // stepped 'I' (read words move one column up, "left" is the previous lane), 2: 'D'
    // (reference words move one column down, "top" is the next lane).  The whole body is
    // instantiated per mode so that no register shuffling is needed where the modes meet.
    auto step = [&](auto mode_tag, auto role_tag) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;
        // ROLE: 0 = only wave of the chunk, 1 = first, 2 = middle, 3 = last (compile-time so that the
        // per-role code needs no joins inside the loop)
        constexpr int ROLE = decltype(role_tag)::value;
        constexpr bool IS_FIRST = (ROLE == 0 || ROLE == 1), IS_LAST = (ROLE == 0 || ROLE == 3);
        const int bl = st.b_local;
        if constexpr (NW > 1 && MODE != 0) {
            // Per-chunk hand-shake instead of a workgroup barrier: this wave may start anti-diagonal bl
            // once its two neighbour waves have finished bl-1 (they own the only columns it reads).
            // LDS requests of a wave are served in order, so a neighbour's progress word becomes
            // visible after the history / exchange words it wrote before it.
            for (;;) {
                // relaxed workgroup-scope atomics keep these plain LDS reads (a volatile access would
                // become a flat system-scope load with a vmcnt(0) wait)
                const int a = !IS_FIRST ? __hip_atomic_load(&prog[cw - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0x7fffffff;
                const int b = !IS_LAST ? __hip_atomic_load(&prog[cw + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0x7fffffff;
                if (uni((a < b ? a : b)) >= bl) break;
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
        }
        // boundary cells written by the neighbour waves at the end of the previous step
        const uint32_t *xin = xchg + ((bl + 1) & 1) * (NW * XCH_WORDS);
        CellIn in[NG];
        if constexpr (MODE == 1) {
            float pm = lane_prev(matv[NG - 1]), pd = lane_prev(delv[NG - 1]);
            uint32_t pr = lane_prev(R2[NG - 1]);
            uint32_t ps = lane_prev(seqw[NG - 1]);
            if constexpr (IS_FIRST) {
                // word for row ins_l + r enters at column 0
                if (st.ins_l + r - sq_base >= 64) {   // uniform
                    sq_base += 64;
                    const int i = sq_base + lane;
                    seq_q = (i <= d.drows) ? seqw_g[i] : SEQW_SENTINEL;
                    asm volatile("" : "+v"(seq_q));   // wait for the reload inside this rare branch (see np_many)
                }
                const uint32_t incoming = (uint32_t)__builtin_amdgcn_readlane((int)seq_q, (st.ins_l + r - sq_base) & 63);
                ps = (lane == 0) ? incoming : ps;
            } else {
                const uint32_t *xl = xin + (cw - 1) * XCH_WORDS;   // last cell of the wave below (broadcast reads)
                const uint32_t x0 = xl[0], x1 = xl[1], x2 = xl[2], x3 = xl[3];
                pm = (lane == 0) ? __uint_as_float(x0) : pm;
                pd = (lane == 0) ? __uint_as_float(x1) : pd;
                pr = (lane == 0) ? x2 : pr;
                ps = (lane == 0) ? x3 : ps;
            }
#pragma unroll
            for (int g = NG - 1; g >= 0; g--) {
                in[g].topM = matv[g]; in[g].topI = insv[g]; in[g].topIrun = (int)(R1[g] >> 16);
                in[g].leftM = g ? matv[g - 1] : pm;
                in[g].leftD = g ? delv[g - 1] : pd;
                const uint32_t lr = g ? R2[g - 1] : pr;
                in[g].leftDrun = (int)(lr >> 16);
                in[g].diagM = LMv[g];
                in[g].diagMrun = (int)(LT[g] & 0xFFFFu);
                LT[g] = (lr & 0xFFFFu) | (R1[g] << 16);
                seqw[g] = g ? seqw[g - 1] : ps;
            }
        } else if constexpr (MODE == 2) {
            float nm = lane_next(matv[0]), ni = lane_next(insv[0]);
            uint32_t nr = lane_next(R1[0]);
            uint32_t nx = lane_next(refx[0]), ny = lane_next(refy[0]);
            uint32_t nc0 = lane_next(rc0[0]), nc1 = lane_next(rc1[0]);
            if constexpr (IS_LAST) {
                // word for col del_l + WPT-1 - r enters at column WPT-1
                if (st.del_l + WPT - 1 - r - rq_base >= 64) {
                    rq_base += 64;
                    const int j = rq_base + lane;
                    ref_q = (j >= 0 && j <= d.dcols) ? refw_g[j] : make_uint4(REFW_SENTINEL, 0u, 0u, 0u);
                    asm volatile("" : "+v"(ref_q.x), "+v"(ref_q.y), "+v"(ref_q.z), "+v"(ref_q.w));   // wait inside the rare branch
                }
                const int ql = (st.del_l + WPT - 1 - r - rq_base) & 63;
                const uint32_t inx = (uint32_t)__builtin_amdgcn_readlane((int)ref_q.x, ql);
                const uint32_t iny = (uint32_t)__builtin_amdgcn_readlane((int)ref_q.y, ql);
                const uint32_t inz = (uint32_t)__builtin_amdgcn_readlane((int)ref_q.z, ql);
                const uint32_t inw = (uint32_t)__builtin_amdgcn_readlane((int)ref_q.w, ql);
                nx = (lane == 63) ? inx : nx;
                ny = (lane == 63) ? iny : ny;
                nc0 = (lane == 63) ? inz : nc0;
                nc1 = (lane == 63) ? inw : nc1;
            } else {
                const uint32_t *xf = xin + (cw + 1) * XCH_WORDS + 5;   // first cell of the wave above
                const uint32_t x0 = xf[0], x1 = xf[1], x2 = xf[2], x3 = xf[3], x4 = xf[4], x5 = xf[5], x6 = xf[6];
                nm = (lane == 63) ? __uint_as_float(x0) : nm;
                ni = (lane == 63) ? __uint_as_float(x1) : ni;
                nr = (lane == 63) ? x2 : nr;
                nx = (lane == 63) ? x3 : nx;
                ny = (lane == 63) ? x4 : ny;
                nc0 = (lane == 63) ? x5 : nc0;
                nc1 = (lane == 63) ? x6 : nc1;
            }
            if (st.del_l + r + 32 >= wfill) {   // keep the L window ahead of the band (32 positions of slack)
                if constexpr (IS_LAST) {
                    const int j = wfill + lane;
                    win[j & env.wmask] = (j <= d.dcols) ? refl_g[j] : make_uint2(0u, 0u);
                }
                wfill += 64;
            }
#pragma unroll
            for (int g = 0; g < NG; g++) {
                in[g].leftM = matv[g]; in[g].leftD = delv[g]; in[g].leftDrun = (int)(R2[g] >> 16);
                in[g].topM = (g < NG - 1) ? matv[g + 1] : nm;
                in[g].topI = (g < NG - 1) ? insv[g + 1] : ni;
                const uint32_t tr = (g < NG - 1) ? R1[g + 1] : nr;
                in[g].topIrun = (int)(tr >> 16);
                in[g].diagM = TMv[g];
                in[g].diagMrun = (int)(LT[g] >> 16);
                LT[g] = (R2[g] & 0xFFFFu) | (tr << 16);
                refx[g] = (g < NG - 1) ? refx[g + 1] : nx;
                refy[g] = (g < NG - 1) ? refy[g + 1] : ny;
                rc0[g] = (g < NG - 1) ? rc0[g + 1] : nc0;
                rc1[g] = (g < NG - 1) ? rc1[g + 1] : nc1;
            }
        } else {
#pragma unroll
            for (int g = 0; g < NG; g++) {
                in[g].topM = in[g].topI = in[g].leftM = in[g].leftD = in[g].diagM = 0.0f;
                in[g].topIrun = in[g].leftDrun = in[g].diagMrun = 0;
            }
        }
#pragma unroll
        for (int g = 0; g < NG; g++) {
            in[g].c = tcol + g;
            in[g].seqw = seqw[g];
            in[g].refx = refx[g];
            in[g].refy = refy[g];
            in[g].sc0 = rc0[g];
            in[g].sc1 = rc1[g];
        }

        CellOut o[NG];
        // band-edge cells (columns 0 and 2r; reference src/aln.pyx:502-507: every state = 100*(b_row+1),
        // TYP = MAT, RUN = 0).  With one column per lane only three values of an edge cell are ever read
        // (by its one in-band neighbour), so only those are patched below; the traceback kernel treats edge columns as "run 0" itself.
        constexpr bool EDGE_PATCH = (NG == 1);
        if (bl >= d.plain_lo && bl < d.plain_hi) cells_update<NG, true, !EDGE_PATCH>(env, st, in, o);    // == step_is_plain(st)
        else cells_update<NG, false, !EDGE_PATCH>(env, st, in, o);

        uint32_t tbw[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            LMv[g] = in[g].leftM;
            TMv[g] = in[g].topM;
            matv[g] = o[g].matv;
            insv[g] = o[g].insv;
            delv[g] = o[g].delv;
            R1[g] = (uint32_t)o[g].matrun | ((uint32_t)o[g].insrun << 16);
            R2[g] = (uint32_t)o[g].matrun | ((uint32_t)o[g].delrun << 16);
            if constexpr (EDGE_PATCH) {
                const float e = (float)(100 * (bl + 1));
                if constexpr (IS_FIRST) {      // column 0 is lane 0 of the first wave; read as a LEFT neighbour
                    matv[0] = (lane == 0) ? e : matv[0];
                    delv[0] = (lane == 0) ? e : delv[0];
                    R2[0] = (lane == 0) ? 0u : R2[0];
                }
                if (has_hi_edge) {             // the wave holding column 2r (wave-uniform); read as a TOP neighbour
                    const bool is_edge = (tcol == 2 * r);
                    matv[0] = is_edge ? e : matv[0];
                    insv[0] = is_edge ? e : insv[0];
                    R1[0] = is_edge ? 0u : R1[0];
                }
            }
            // the row's spare last word carries inss[b] for the traceback
            tbw[g] = (tcol + g == p.tbstride - 1) ? (uint32_t)(d.row0 + st.ins_l) : o[g].tb;
            if (hist_lane)
                hist[env.slot * hw + g * p.lstr + lpos] =
                    HistCell{o[g].matv, o[g].lenstart, o[g].shrstart,
                             (uint32_t)o[g].lenrun_h | ((uint32_t)o[g].shrrun_h << 16)};
        }
        if constexpr (NW > 1) {
            // boundary cells for the neighbour waves: words 0-4 from the last lane, 5-9 from the first
            if (lane == 0 || lane == 63) {
                const bool first = (lane == 0);
                uint32_t *xout = xchg + (bl & 1) * (NW * XCH_WORDS) + cw * XCH_WORDS + (first ? 5 : 0);
                xout[0] = __float_as_uint(first ? matv[0] : matv[NG - 1]);
                xout[1] = __float_as_uint(first ? insv[0] : delv[NG - 1]);
                xout[2] = first ? R1[0] : R2[NG - 1];
                xout[3] = first ? refx[0] : seqw[NG - 1];
                xout[4] = refy[0];
                if (first) { xout[5] = rc0[0]; xout[6] = rc1[0]; }
            }
        }
        // one traceback word per cell, NG consecutive words per lane (tbstride is a multiple of 4)
        uint32_t *trow = tb_g + (size_t)bl * p.tbstride;
        if constexpr (NG == 1) {
            if (tcol < p.tbstride) trow[tcol] = tbw[0];
        } else if constexpr (NG == 2) {
            if (tcol < p.tbstride) *reinterpret_cast<uint2 *>(trow + tcol) = make_uint2(tbw[0], tbw[1]);
        } else {
#pragma unroll
            for (int q = 0; q < NG / 4; q++) {
                const int col = tcol + 4 * q;
                if (col < p.tbstride)
                    *reinterpret_cast<uint4 *>(trow + col) = make_uint4(tbw[4 * q], tbw[4 * q + 1], tbw[4 * q + 2], tbw[4 * q + 3]);
            }
        }
        // publish progress after this step's LDS writes (same in-order LDS queue); never vmcnt:
        // the traceback stores above must stay in flight
        if constexpr (NW > 1) {
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_store(&prog[cw], bl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    auto run = [&](auto role_tag) __attribute__((always_inline)) {
        step(std::integral_constant<int, 0>{}, role_tag);
        int left = 64;                         // steps left in stepmask
        for (int bl = 1; bl < d.nrows; bl++) {
            if (left == 0) {                   // step bl-1 leads from local row bl-1 to bl
                stepmask = nextmask;
                nextmask = __builtin_amdgcn_ballot_w64(steps_g[bl - 1 + 64 + lane] != 0);
                left = 64;
            }
            const int I = (int)(stepmask & 1ull);
            stepmask >>= 1;
            left--;
            st.b_local = bl;
            st.ins_l += I;
            st.del_l = bl - st.ins_l;
            st.hist6 = ((st.hist6 << 1) | (uint32_t)I) & 63u;
            env.slot = (env.slot + 1 == NSR) ? 0 : env.slot + 1;
            if (I) step(std::integral_constant<int, 1>{}, role_tag);
            else step(std::integral_constant<int, 2>{}, role_tag);
        }
    };
    // the wave's role within its chunk decides where annotation words and boundary cells come from
    if constexpr (NW == 1) run(std::integral_constant<int, 0>{});
    else if (cw == 0) run(std::integral_constant<int, 1>{});
    else if (cw == NW - 1) run(std::integral_constant<int, 3>{});
    else run(std::integral_constant<int, 2>{});
}

// ---------------------------------------------------------------------------
struct TParams {
    const ChunkDesc *descs;
    const int32_t *n_chunks;
    const uint32_t *tb;
    const uint8_t *seqs, *refs;
    uint8_t *chunk_out;        // per-chunk slots, ops right-aligned
    int32_t *chunk_len;        // ops emitted
    int32_t *chunk_status;
    int r;
    int tbstride;
};

// One wavefront per chunk.  Every hop of the traceback needs the cell's word and the
// band position of its anti-diagonal (inss[b]); the fill kernel stores the latter in
// the row's spare last word, so ONE coalesced load of the whole row (16 bytes per
// lane) serves the hop, and the next row is requested before the ops of the current
// run are emitted (lane-parallel), so emission overlaps the load latency.
template <int NL>   // uint4 loads per lane covering a row: tbstride <= 256 * NL
__global__ __launch_bounds__(64) void traceback_kernel(TParams p)
{
    const int k = blockIdx.x;
    if (k >= *p.n_chunks) return;
    const int lane = threadIdx.x;
    const ChunkDesc d = p.descs[k];
    const uint32_t *tb = p.tb + d.tb_off;
    const uint8_t *seq = p.seqs + d.seq_off, *ref = p.refs + d.ref_off;
    uint8_t *out = p.chunk_out + d.out_off;
    const int W = 2 * p.r + 1, stride = p.tbstride;
    int a_row = d.row0 + d.drows, a_col = d.col0 + d.dcols;
    int pos = d.out_cap;   // ops are written backwards
    int status = 0;

    uint4 row[NL];
    auto load_row = [&](int bl) {
#pragma unroll
        for (int q = 0; q < NL; q++) {
            const int idx = (q * 64 + lane) * 4;
            row[q] = (idx < stride) ? *reinterpret_cast<const uint4 *>(tb + (size_t)bl * stride + idx)
                                    : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto word = [&](int col) -> uint32_t {   // col is wave-uniform
        uint4 v = row[0];
        if constexpr (NL > 1) { if ((col >> 8) & 1) v = row[NL - 1]; }
        const int comp = col & 3;
        const uint32_t sel = comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w;
        return (uint32_t)__builtin_amdgcn_readlane((int)sel, (col & 255) >> 2);
    };
    auto in_chunk = [&](int ar, int ac) {
        const int bl = ar + ac - d.brk;
        return ar >= d.row0 && ac >= d.col0 && bl >= 0 && bl < d.nrows;
    };

    if ((a_row > d.row0 || a_col > d.col0) && in_chunk(a_row, a_col)) load_row(a_row + a_col - d.brk);
    while (a_row > d.row0 || a_col > d.col0) {
        if (!in_chunk(a_row, a_col)) { status |= 16; break; }
        const int bc = (int)word(stride - 1) - a_row + p.r;     // inss[b] - a_row + r, src/aln.pyx:322-326
        if (bc < 0 || bc >= W) { status |= 16; break; }
        const uint32_t w = (bc == 0 || bc == W - 1) ? 0u : word(bc);   // band edge: TYP = MAT, RUN = 0 (src/aln.pyx:502-507)
        const int typ = (int)(w & 7u), run = (int)(w >> 3);     // src/aln.pyx:684-685
        if (run < 1) { status |= 4; break; }
        if (run > pos) { status |= 16; break; }
        int n_row = a_row, n_col = a_col, emit = run;
        uint8_t ch = 0;
        if (typ == T_LEN || typ == T_INS) { ch = 'I'; n_row -= run; }
        else if (typ == T_SHR || typ == T_DEL) { ch = 'D'; n_col -= run; }
        else if (typ == T_MAT) {
            const int lim = min(a_row - d.row0, a_col - d.col0);
            emit = run < lim ? run : lim;                          // diagonal steps that stay in the chunk
            n_row -= emit; n_col -= emit;
        } else { status |= 8; break; }
        // request the next row now; the emission below overlaps its latency
        if ((n_row > d.row0 || n_col > d.col0) && in_chunk(n_row, n_col)) load_row(n_row + n_col - d.brk);
        if (ch) {
            for (int q = lane; q < emit; q += 64) out[pos - 1 - q] = ch;
        } else {
            for (int q = lane; q < emit; q += 64)
                out[pos - 1 - q] = (ref[a_col - 1 - q] == seq[a_row - 1 - q]) ? '=' : 'X';   // src/aln.pyx:732-735
        }
        pos -= emit;
        if (emit < run) { status |= 16; break; }
        a_row = n_row;
        a_col = n_col;
    }
    if (lane == 0) {
        p.chunk_len[k] = d.out_cap - pos;
        p.chunk_status[k] = status;
    }
}

// ---------------------------------------------------------------------------
struct GParams {
    const ChunkDesc *descs;
    const int32_t *read_first_chunk;   // [n_reads+1]
    const uint8_t *chunk_out;
    const int32_t *chunk_len, *chunk_status;
    const int32_t *read_status_in;     // prep status per read (bad input)
    const int32_t *counters;
    uint8_t *out;
    const int64_t *out_off;            // [n_reads+1] in the caller's buffer
    int64_t *out_len;
    int32_t *status;
    int64_t read_base;                 // index of this group's first read in the caller's arrays
};

__global__ __launch_bounds__(256) void gather_kernel(GParams p)
{
    const int rd = blockIdx.x;
    const int64_t grd = p.read_base + rd;
    const int c0 = p.read_first_chunk[rd], c1 = p.read_first_chunk[rd + 1];
    int st = p.read_status_in[rd];
    int64_t total = 0;
    for (int c = c0; c < c1; c++) { total += p.chunk_len[c]; st |= p.chunk_status[c]; }
    const int64_t cap = p.out_off[grd + 1] - p.out_off[grd];
    if (st & 32) total = -1;
    else if (total > cap) { st |= 64; total = -1; }
    if (threadIdx.x == 0) { p.out_len[grd] = total; p.status[grd] = st; }
    if (total < 0) return;
    uint8_t *dst = p.out + p.out_off[grd];
    int64_t w = 0;
    for (int c = c0; c < c1; c++) {
        const ChunkDesc d = p.descs[c];
        const int len = p.chunk_len[c];
        const uint8_t *src = p.chunk_out + d.out_off + (d.out_cap - len);
        for (int q = threadIdx.x; q < len; q += blockDim.x) dst[w + q] = src[q];
        w += len;
    }
}

// exhaustive check of div_small on its domain: counts mismatches
__global__ void divcheck_kernel(unsigned long long *bad)
{
    const int run = blockIdx.x * blockDim.x + threadIdx.x;   // 0..65535
    unsigned long long b = 0;
    for (int n = 1; n <= MAX_PERIOD; n++) b += (div_small(run, n) != run / n);
    if (b) atomicAdd(bad, b);
}

// DPP direction self-test: out[l] = lane_prev(l), out[64+l] = lane_next(l)
__global__ void dpp_selftest_kernel(uint32_t *out)
{
    const uint32_t l = threadIdx.x;
    out[l] = lane_prev(l);
    out[64 + l] = lane_next(l);
}

}  // namespace npore
