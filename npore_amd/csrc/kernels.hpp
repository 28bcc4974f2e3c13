// kernels.hpp -- gfx950 (CDNA4, wave64) kernels of the realignment path.
//
//   fill_kernel<NW>   NW wavefronts per chunk, band column c of the current
//                     anti-diagonal in lane c % 64 of wave c / 64, so "column +-1"
//                     is one DPP wave shift (and one LDS exchange record where two
//                     waves meet).  Per step (anti-diagonal) each cell needs its
//                     top / left / diagonal neighbours (previous two
//                     anti-diagonals: registers) and, for the n-polymer LEN/SHR
//                     states, values from up to 6 anti-diagonals back (LDS ring of
//                     NSR rows).  The read/reference annotation words travel
//                     through the lanes systolically: an 'I' step of the input
//                     path shifts the read words one column up, a 'D' step shifts
//                     the reference words one column down; the word entering at
//                     the band edge comes from a 64-entry queue register (first /
//                     last wave) and crosses between waves in the exchange record.
//                     The only per-cell HBM traffic is one 32-bit traceback word.
//                     The launch is persistent: its groups of NW waves pull chunk
//                     after chunk from a device-side queue.
//                     What is the same for all lanes of a step (ring row, traceback row,
//                     100*b, progress count, record addresses) is kept in VECTOR registers:
//                     scalar instructions cost a wave about twice as much here, and the
//                     kernel runs at the vector issue rate (DESIGN.md sections 4.1, 6).
//   traceback_rows_kernel   one wavefront per chunk: follows MAT.TYP/MAT.RUN words
//                     (reference src/aln.pyx:670-742) and records the path as (type, length)
//                     runs; per hop one 64-column group of the anti-diagonal landed on and
//                     of the one below it, the hop loop written out in scalar instructions.
//   gather_scan /     per read: length and status of the output, where every chunk's ops go;
//   gather_kernel     one workgroup per chunk: expands the chunk's runs into the op
//                     string in the caller's output buffer (src/aln.pyx:719-742);
//                     <false>: without its LDS tile, to run beside a fill kernel.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "cell.hpp"
#if !defined(NPORE_FILL_ASM_INC)
#define NPORE_FILL_ASM_INC "fill_step_asm.inc"     // (measurement builds: another generated variant)
#endif
#include NPORE_FILL_ASM_INC
#include "layout.hpp"
#include "std_stream.hpp"

namespace npore {

static_assert(NP_LT == NP_CT - NP_C0 && (NP_LT & (NP_LT - 1)) == 0, "np_full's range test");
constexpr int MAX_WAVES_PER_CHUNK = 16;   // a workgroup of 1 024 threads: r <= 511
constexpr int XCH_WORDS = 12;   // per wave, per parity: boundary cells handed to the neighbour waves
                                // (words 0-3 last cell, 5-10 first cell)

// history ring rows.  One wave per chunk: row b overwrites row b-6 after this wave
// has read it (LDS ops of a wave are in order).  Several waves per chunk: a wave may
// be one anti-diagonal ahead of its neighbours (it starts b+1 once they finished b),
// so row b+1 must not land on a row (b..b-5) a neighbour may still be reading: 7 rows.
#if defined(NPORE_X_RING)
__host__ __device__ constexpr int ring_rows(int nw) { return nw > 1 ? NPORE_X_RING : 6; }      // (race hunt: more slack rows)
#else
__host__ __device__ constexpr int ring_rows(int nw) { return nw > 1 ? 7 : 6; }
#endif

struct KParams {
    const ChunkDesc *descs;
    const int32_t *sched;   // chunk slot -> chunk index (largest chunks first)
    const int32_t *n_chunks;   // device-side count (kernels are launched over an upper bound)
    int32_t *queue;            // next slot of the schedule to hand out (zeroed by read_scan_kernel)
    const uint8_t *steps;
    const int32_t *inss;
    const uint32_t *seqw;
    const uint4 *refw;      // x, y and the two pre-decoded SHR candidates (layout.hpp)
    const uint2 *refl;      // 8 bytes per reference position
    uint32_t *tb;
    uint32_t *dbg = nullptr;   // experiments: MAT.VAL of every cell, laid out like tb (NPORE_X_DBGMAT)
    const float *sub_scores;  // [5][5]
    const float *np_scores;   // [max_n][max_l+1][max_l+1]
    int max_n, max_l;
    int r;
    int tbstride;
    int hw;                 // history records per ring row: 2r+1 columns + HIST_PAD
    int rwin;               // reference-L window entries (power of two)
    float indel_start, indel_extend;
};

// LDS floats: shared score tables + per chunk (history ring, reference-L window, exchange)
__host__ __device__ static inline size_t chunk_lds_floats(int nw, int hw, int rwin)
{
    // history ring + reference-L window + (several waves per chunk) exchange records, progress words, slot ring
    return (size_t)4 * (ring_rows(nw) * hw + HIST_PAD) + 2 * (size_t)rwin + (nw > 1 ? 2 * nw * XCH_WORDS + MAX_WAVES_PER_CHUNK + 12 : 0);
}
static inline size_t fill_lds_floats(int nw, int chunks, int hw, int rwin)
{
    return (size_t)MAX_PERIOD * NP_LT * NP_CT + SUBT_ENTRIES + (size_t)chunks * chunk_lds_floats(nw, hw, rwin);
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
// per-lane lookup in a table spread over the lanes of a register: lane l reads tab[lane addr4/4]
__device__ __forceinline__ uint32_t lane_table(uint32_t addr4, uint32_t tab)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr4, (int)tab);
}

// LDS reads of the workgroup-shared score tables by ABSOLUTE LDS byte address.  The kernels here use dynamic LDS only,
// so the `lds` array starts at LDS address 0 (checked once per launch configuration on the host:
// hipFuncGetAttributes().sharedSizeBytes == 0, npore_api.cpp); going through the array symbol instead costs one
// `v_add_u32 v, <lds>, v` per lookup, because the symbol's value is only known at link time.
typedef const __attribute__((address_space(3))) float lds_cfloat;
__device__ __forceinline__ float lds_abs_f32(uint32_t byte_addr)
{
    return *reinterpret_cast<lds_cfloat *>(byte_addr);
}
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) int lds_i32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
constexpr uint32_t LDS_NP_BASE = 0u;                                      // [6][NP_LT][NP_CT] floats
constexpr uint32_t LDS_SUB_BASE = MAX_PERIOD * NP_LT * NP_CT * 4u;        // then the substitution table

#if defined(NPORE_EXPERIMENTS) && defined(NPORE_STATS)
__device__ unsigned long long g_npore_stats[16];
#endif
#if !defined(NPORE_EXPERIMENTS)
__device__ __forceinline__ void pad_hook(uint32_t, unsigned long long) {}      // (experiments.hpp in measurement builds)
#endif

template <int NSR>
struct DevEnv {
#if defined(NPORE_EXPERIMENTS) && defined(NPORE_STATS)
    __device__ __forceinline__ void count(int k) const { if ((threadIdx.x & 63) == 0) atomicAdd(&g_npore_stats[k], 1ull); }
#endif
    static constexpr bool LEN_ARITH = NSR != 6;   // several waves per chunk (ring_rows): cell.hpp, LEN filter
    static constexpr bool MIN3 = NSR != 6;      // (chunks of several waves: -0.5 % fill at r = 64 ... 200; a lone wave: +0.5 %)
    __device__ __forceinline__ float min3(float a, float b, float c) const
    {
        float r;      // (the instruction itself: fminf would first canonicalise its operands, one v_max each)
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
    }
    const char *lds_sub;      // [ref 8][seq 8][4] copy of sub_scores (layout.hpp SUBT_*)
    const char *lds_np;       // [6][NP_LT][NP_CT] floats (layout.hpp)
    const float *g_np;        // full table in global memory
    const uint8_t *win;       // LDS window of reference L bytes, 8 per position
    const char *hist_c;       // LDS: this lane's own band column in ring row 0
    const uint4 *refw_g;      // the chunk's reference words (rare re-reads)
    int np_dim, clampv, hw16, wmask, dcols;
    // tables over n, spread over the lanes: lane l holds the entry of n = l & 7
    uint32_t t_n, t_recip, t_msh, t_mmask;
    // lane l: byte offset from "own column, ring row 0" to column c - dI of row b-n, n = l & 7 and
    // dI = inss[b] - inss[b-n]; kept current by the step loop with one lane shift per anti-diagonal (fill_kernel)
    uint32_t tab_e;
    unsigned long long n0_lanes;     // lanes holding entry 0 of the lane tables

    struct Tab { uint32_t e; };
    __device__ __forceinline__ Tab step_tables(const StepInfo &) const { return Tab{tab_e}; }
    __device__ __forceinline__ HistCell h_shr(const Tab &tab, uint32_t n4, int) const
    {
        return *reinterpret_cast<const HistCell *>(hist_c + (int)lane_table(n4, tab.e));
    }
    __device__ __forceinline__ uint32_t h_off(const Tab &tab, uint32_t n4) const { return lane_table(n4, tab.e); }
    __device__ __forceinline__ HistCell h_shr_at(const Tab &, uint32_t off, uint32_t, int) const
    {
        return *reinterpret_cast<const HistCell *>(hist_c + (int)off);
    }
    // (only where a chunk is several waves in lock step: a wave on its own -- NSR == 6 -- is scheduled better
    // without: measured +2 % fill at r = 30 with the pins, -3 % at r = 100)
    static constexpr bool PIN = NSR != 6;
    template <class A, class B>
    __device__ __forceinline__ void pin(A &a, B &b) const { if constexpr (PIN) asm volatile("" : "+v"(a), "+v"(b)); }
    template <class A, class B, class C>
    __device__ __forceinline__ void pin(A &a, B &b, C &c) const { if constexpr (PIN) asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }
    template <class A, class B, class C, class D>
    __device__ __forceinline__ void pin(A &a, B &b, C &c, D &d) const
    {
        if constexpr (PIN) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    template <class A, class B, class C, class D, class E, class F>
    __device__ __forceinline__ void pin(A &a, B &b, C &c, D &d, E &e, F &f) const
    {
        if constexpr (PIN) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    }
    template <class T>
    __device__ __forceinline__ T opaque(T x) const { asm volatile("" : "+v"(x)); return x; }
    __device__ __forceinline__ HistCell h_len(const Tab &tab, uint32_t n4, int) const
    {
        return *reinterpret_cast<const HistCell *>(hist_c + (int)(lane_table(n4, tab.e) + (n4 << 2)));
    }
    __device__ __forceinline__ uint32_t recip(const Tab &, uint32_t n4) const { return lane_table(n4, t_recip); }
    __device__ __forceinline__ int mer_shift(const Tab &, uint32_t n4) const { return (int)lane_table(n4, t_msh); }
    __device__ __forceinline__ uint32_t mer_mask(const Tab &, uint32_t n4) const { return lane_table(n4, t_mmask); }
    __device__ __forceinline__ float sub(uint32_t seqw, uint32_t refx) const
    {
        // bits 2-9 of {refx, seqw} >> 25: ref[j-1] (3 bits) | seq[i-1] (3 bits) | 2 bits of seq[i-2] (don't care)
        return lds_abs_f32(LDS_SUB_BASE + (__builtin_amdgcn_alignbit(refx, seqw, 25) & 0x3FCu));
    }
    __device__ __forceinline__ float np_small(uint32_t dsc, int q) const
    {
        const uint32_t a = (dsc >> 15) & 0xFFFFu;
        // q >= L copies deleted: the call length L - 1 - q is negative -> the guard entry behind the row
        const uint32_t L = (dsc >> 8) & 0xFFu;      // bits 8-14; bit 15 is clear (the address field holds a multiple of 4)
        return lds_abs_f32(LDS_NP_BASE + a + 4u * min((uint32_t)q, L));
    }
    __device__ __forceinline__ int clamp() const { return clampv; }
    __device__ __forceinline__ int refl(int j, int n_idx) const { return win[(j & wmask) * 8 + n_idx]; }
    __device__ __forceinline__ uint32_t refy(int j) const
    {
        uint32_t y = 0u;
        if (j >= 0 && j <= dcols) {
            y = refw_g[j].y;
            asm volatile("" : "+v"(y));   // wait for it inside this rare branch (see np_full)
        }
        return y;
    }
    __device__ __forceinline__ bool any(bool x) const { return __builtin_amdgcn_ballot_w64(x) != 0ull; }
    __device__ __forceinline__ bool any2(bool a, bool b) const
    {
        return (__builtin_amdgcn_ballot_w64(a) & __builtin_amdgcn_ballot_w64(b)) != 0ull;
    }
    __device__ __forceinline__ float np_full(int n_idx, int a, int b, bool active) const
    {
        // Always an LDS read (ds_read); lengths beyond the LDS copy are rare and patched
        // from global memory under a wave-uniform branch.  (Selecting between an LDS and a
        // global *pointer* would turn every lookup into a flat load, whose completion
        // wait also drains the outstanding traceback stores.)
        const bool big = (unsigned)(a | b) >= (unsigned)NP_LT;     // a, b >= 0 and NP_LT == NP_CT - NP_C0 is a power of two
        const bool oot = active && big;
        float out = lds_abs_f32(LDS_NP_BASE + 4u * (uint32_t)(((n_idx * NP_LT + (a & (NP_LT - 1))) * NP_CT) + (NP_LT - 1) -
                                                              (b & (NP_LT - 1))));
        asm volatile("" : "+v"(out));   // keep this a ds_read: do not fold it with the global load below
        if (any2(active, big)) {
            if (oot) {
                out = g_np[((size_t)n_idx * np_dim + a) * np_dim + b];
                // consume the value HERE: otherwise the compiler parks its s_waitcnt vmcnt(0) at the
                // join below, where it would run on every pass and drain the traceback stores
                asm volatile("" : "+v"(out));
            }
        }
        return out;
    }
};

// value of the previous / next lane, with `edge` (a per-lane register, usually a broadcast value) kept in the lane
// that has no neighbour (lane 0 / 63): ONE instruction where a shift and a select used to be -- the DPP move
// leaves the destination's old contents in lanes without a valid source when bound_ctrl is off
__device__ __forceinline__ uint32_t lane_prev_or(uint32_t edge, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t lane_next_or(uint32_t edge, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ float lane_prev_or(float edge, float v)
{
    return __uint_as_float(lane_prev_or(__float_as_uint(edge), __float_as_uint(v)));
}
__device__ __forceinline__ float lane_next_or(float edge, float v)
{
    return __uint_as_float(lane_next_or(__float_as_uint(edge), __float_as_uint(v)));
}

// "progress word has reached target".  The words count anti-diagonals over all the chunks a group of waves has
// worked on in one launch: bounded by the traceback words a launch can hold (4 bytes x the band per anti-diagonal
// within the device's memory: < 2^30 anti-diagonals for every band), so a plain signed compare is safe
__device__ __forceinline__ bool reached(int word, int target) { return word >= target; }

// Ordering of the LDS hand-shakes between the waves of a group (progress words, slot ring, window-ready word): a
// workgroup-scope release fence before every publishing store, an acquire fence behind every polling load.
// -DNPORE_EXPERIMENTS -DNPORE_RELAXED_SYNC builds the shortcut of round 1 instead -- rely on the hardware serving the LDS requests
// of a wave in order, and only stop the COMPILER from reordering (an empty asm with a memory clobber).  Both were
// run over the same 40 000 fuzz reads against the oracle (tests/tools/ab_sync.py: equal) and timed: the fences
// cost nothing measurable (C2 fill 22.31 vs 22.27 ms), so they are the default.
#if defined(NPORE_EXPERIMENTS) && defined(NPORE_RELAXED_SYNC)
#define NPORE_PUBLISH_FENCE() asm volatile("" ::: "memory")
#define NPORE_OBSERVE_FENCE() asm volatile("" ::: "memory")
#else
#define NPORE_PUBLISH_FENCE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup")
#define NPORE_OBSERVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup")
#endif

constexpr int SLOT_RING = 8;    // chunk slots published by a group's first wave and not yet read by its last (<= NW - 1)

// NW waves per chunk ("a group"), each owning 64 consecutive band columns.  PERSISTENT: the launch holds as many
// workgroups as the GPU keeps resident, and every group of NW waves pulls its next chunk from a device-wide queue
// (the schedule lists the chunks largest first) the moment it has finished one, until the queue is empty -- no
// group waits for a sibling of its workgroup, no workgroup waits for a "round" to drain.
// Registers: four of these waves share a SIMD with whatever the NEXT batch runs beside them (npore_api.cpp: its
// preparation, the previous one's traceback), so the kernel is held to 112 of the SIMD's 512 / 4 = 128 vector
// registers -- amdgpu_num_vgpr counts in halves on this target (arch + acc registers) -- which leaves 64 for one
// light wave per SIMD (this round's annotate_wave_kernel is held to those 64).  The cap costs spills (NW >= 2: 59 vector
// registers, 168 bytes of scratch per lane; NW = 1: 16 / 68 -- hipcc -Rpass-analysis=kernel-resource-usage): none inside the
// assembly text, four stores + reloads around every ENTRY of the assembly (v97 - v100 are live across the statement
// that clobbers them; an entry is a hand-over, ~1 % of the steps) and two 8-byte reloads in the compiled rare step's
// per-period candidate loop -- under half a percent of the step time, and the price of the co-residency.
template <int NWT, int MAXT>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(4, 4), amdgpu_num_vgpr(56))) void fill_kernel(KParams p)
{
    // NWT = 0: the number of waves per chunk comes with the launch (one chunk per workgroup: bands of more than eight
    // waves, r = 256 ... 511 -- one instantiation for all of them; what depends on NW only as "one wave or several" stays
    // compile-time)
    constexpr bool MULTI = NWT != 1;
    constexpr int NSR = ring_rows(NWT ? NWT : 2);
    const int NW = NWT ? NWT : uni((int)(blockDim.x >> 6));
    const int WPT = NW * 64;              // physical columns per chunk
    // reference-L window (LDS, kept by the chunk's last wave): refilled WIN_STEP positions at a time once the band
    // comes within WIN_SLACK of its end.  Several waves: a whole wave's worth, and enough slack for the waves that
    // run behind the last one.  One wave: it refills for itself, so little of both does (host: fill_geometry)
    constexpr int WIN_STEP = NWT == 1 ? 32 : 64, WIN_SLACK = NWT == 1 ? 8 : 32;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *lds_np = lds;
    float *lds_sub = lds + MAX_PERIOD * NP_LT * NP_CT;
    const int wave = uni((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // Role-major numbering: consecutive waves (which the hardware deals round-robin over the CU's four
    // SIMDs) belong to DIFFERENT chunks.  With four chunks per workgroup every chunk has its waves on
    // ONE SIMD: whichever of them is on the critical path of the anti-diagonal gets that SIMD's whole
    // issue rate while its neighbours wait for it.  (Chunk-major numbering puts all the nearly empty last
    // waves -- r=100: 9 live columns of 64 -- on one SIMD: 15 % slower; spreading each chunk over the four
    // SIMDs with mixed roles: 12 % slower, the critical wave then competes with three busy strangers.)
    const int cpg = (int)(blockDim.x >> 6) / NW;   // chunks (groups) per workgroup
    const int cw = xp::CHUNKMAJOR ? wave % NW : wave / cpg;            // wave within the group
    const int cg = xp::CHUNKMAJOR ? wave / NW : wave % cpg;            // group within the workgroup
    // The middle waves of a chunk (all 64 lanes live, a neighbour wave on either side) are issued first when
    // several waves of the SIMD are ready: measured 1.5-2 % on the fill at NW = 3...7 (r = 70, 100, 140, 200)
    if constexpr (xp::PRIO == 0) {
        // (the step assembly drops a wave to priority 0 while it waits for a neighbour: gen_fill_asm.py polls)
        if constexpr (MULTI) {
            if (NW > 2 && cw != 0 && cw != NW - 1) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(1);
        }
    } else if constexpr (xp::PRIO == 1) {          // first wave highest
        const int pr = min(3, NW - 1 - cw);
        if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1);
    } else if constexpr (xp::PRIO == 2) {          // last wave highest
        const int pr = min(3, cw);
        if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1);
    } else if constexpr (xp::PRIO == 4) {          // middle 2, first 1, last 0
        if (NW > 2 && cw != 0 && cw != NW - 1) __builtin_amdgcn_s_setprio(2); else if (cw == 0 && MULTI) __builtin_amdgcn_s_setprio(1);
    }
    const int hw = p.hw;
    float *chunk_lds = lds_sub + SUBT_ENTRIES + (size_t)cg * chunk_lds_floats(NW, hw, p.rwin);
    HistCell *hist = reinterpret_cast<HistCell *>(chunk_lds) + HIST_PAD;     // row 0, column 0
    uint2 *win = reinterpret_cast<uint2 *>(chunk_lds + 4 * (NSR * hw + HIST_PAD));
    uint32_t *xchg = reinterpret_cast<uint32_t *>(chunk_lds + 4 * (NSR * hw + HIST_PAD) + 2 * p.rwin);   // [NW][2][XCH_WORDS]
    int *prog = reinterpret_cast<int *>(xchg + 2 * NW * XCH_WORDS);          // [NW] anti-diagonals completed, all chunks
    int *slotbox = prog + MAX_WAVES_PER_CHUNK;                                // [SLOT_RING] + generation word

    // workgroup-shared tables.  A score row is stored with the call length DEcreasing, so that "q more copies
    // deleted" is q entries UP from the address a column descriptor holds, and ends in one guard entry holding
    // the constant 100 for negative call lengths (layout.hpp)
    const int np_dim = p.max_l + 1;
    for (int idx = threadIdx.x; idx < MAX_PERIOD * NP_LT * NP_CT; idx += blockDim.x) {
        const int n = idx / (NP_LT * NP_CT), a = (idx / NP_CT) % NP_LT, b = NP_LT - 1 - idx % NP_CT;
        // (entry 0 -- period 1, L = 0, which no lookup uses: L = 0 is np_score's "100" -- is what the EMPTY column
        // descriptor 0 points at: +infinity, so that a lane without a candidate never wins, cell.hpp shr_small)
        lds_np[idx] = idx == 0 ? huge_f()
                    : b < 0   ? INF_F
                              : (n < p.max_n && a < np_dim && b < np_dim) ? p.np_scores[((size_t)n * np_dim + a) * np_dim + b] : 0.0f;
    }
    if constexpr (xp::ZEROLDS != 0) {      // (diagnostic: a defined value in every word of the chunks' LDS)
        float *base = lds_sub + SUBT_ENTRIES;
        const size_t nfl = (size_t)cpg * chunk_lds_floats(NW, hw, p.rwin);
        for (size_t idx = threadIdx.x; idx < nfl; idx += blockDim.x) base[idx] = xp::ZEROLDS == 1 ? 0.0f : __uint_as_float(0x7fc00000u);
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < SUBT_ENTRIES; idx += blockDim.x) {
        const int rb = idx >> 5, sb = (idx >> 2) & 7;
        lds_sub[idx] = (rb < 5 && sb < 5) ? p.sub_scores[sb * 5 + rb] : 0.0f;
    }
    const int r = p.r;
    const int lpos = cw * 64 + lane;          // band column of this lane
    const int tcol = lpos;
    // history: every record starts as "no candidate can come from here" (cell.hpp).  The band edges, the columns
    // beyond the band and the pad records either side of a row are never written and stay that way for the whole
    // launch; the band-interior columns are reset by their own lanes before every chunk
    for (int k = lpos - HIST_PAD; k < NSR * hw; k += WPT) hist[k] = hist_none();
    if constexpr (MULTI) {
        if (lane == 0) prog[cw] = 0;
        if (cw == 0 && lane <= SLOT_RING + 1) slotbox[lane] = 0;
    }
    __syncthreads();      // the only workgroup barrier: from here on the groups run on their own

    DevEnv<NSR> env;
    env.lds_sub = reinterpret_cast<const char *>(lds_sub);
    env.lds_np = reinterpret_cast<const char *>(lds_np);
    env.g_np = p.np_scores;
    env.win = reinterpret_cast<const uint8_t *>(win);
    env.hist_c = reinterpret_cast<const char *>(hist + tcol);
    env.np_dim = np_dim;
    env.clampv = p.max_l - 1;
    env.hw16 = hw * 16;
    env.wmask = p.rwin - 1;
    {
        const int nl = lane & 7;
        env.t_n = (uint32_t)nl;
        env.t_recip = recip16(nl);
        env.t_msh = nl ? 32u - 3u * (uint32_t)nl : 0u;
        env.t_mmask = (1u << (3 * nl)) - 1u;
    }
    const bool hist_lane = (tcol >= 1) && (tcol <= 2 * r - 1);     // band-interior columns leave history
    env.n0_lanes = __builtin_amdgcn_ballot_w64((lane & 7) == 0);
    const uint32_t tcol4 = (uint32_t)tcol * 4u;
    // (dynamic LDS starts at LDS address 0, see lds_abs_f32)
    const uint32_t hist_c_addr = (uint32_t)(reinterpret_cast<const char *>(hist + tcol) - reinterpret_cast<const char *>(lds));
    const uint32_t xchg_addr = (uint32_t)(reinterpret_cast<const char *>(xchg) - reinterpret_cast<const char *>(lds));
    // progress word of the wave below (the wave above's is two words on)
    uint32_t pnb_addr = (uint32_t)(reinterpret_cast<const char *>(prog + (cw - 1)) - reinterpret_cast<const char *>(lds));
    asm volatile("" : "+v"(pnb_addr));
    const int n_chunks = *p.n_chunks;
    const int dealt = (int)gridDim.x * cpg;      // slots handed out without the queue: one per group
    int pbase = 0;        // anti-diagonals of the chunks this group has finished (what prog[] counts from)
    int gen = 0;          // chunks this group has started

    for (;;) {
        // ---- next chunk slot of the schedule.  One wave of the group asks the queue; the others read its answer
        // from the group's slot ring in LDS (value released before the generation word, acquired behind the poll)
        int slot_id;
        if (gen == 0) {
            // the first chunk of every group is dealt like cards -- slot q of the schedule to workgroup q % grid,
            // group q / grid -- so that the heavy chunks of a batch smaller than the launch are spread over all
            // CUs and SIMDs instead of filling the first workgroups; the queue hands out what lies beyond
            slot_id = cg * (int)gridDim.x + (int)blockIdx.x;
        } else if constexpr (NWT == 1) {
            int v = 0;
            if (lane == 0) v = atomicAdd(p.queue, 1);
            slot_id = uni(v) + dealt;
        } else {
            if (cw == 0) {
                int v = 0;
                if (lane == 0) {
                    v = atomicAdd(p.queue, 1) + dealt;
                    __hip_atomic_store(&slotbox[gen & (SLOT_RING - 1)], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    NPORE_PUBLISH_FENCE();
                    __hip_atomic_store(&slotbox[SLOT_RING], gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                slot_id = uni(v);
            } else {
                for (;;) {
                    const int g = __hip_atomic_load(&slotbox[SLOT_RING], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (reached(uni(g), gen + 1)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                NPORE_OBSERVE_FENCE();
                slot_id = uni(__hip_atomic_load(&slotbox[gen & (SLOT_RING - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            }
        }
        gen++;
        if (slot_id >= n_chunks) break;     // the queue only grows: every wave of the group sees the same answer
        ChunkDesc d = p.descs[uni(p.sched[slot_id])];
        d.brk = uni(d.brk); d.nrows = uni(d.nrows); d.row0 = uni(d.row0); d.col0 = uni(d.col0);
        d.drows = uni(d.drows); d.dcols = uni(d.dcols); d.plain_lo = uni(d.plain_lo); d.plain_hi = uni(d.plain_hi);
        d.steps_off = uni(d.steps_off); d.seqw_off = uni(d.seqw_off); d.refw_off = uni(d.refw_off); d.tb_off = uni(d.tb_off);

        const uint32_t *seqw_g = p.seqw + d.seqw_off;
        const uint4 *refw_g = p.refw + d.refw_off;
        const uint2 *refl_g = p.refl + d.refw_off;
        const uint8_t *steps_g = p.steps + d.steps_off + d.brk;   // steps_g[k] = step from local row k to k+1
        uint32_t *tb_g = p.tb + d.tb_off;
        uint32_t *dbg_g = p.dbg ? p.dbg + d.tb_off : p.tb + d.tb_off;      // (experiments)
        (void)dbg_g;
        env.refw_g = refw_g;
        env.dcols = d.dcols;

        // per-cell state of the previous anti-diagonal
        float matv = 0.0f, insv = 0.0f, delv = 0.0f, LMv = 0.0f, TMv = 0.0f;
        uint32_t R1 = 0u, R2 = 0u;      // matrun|insrun<<16, matrun|delrun<<16
        uint32_t LMr = 0u, TMr = 0u;    // low half: MAT.RUN of the left / top neighbour of the previous step
        uint32_t seqw, refx, rc0, rc1;
        {
            const int i = r - tcol, j = tcol - r;
            seqw = (i >= 0 && i <= d.drows) ? seqw_g[i] : SEQW_SENTINEL;
            uint4 rw = (j >= 0 && j <= d.dcols) ? refw_g[j] : make_uint4(REFW_SENTINEL, 0u, 0u, 0u);
            refx = rw.x;
            rc0 = rw.z;
            rc1 = rw.w;
        }
        // queues of words that will enter at column 0 (read; first wave) / column WPT-1 (reference; last wave)
        int sq_base = r + 1;              // next read index entering at column 0 is ins_l + r
        int rq_base = WPT - r;            // next reference index entering at column WPT-1 is del_l + WPT-1 - r
        int sq_idx = 0, rq_idx = 0;       // lane of seq_q / ref_q that holds it
        uint32_t seq_q = SEQW_SENTINEL;
        uint4 ref_q = make_uint4(REFW_SENTINEL, 0u, 0u, 0u);
        if (cw == 0) {
            const int i = sq_base + lane;
            seq_q = (i <= d.drows) ? seqw_g[i] : SEQW_SENTINEL;
        }
        if (cw == NW - 1) {
            const int j = rq_base + lane;
            ref_q = (j >= 0 && j <= d.dcols) ? refw_g[j] : make_uint4(REFW_SENTINEL, 0u, 0u, 0u);
        }
        // input-path steps, 64 per coalesced load: window m holds the steps that lead to anti-diagonals 64m ... 64m+63
        // of the chunk, the one of anti-diagonal bl in bit bl & 63 (so the bit test needs no index arithmetic); one
        // window is fetched ahead (the buffer is padded behind; nothing precedes the first step of the first read)
        auto step_window = [&](int m) {
            const int k = 64 * m + lane - 1;
            return __builtin_amdgcn_ballot_w64(k >= 0 && steps_g[k] != 0);
        };
        unsigned long long stepmask = step_window(0);
        unsigned long long nextmask = step_window(1);

        // The group's LDS (history ring, L window, exchange records) still serves its waves until all of them have
        // finished the previous chunk, and the L window (kept by the last wave, read by every wave from its first
        // anti-diagonals on) must be in place before any of them starts: one rendezvous per chunk.  The last wave
        // waits for everybody's progress word, refills the window and says so; the others wait for that word.
        int wfill = 0;
        if constexpr (MULTI) {
            if (cw == NW - 1) {
                for (;;) {
                    const int v = lane < NW - 1 ? __hip_atomic_load(&prog[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : pbase;
                    if (__builtin_amdgcn_ballot_w64(!reached(v, pbase)) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                NPORE_OBSERVE_FENCE();
            }
        }
        // reference-L window: positions [0, wfill) are resident (modulo rwin)
        while (r + WIN_SLACK >= wfill) {
            if (cw == NW - 1 && lane < WIN_STEP) {
                const int j = wfill + lane;
                win[j & env.wmask] = (j <= d.dcols) ? refl_g[j] : make_uint2(0u, 0u);
            }
            wfill += WIN_STEP;
        }
        if constexpr (MULTI) {
            if (cw == NW - 1) {
                NPORE_PUBLISH_FENCE();
                if (lane == 0) __hip_atomic_store(&slotbox[SLOT_RING + 1], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                for (;;) {
                    const int g = __hip_atomic_load(&slotbox[SLOT_RING + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (reached(uni(g), gen)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                NPORE_OBSERVE_FENCE();
            }
        }
        if (gen > 1 && hist_lane) {
#pragma unroll
            for (int q = 0; q < NSR; q++) hist[q * hw + tcol] = hist_none();
        }
        // lane table of history offsets (DevEnv::tab_e): rows before the chunk, band not moved yet
        env.tab_e = __umul24(min((uint32_t)0 - env.t_n, (uint32_t)0 - env.t_n + (uint32_t)NSR), (uint32_t)env.hw16);

        // Per-step bookkeeping that is the same for every lane still lives in VECTOR registers: on this machine a
        // scalar instruction costs a wave about twice the issue time of a vector one (measured: 16 extra s_add per
        // step +7.7 % fill time at r=100, 16 extra v_add +3.9 %), and the scalar unit is shared by the CU's 16 waves.
        uint32_t slot_v = 0u;                 // byte offset of this anti-diagonal's row in the history ring
        uint32_t tboff_v = tcol4;             // byte offset of this lane's traceback word from the chunk's first
        float e_v = 0.0f;                     // 100 * b_local, then 100 * (b_local + 1)  (exact in fp32: b_local < 2^16)
        asm volatile("" : "+v"(slot_v), "+v"(e_v));
        uint32_t ring_bytes = (uint32_t)(NSR * env.hw16), tbstride4 = (uint32_t)p.tbstride * 4u;
        uint32_t prog_v = (uint32_t)pbase;    // this wave's progress word: anti-diagonals finished, all chunks
        // exchange records, [wave][parity][XCH_WORDS]: the record of the wave below for this anti-diagonal's parity
        // (own record = + 2 records) and the same for the other parity (what the neighbours wrote last step: the
        // wave below's at + 0, the wave above's at + 4 records); exchanged after every step
        uint32_t xown = xchg_addr + (uint32_t)((cw - 1) * 2) * (XCH_WORDS * 4u), xoth = xown + XCH_WORDS * 4u;
        uint32_t xsum = xown + xoth;      // (not const: used as an asm operand inside the step lambda)
        asm volatile("" : "+v"(prog_v), "+v"(xown), "+v"(xoth));

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

        // One anti-diagonal.  MODE 0: first row of the chunk (no neighbours), 1: the input path
        // stepped 'I' (read words move one column up, "left" is the previous lane), 2: 'D'
        // (reference words move one column down, "top" is the next lane).  The whole body is
        // instantiated per mode: the neighbour fetch then needs no selects.  (The compiler still places 14 + 5
        // register copies per step where the two bodies meet; ONE body with a short per-kind branch in front of a
        // common cell update has 5 inherent copies and measured 2.6 % slower at r = 100 -- DESIGN.md section 6.)
        // book_tag = false: ring row, lane table of history offsets and traceback row of this anti-diagonal have been
        // advanced already (a 'D' step that the assembly loop handed over behind its poll, plain_span below)
        auto step4 = [&](auto mode_tag, auto role_tag, auto fast_tag, auto book_tag) __attribute__((always_inline)) {
            constexpr int MODE = decltype(mode_tag)::value;
            constexpr bool BOOK = decltype(book_tag)::value;
            // ROLE: 0 = only wave of the chunk, 1 = first, 2 = middle, 3 = last (compile-time so that the
            // per-role code needs no joins inside the loop)
            constexpr int ROLE = decltype(role_tag)::value;
            constexpr bool IS_FIRST = (ROLE == 0 || ROLE == 1), IS_LAST = (ROLE == 0 || ROLE == 3);
            // FASTSEL: every band cell of this anti-diagonal is an ordinary one (cell.hpp step_is_plain)
            constexpr bool FASTSEL = decltype(fast_tag)::value;
            if constexpr (MODE != 0 && BOOK) {
                // ring row of this anti-diagonal, then the lane table of history offsets: its entry n is entry n-1
                // of the previous anti-diagonal's (same ring row), 16 bytes lower if the band has just moved (an
                // 'I' step); entry 0 = this anti-diagonal's own row.  Nothing here depends on the neighbour waves,
                // so it sits in front of the hand-shake.
                // (written as in-place asm: with the plain expressions the compiler forms the new values in fresh
                // registers and copies them back where the two step bodies meet)
                slot_v += (uint32_t)env.hw16;
                slot_v = (slot_v == ring_bytes) ? 0u : slot_v;
                // (one v_cndmask on a lane mask held in scalar registers; left to itself the compiler branches on exec)
                if constexpr (MODE == 1)
                    asm("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_u32 %0, -16, %0\n\t"
                        "v_cndmask_b32 %0, %0, %1, %2" : "+v"(env.tab_e) : "v"(slot_v), "s"(env.n0_lanes));
                else
                    asm("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                        "v_cndmask_b32 %0, %0, %1, %2" : "+v"(env.tab_e) : "v"(slot_v), "s"(env.n0_lanes));
                const uint32_t ts4 = tbstride4;
                asm("v_add_u32 %0, %1, %0" : "+v"(tboff_v) : "s"(ts4));
            }
            st.init_f = e_v;
            e_v += 100.0f;
            if constexpr (MULTI && MODE != 0 && !xp::NOPOLL) {
                // Per-chunk hand-shake instead of a workgroup barrier: this wave may start an anti-diagonal once
                // its two neighbour waves have finished the previous one (they own the only columns it reads),
                // i.e. once their progress words have reached this wave's own.  The neighbour released its
                // history / exchange words before its progress word; the fence behind the loop acquires them.
                // One neighbour at a time, the wave ABOVE first: the waves of a group finish an anti-diagonal in the
                // order first ... last (the SIMD serves the older wave first), so that is the one a look usually
                // fails for (87 % of the failed looks of a middle wave, measured), and a look at one word is one LDS
                // read and one compare instead of two and two.
                // (relaxed workgroup-scope atomics keep these plain LDS reads -- a volatile access would become a
                // flat system-scope load with a vmcnt(0) wait; their address comes from a vector register -- as a
                // wave-uniform value the compiler moves it there inside the loop; every lane reads the same word:
                // a vector compare + branch on vcc instead of readfirstlane + s_cmp.  No s_sleep between two looks:
                // measured -1 % at r = 64 / 100, -2 % at r = 200 against s_sleep 1; s_sleep 2 and 4 equal s_sleep 1.)
                if constexpr (!IS_LAST) {
                    for (;;) {
                        const int b = __hip_atomic_load(reinterpret_cast<lds_i32 *>(pnb_addr) + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (__builtin_amdgcn_ballot_w64(b < (int)prog_v) == 0ull) break;
                        if constexpr (xp::POLLSLEEP > 0) __builtin_amdgcn_s_sleep(xp::POLLSLEEP);
                    }
                }
                if constexpr (!IS_FIRST) {
                    for (;;) {
                        const int a = __hip_atomic_load(reinterpret_cast<lds_i32 *>(pnb_addr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (__builtin_amdgcn_ballot_w64(a < (int)prog_v) == 0ull) break;
                        if constexpr (xp::POLLSLEEP > 0) __builtin_amdgcn_s_sleep(xp::POLLSLEEP);
                    }
                }
                NPORE_OBSERVE_FENCE();
            }
            // boundary cells written by the neighbour waves at the end of the previous step: xoth = this group's
            // exchange records of the previous anti-diagonal's parity, counted from the record of the wave below
            CellIn in;
            if constexpr (MODE == 1) {
                float pm, pd;
                uint32_t pr, ps;
                if constexpr (IS_FIRST) {
                    pm = lane_prev(matv); pd = lane_prev(delv); pr = lane_prev(R2);
                    // word for row ins_l + r enters at column 0
                    if (sq_idx >= 64) {   // uniform
                        sq_idx -= 64;
                        sq_base += 64;
                        const int i = sq_base + lane;
                        seq_q = (i <= d.drows) ? seqw_g[i] : SEQW_SENTINEL;
                        asm volatile("" : "+v"(seq_q));   // wait for the reload inside this rare branch (see np_full)
                    }
                    const uint32_t incoming = (uint32_t)__builtin_amdgcn_readlane((int)seq_q, sq_idx);
                    sq_idx++;
                    ps = lane_prev_or(incoming, seqw);
                } else {
                    // last cell of the wave below (broadcast reads)
                    // (word by word into four free registers: a ds_read2 / ds_read_b128 lands in a register tuple, and
                    // the shifts below, which write in place, would leave their results there to be copied out)
                    uint32_t x0, x1, x2, x3;
                    asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\t"
                                 "ds_read_b32 %3, %4 offset:12\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(xoth) : "memory");
                    // lane 0 keeps the neighbour wave's cell, the others take their previous lane's
                    pm = lane_prev_or(__uint_as_float(x0), matv);
                    pd = lane_prev_or(__uint_as_float(x1), delv);
                    pr = lane_prev_or(x2, R2);
                    ps = lane_prev_or(x3, seqw);
                }
                in.topM = matv; in.topI = insv; in.topIrun = (int)(R1 >> 16);
                in.leftM = pm;
                in.leftD = pd;
                in.leftDrun = (int)(pr >> 16);
                in.diagM = LMv;
                in.diagMrun = (int)(LMr & 0xFFFFu);
                LMr = pr;
                TMr = R1;
                seqw = ps;
                st.ins_l++;
            } else if constexpr (MODE == 2) {
                float nm, ni;
                uint32_t nr, nx, nc0, nc1;
                if constexpr (IS_LAST) {
                    nm = lane_next(matv); ni = lane_next(insv); nr = lane_next(R1);
                    // word for col del_l + WPT-1 - r enters at column WPT-1
                    if (rq_idx >= 64) {
                        rq_idx -= 64;
                        rq_base += 64;
                        const int j = rq_base + lane;
                        ref_q = (j >= 0 && j <= d.dcols) ? refw_g[j] : make_uint4(REFW_SENTINEL, 0u, 0u, 0u);
                        asm volatile("" : "+v"(ref_q.x), "+v"(ref_q.z), "+v"(ref_q.w));   // wait inside the rare branch
                    }
                    const uint32_t inx = (uint32_t)__builtin_amdgcn_readlane((int)ref_q.x, rq_idx);
                    const uint32_t inz = (uint32_t)__builtin_amdgcn_readlane((int)ref_q.z, rq_idx);
                    const uint32_t inw = (uint32_t)__builtin_amdgcn_readlane((int)ref_q.w, rq_idx);
                    rq_idx++;
                    nx = lane_next_or(inx, refx);
                    nc0 = lane_next_or(inz, rc0);
                    nc1 = lane_next_or(inw, rc1);
                } else {
                    // first cell of the wave above: record + 4 records, words 5-10
                    uint32_t x0, x1, x2, x3, x4, x5;
                    asm volatile("ds_read_b32 %0, %6 offset:212\n\tds_read_b32 %1, %6 offset:216\n\tds_read_b32 %2, %6 offset:220\n\t"
                                 "ds_read_b32 %3, %6 offset:224\n\tds_read_b32 %4, %6 offset:228\n\tds_read_b32 %5, %6 offset:232\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5) : "v"(xoth) : "memory");
                    static_assert((4 * XCH_WORDS + 5) * 4 == 212, "offsets in the asm above");
                    nm = lane_next_or(__uint_as_float(x0), matv);
                    ni = lane_next_or(__uint_as_float(x1), insv);
                    nr = lane_next_or(x2, R1);
                    nx = lane_next_or(x3, refx);
                    nc0 = lane_next_or(x4, rc0);
                    nc1 = lane_next_or(x5, rc1);
                }
                st.del_l++;
                if (st.del_l + r + WIN_SLACK >= wfill) {   // keep the L window ahead of the band
                    if constexpr (IS_LAST) {
                        const int j = wfill + lane;
                        if (WIN_STEP == 64 || lane < WIN_STEP) win[j & env.wmask] = (j <= d.dcols) ? refl_g[j] : make_uint2(0u, 0u);
                    }
                    wfill += WIN_STEP;
                }
                in.leftM = matv; in.leftD = delv; in.leftDrun = (int)(R2 >> 16);
                in.topM = nm;
                in.topI = ni;
                in.topIrun = (int)(nr >> 16);
                in.diagM = TMv;
                in.diagMrun = (int)(TMr & 0xFFFFu);
                LMr = R2;
                TMr = nr;
                refx = nx;
                rc0 = nc0;
                rc1 = nc1;
            } else {
                in.topM = in.topI = in.leftM = in.leftD = in.diagM = 0.0f;
                in.topIrun = in.leftDrun = in.diagMrun = 0;
            }
            pad_hook(tcol4, env.n0_lanes);      // (empty in the product build: experiments.hpp)
            in.c = tcol;
            in.seqw = seqw;
            in.refx = refx;
            in.sc0 = rc0;
            in.sc1 = rc1;

            CellOut o;
            // band-edge cells (columns 0 and 2r; reference src/aln.pyx:502-507: every state = 100*(b_row+1),
            // TYP = MAT, RUN = 0).  Only three values of an edge cell are ever read (by its one in-band
            // neighbour), so only those are patched below; edge columns leave no history and the traceback
            // kernel treats them as "run 0" itself.  A middle wave holds band-interior columns only (MID).
            cell_update<FASTSEL, false, ROLE == 2>(env, st, in, o);

            LMv = in.leftM;
            TMv = in.topM;
            matv = o.matv;
            insv = o.insv;
            delv = o.delv;
            R1 = (uint32_t)o.matrun | ((uint32_t)o.insrun << 16);
            R2 = (uint32_t)o.matrun | ((uint32_t)o.delrun << 16);
            {
                const float e = e_v;
                if constexpr (IS_FIRST) {      // column 0 is lane 0 of the first wave; read as a LEFT neighbour
                    matv = (lane == 0) ? e : matv;
                    delv = (lane == 0) ? e : delv;
                    R2 = (lane == 0) ? 0u : R2;
                }
                if constexpr (IS_LAST) {       // column 2r lies in the last wave (NW = ceil((2r+1)/64)); read as a TOP neighbour
                    const bool is_edge = (tcol == 2 * r);
                    matv = is_edge ? e : matv;
                    insv = is_edge ? e : insv;
                    R1 = is_edge ? 0u : R1;
                }
            }
            // history record and traceback word of the band-interior columns (a middle wave holds no others: no
            // lane mask).  The traceback never looks at the words of the two edge columns (it stops there with
            // "run < 1", like the reference on their TYP = MAT / RUN = 0), so they are not stored.  The word goes
            // to uniform chunk base in SGPRs + per-lane byte offset, advanced by one row per step.
            if (ROLE == 2 || hist_lane) {
                *reinterpret_cast<lds_u32x4 *>(hist_c_addr + slot_v) =      // a HistCell
                    u32x4{__float_as_uint(o.matv), __float_as_uint(o.lenstart), __float_as_uint(o.shrstart),
                          (uint32_t)o.lenrun_h | ((uint32_t)o.shrrun_h << 16)};
                asm volatile("global_store_dword %0, %1, %2" : : "v"(tboff_v), "v"(o.tb), "s"(tb_g) : "memory");
                if constexpr (xp::DBGMAT) asm volatile("global_store_dword %0, %1, %2" : : "v"(tboff_v), "v"(__float_as_uint(o.matv)), "s"(dbg_g) : "memory");
            }
            if constexpr (MULTI) {
                // boundary cells for the neighbour waves: the last lane's cell for the wave above (it reads
                // it as a LEFT neighbour), the first lane's for the wave below (TOP neighbour + reference words);
                // then the progress word, after this step's LDS writes (workgroup release: LDS only, it does not
                // wait for the traceback stores above, which must stay in flight)
                lds_u32 *xout = reinterpret_cast<lds_u32 *>(xown) + 2 * XCH_WORDS;
                if constexpr (!IS_LAST) {
                    if (lane == 63) {
                        xout[0] = __float_as_uint(matv);
                        xout[1] = __float_as_uint(delv);
                        xout[2] = R2;
                        xout[3] = seqw;
                    }
                }
                asm("v_add_u32 %0, 1, %0" : "+v"(prog_v));
                if (lane == 0) {
                    if constexpr (!IS_FIRST) {
                        xout[5] = __float_as_uint(matv);
                        xout[6] = __float_as_uint(insv);
                        xout[7] = R1;
                        xout[8] = refx;
                        xout[9] = rc0;
                        xout[10] = rc1;
                    }
                    NPORE_PUBLISH_FENCE();
                    __hip_atomic_store(&prog[cw], (int)prog_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                // the other parity's records next
                const uint32_t xs = xsum;      // (an asm operand alone does not make the lambda capture it)
                asm("v_sub_u32 %0, %2, %0\n\tv_sub_u32 %1, %2, %1" : "+v"(xown), "+v"(xoth) : "s"(xs));
            }
        };

        auto step = [&](auto mode_tag, auto role_tag, auto fast_tag) __attribute__((always_inline)) {
            step4(mode_tag, role_tag, fast_tag, std::true_type{});
        };

        // anti-diagonals [b0, b1) of one step window
        auto span = [&](int b0, int b1, auto role_tag, auto fast_tag) __attribute__((always_inline)) {
            if constexpr (NWT == 1) {
                // A match of the input path is a 'D' step followed by an 'I' step: the two as ONE straight-line body
                // (the register shuffle where step bodies meet then comes once per two steps).  Only where a chunk
                // is one wave: measured -2.6 % fill at r = 30, +0.5 % at r = 100, +-0 at r = 64 / 200.
                for (int bl = b0; bl < b1;) {
                    const unsigned long long m = stepmask >> (bl & 63);
                    if ((m & 3ull) == 2ull && bl + 1 < b1) {
                        step(std::integral_constant<int, 2>{}, role_tag, fast_tag);
                        step(std::integral_constant<int, 1>{}, role_tag, fast_tag);
                        bl += 2;
                    } else if (m & 1ull) {
                        step(std::integral_constant<int, 1>{}, role_tag, fast_tag);
                        bl++;
                    } else {
                        step(std::integral_constant<int, 2>{}, role_tag, fast_tag);
                        bl++;
                    }
                }
            } else {
                for (int bl = b0; bl < b1; bl++) {
                    if ((stepmask >> (bl & 63)) & 1ull) step(std::integral_constant<int, 1>{}, role_tag, fast_tag);
                    else step(std::integral_constant<int, 2>{}, role_tag, fast_tag);
                }
            }
        };
        // The plain anti-diagonals [b0, b1) of one step window through the hand-scheduled loop (fill_step_asm.inc,
        // generated by gen_fill_asm.py: the same step as `step` with FASTSEL, every loop-carried value in ONE register
        // for both step kinds).  The text stops in front of a step that needs a rare path (a column descriptor
        // with DSC_RARE, a word queue or the L window about to run out); that one step goes through the C++ body.
        auto plain_span = [&](int b0, int b1, auto role_tag) __attribute__((always_inline)) {
            constexpr int ROLE = decltype(role_tag)::value;
            int a_bl = b0, a_sdel = uni(st.del_l);      // (uni: the compiler does not always see that these are wave-uniform)
            const int a_b1 = b1;
            const uint32_t a_hw16 = (uint32_t)env.hw16, a_wmask = (uint32_t)env.wmask, a_clampv = (uint32_t)env.clampv,
                           a_clamp1 = (uint32_t)env.clampv + 1u, a_npdim = (uint32_t)env.np_dim;
            const float a_istart = st.indel_start, a_iext = st.indel_extend;
            const uint32_t a_winaddr = (uint32_t)(reinterpret_cast<const char *>(win) - reinterpret_cast<const char *>(lds));
            const unsigned long long a_mhist = __builtin_amdgcn_ballot_w64(hist_lane), a_ml0 = 1ull, a_ml63 = 1ull << 63,
                                     a_medge = __builtin_amdgcn_ballot_w64(tcol == 2 * r), a_me = a_ml0 | a_medge;
            uint32_t a_one = 1u, a_lanej = (uint32_t)(tcol - r);
            uint32_t a_progaddr = (uint32_t)(reinterpret_cast<const char *>(prog + cw) - reinterpret_cast<const char *>(lds));
            float a_inf = huge_f(), a_c100 = INF_F;          // (a literal and VCC do not fit one v_cndmask: constants in registers)
            // run registers carry the type tag of their traceback word (layout.hpp tb_word): "run + 1" / "run 1" of INS and DEL, "run 0" of SHR
            uint32_t a_oneI = tb_word(T_INS, 1u), a_oneD = tb_word(T_DEL, 1u), a_tagS = tb_word(T_SHR, 0u);
            uint32_t a_livebc = hist_lane ? (DSC_N4 | DSC_HAS2 | DSC_RARE) : 0u;      // the descriptor's summary bits, band-interior columns only
            uint32_t a_laneid = (uint32_t)lane;
            const int a_drows = d.drows, a_dcols = d.dcols;
            asm volatile("" : "+v"(a_one), "+v"(a_lanej), "+v"(a_progaddr), "+v"(a_inf), "+v"(a_c100), "+v"(a_laneid), "+v"(a_oneI), "+v"(a_oneD), "+v"(a_tagS), "+v"(a_livebc));
            (void)a_livebc;
            (void)a_laneid; (void)a_drows; (void)a_dcols;
            (void)a_mhist; (void)a_ml0; (void)a_ml63; (void)a_medge; (void)a_me; (void)a_progaddr;
            for (;;) {
                int a_status = 1, a_sx, a_bend;
                // A column descriptor that needs the rare path (DSC_RARE) stays in the wave's lanes for up to 64 'D' steps, and
                // every step of that stretch is a compiled one: the text is not visited only to leave it again at its
                // entry test (its wait for everything in flight, the operands' way into and out of the statement)
                const bool rare_here = __builtin_amdgcn_ballot_w64((rc0 & DSC_RARE) != 0u && (ROLE != 1 || hist_lane)) != 0ull;
                if (!rare_here) {
                unsigned long long a_mask = stepmask, a_nmask = nextmask;
                int a_kbase = 64 * ((a_bl >> 6) + 2) - 1;           // first step byte of the window after the next one, for lane 0
                unsigned long long a_sa, a_sb, a_sc;
                int a_wfill = uni(wfill);
                int a_dlim = a_wfill - r - WIN_SLACK - 1;       // (the 'D' step that makes the L window refill)
                const int bl_in = a_bl, sdel_in = a_sdel;
                int a_sq = uni(sq_idx), a_rq = uni(rq_idx), a_sqb = uni(sq_base), a_rqb = uni(rq_base);
                (void)a_dlim; (void)a_sx; (void)a_bend; (void)a_sq; (void)a_rq; (void)a_sqb; (void)a_rqb; (void)a_wfill;
                if constexpr (ROLE == 0)
                    asm volatile(NPORE_FILL_ASM_TEXT_0 : NPORE_FILL_ASM_OUTS_0 : NPORE_FILL_ASM_INS_0 : NPORE_FILL_ASM_CLOBBERS);
                else if constexpr (ROLE == 1)
                    asm volatile(NPORE_FILL_ASM_TEXT_1 : NPORE_FILL_ASM_OUTS_1 : NPORE_FILL_ASM_INS_1 : NPORE_FILL_ASM_CLOBBERS);
                else if constexpr (ROLE == 2)
                    asm volatile(NPORE_FILL_ASM_TEXT_2 : NPORE_FILL_ASM_OUTS_2 : NPORE_FILL_ASM_INS_2 : NPORE_FILL_ASM_CLOBBERS);
                else
                    asm volatile(NPORE_FILL_ASM_TEXT_3 : NPORE_FILL_ASM_OUTS_3 : NPORE_FILL_ASM_INS_3 : NPORE_FILL_ASM_CLOBBERS);
                stepmask = a_mask; nextmask = a_nmask;
                sq_idx = a_sq; sq_base = a_sqb;
                rq_idx = a_rq; rq_base = a_rqb;
                if constexpr (ROLE == 0 || ROLE == 3) wfill = a_wfill;
                // the scalar bookkeeping the text does not carry: local row / column of the input path
                // 'I' steps = steps taken - 'D' steps taken (a 'D' step stopped behind its poll has counted itself already)
                st.ins_l += (a_bl - bl_in) - (a_sdel - sdel_in - (a_status == 2 ? 1 : 0));
                st.del_l = a_sdel;
                }
                if (!a_status) break;
                if (a_status == 2) {       // a 'D' step, stopped behind its poll: bookkeeping and column count already advanced
                    st.del_l = a_sdel - 1;
                    step4(std::integral_constant<int, 2>{}, role_tag, std::true_type{}, std::false_type{});
                } else if ((stepmask >> (a_bl & 63)) & 1ull) step(std::integral_constant<int, 1>{}, role_tag, std::true_type{});
                else step(std::integral_constant<int, 2>{}, role_tag, std::true_type{});
                a_bl++;
                a_sdel = uni(st.del_l);
                if ((a_bl & 63) == 0) {          // the handed-over step was the last of its window
                    stepmask = nextmask;
                    nextmask = step_window((a_bl >> 6) + 1);
                }
                if (a_bl >= b1) break;
            }
        };
        auto run = [&](auto role_tag) __attribute__((always_inline)) {
            step(std::integral_constant<int, 0>{}, role_tag, std::false_type{});
            // anti-diagonals [plain_lo, plain_hi) are plain (cell.hpp step_is_plain: both conditions are monotone along the
            // input path): ONE run of the assembly loop; the ~2r anti-diagonals either side of it, where the band touches
            // the chunk rectangle's border, go through the general cell update window by window
            constexpr bool NOASM_HERE = xp::NOASM || ((xp::NOASM_ROLES >> decltype(role_tag)::value) & 1);
            const int plo = NOASM_HERE ? d.nrows : max(d.plain_lo, 1), phi = min(d.plain_hi, d.nrows);
            auto rotate = [&](int w0) {
                stepmask = nextmask;
                nextmask = step_window((w0 >> 6) + 2);
            };
            int w0 = 0;
            while (w0 < d.nrows) {
                const int b0 = w0 ? w0 : 1, b1 = min(w0 + 64, d.nrows);
                if (plo < phi && plo < b1 && plo >= b0) {
                    if (b0 < plo) span(b0, plo, role_tag, std::false_type{});
                    plain_span(plo, phi, role_tag);                 // (rotates the step windows it crosses itself)
                    const int wend = min(((phi >> 6) << 6) + 64, d.nrows);
                    if ((phi & 63) != 0 || phi == d.nrows) {        // the rest of the window the plain range ends in
                        if (phi < wend) span(phi, wend, role_tag, std::false_type{});
                        rotate(wend - 64);
                        w0 = wend;
                    } else {
                        w0 = phi;                                   // ended on a window boundary: windows already rotated
                    }
                    continue;
                }
                if (NOASM_HERE && b0 >= d.plain_lo && b1 <= d.plain_hi) span(b0, b1, role_tag, std::true_type{});   // (A/B: compiled plain steps)
                else span(b0, b1, role_tag, std::false_type{});
                rotate(w0);
                w0 += 64;
            }
        };
        // the wave's role within its chunk decides where annotation words and boundary cells come from
        if constexpr (NWT == 1) run(std::integral_constant<int, 0>{});
        else if (cw == 0) run(std::integral_constant<int, 1>{});
        else if (cw == NW - 1) run(std::integral_constant<int, 3>{});
        else run(std::integral_constant<int, 2>{});
        pbase += d.nrows;
    }
}

// ---------------------------------------------------------------------------
struct TParams {
    const ChunkDesc *descs;
    const int32_t *n_chunks;
    const uint32_t *tb;
    const int32_t *inss;       // per read: inss[b] for every anti-diagonal of its input path
    uint32_t *chunk_runs;      // per-chunk slots (same offsets as the op slots): typ | length << 3 per run, last run first
    int32_t *chunk_nruns;      // runs recorded
    int32_t *chunk_len;        // ops they stand for
    int32_t *chunk_status;
    int r;
    int tbstride;
};

// One wavefront per chunk (reference src/aln.pyx:670-742).  A hop of the traceback needs the word of one cell and the
// band position of its anti-diagonal (inss[b]); the next cell is only known once that word has arrived, so a chunk's
// traceback is a chain of dependent HBM round trips: the row the path lands on is requested as soon as the next cell is
// known, and the anti-diagonal BELOW it comes with it (an indel of one base behind a diagonal run lands there and needs
// no round trip of its own).  With thousands of chunks in flight the round trips of different chunks overlap and the
// SCALAR instruction count per hop decides (everything here is wave-uniform: a SIMD issues about one scalar instruction
// per four cycles whatever the number of waves it holds); below that the chunk's chain of round trips does.  The kernel
// records RUNS (type, length), 64 at a time from a register buffer -- no store per hop whose completion the next hop's
// counter wait would expose; gather_kernel / standardize_kernel work from the runs.  (Rounds 1 - 4 kept a second
// implementation, windows of the band strip in registers; the row kernel has been the faster one at every batch size
// since it holds the row below, and the duplicate was removed in round 5: LABNOTES.)
// Of a row the wave holds ONE group of 64 columns, one word per lane (lane l = column 64 g + l), so the cell's word is
// one v_readlane.  The group requested with a row is the one the path is in now: its band column changes only where
// the new alignment leaves the input one, so the next cell is nearly always in the same group; where it is not, the row
// is requested again with the right group (one more round trip, once per crossing of a multiple of 64).  The bytes per
// hop are those of r <= 31 whatever the band.
// Hops run through a FAST loop while nothing special happens -- the word is a valid run that fits the slot, the cell is
// inside the band, and the next cell lies strictly inside the chunk rectangle (row0 + col0 == brk, so it is then inside
// the chunk's anti-diagonals too: they only decrease) -- ~30 scalar instructions; the first hop that fails one of
// those tests is handed, untouched, to the general loop below, which decides in the reference's order
// (src/aln.pyx:680-716) what it was.  Every such hop is one of the chunk's last few.
__global__ __launch_bounds__(64) void traceback_rows_kernel(TParams p)
{
    const int k = blockIdx.x;
    if (k >= *p.n_chunks) return;
    const int lane = threadIdx.x;
    const ChunkDesc d = p.descs[k];
    const char *tb = reinterpret_cast<const char *>(p.tb + d.tb_off);
    uint32_t *runs = p.chunk_runs + d.out_off;
    const int W = 2 * p.r + 1, stride = p.tbstride;
    int a_row = d.row0 + d.drows, a_col = d.col0 + d.dcols;
    int pos = d.out_cap;   // ops still available in the chunk's slot
    int status = 0;
    int nruns = 0;
    uint32_t rbuf = 0u;    // lane l: run number (nruns & ~63) + l

    const char *inss = reinterpret_cast<const char *>(p.inss + d.inss_off + d.brk);
    const uint32_t stride4 = (uint32_t)stride * 4u;
    uint32_t row = 0u;
    int row_ins = 0;
    int grp = p.r >> 6;    // the group held (the input path runs along band column r)
    auto load_row = [&](int bl, int g) {            // a chunk's block is < 2^32 bytes (60 000 rows x 1 024 words)
        row_ins = *reinterpret_cast<const int32_t *>(inss + (uint32_t)bl * 4u);
        const uint32_t o = (uint32_t)bl * stride4;
        row = *reinterpret_cast<const uint32_t *>(tb + (o + min((uint32_t)(g * 256 + lane * 4), stride4 - 4u)));   // kept inside the row
    };
    auto word = [&](int col) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)row, col & 63); };      // col is wave-uniform
    auto in_chunk = [&](int ar, int ac) {
        const int bl = ar + ac - d.brk;
        return ar >= d.row0 && ac >= d.col0 && bl >= 0 && bl < d.nrows;
    };

    if ((a_row > d.row0 || a_col > d.col0) && in_chunk(a_row, a_col)) {
        load_row(a_row + a_col - d.brk, grp);
        // ---- fast hops: the row of (a_row, a_col) is loaded and the cell is inside the chunk.  Written out (31 scalar
        // instructions per hop that requests rows, 29 per hop into the row already held; the compiler's form of the same loop has 55, a third of them moves and masks of its
        // control flow): every exit leaves a_row / a_col / pos / nruns / grp / row / row_ins / rbuf as they were
        // before the hop that could not be taken, with no load in flight.  Inside, the read position is kept relative to
        // the chunk's first anti-diagonal (arel = a_row - brk: the next anti-diagonal is arel + a_col), the run count as
        // n64 + nl (nl = the lane of rbuf the next run goes to), and the columns of the band the held group covers as
        // [lo, lo + span): one unsigned compare tells "inside the band and inside the group" from everything else.
        // gfx950 wait states kept by the order of the text: the compare that writes vcc and the select that reads it
        // are three scalar instructions apart (two needed); x (written by v_readlane) is read by a vector instruction
        // ~20 instructions later; lane selects are written by scalar instructions (no wait needed).
        {
            a_col = uni(a_col); pos = uni(pos); grp = uni(grp);
            int arel = uni(a_row) - d.brk, nl = uni(nruns) & 63, n64 = uni(nruns) & ~63;
            int lo = max(1, grp * 64), span = min(W - 1, grp * 64 + 64) - lo;
            uint32_t vg = min((uint32_t)(grp * 256 + lane * 4), stride4 - 4u);       // this lane's byte in a row
            const int lane4 = lane * 4;
            const uint32_t vs4 = stride4;
            int t0, t1, bc, run, x;
            uint32_t va, vt;
            // With every row the one BELOW it (one anti-diagonal earlier) is requested as well: a diagonal run is mostly
            // followed by an indel of one base, whose cell lies there -- that hop then needs no round trip (`below`).
            uint32_t row2 = 0u;
            int ins2 = 0, v2 = 0;       // v2 = 1: row2 / ins2 hold the group's words / inss of the anti-diagonal below the current one
            asm volatile(
                "hop_%=:\n\t"
                "s_waitcnt vmcnt(0)\n\t"
                "hop_nw_%=:\n\t"
                "v_readfirstlane_b32 %[t0], %[ins]\n\t"
                "s_sub_i32 %[bc], %[t0], %[A]\n\t"
                "s_add_i32 %[bc], %[bc], %[R]\n\t"                 // inss[b] - a_row + r
                "s_sub_i32 %[t0], %[bc], %[LO]\n\t"
                "s_cmp_ge_u32 %[t0], %[SPAN]\n\t"
                "s_cbranch_scc1 other_%=\n\t"                       // not a band column of the held group
                "v_readlane_b32 %[x], %[row], %[bc]\n\t"
                "s_and_b32 %[run], %[x], 0x1fffffff\n\t"
                "s_add_i32 %[t0], %[run], -1\n\t"
                "s_cmp_ge_u32 %[t0], %[P]\n\t"
                "s_cbranch_scc1 out_%=\n\t"                         // run < 1 or run > pos
                "s_cmp_ge_u32 %[x], 0xa0000000\n\t"
                "s_cbranch_scc1 out_%=\n\t"                         // no such state
                "s_cmp_lt_u32 %[x], 0x60000000\n\t"                // MAT, INS, LEN: the read position moves
                "s_cselect_b32 %[t0], %[run], 0\n\t"
                "s_add_i32 %[t1], %[x], 0xe0000000\n\t"
                "s_cmp_ge_u32 %[t1], 0x40000000\n\t"               // not INS / LEN: the reference position moves
                "s_cselect_b32 %[t1], %[run], 0\n\t"
                "s_sub_i32 %[A], %[A], %[t0]\n\t"
                "s_sub_i32 %[C], %[C], %[t1]\n\t"
                "s_cmp_le_i32 %[A], %[ROW0]\n\t"
                "s_cbranch_scc1 undo_%=\n\t"                        // the chunk's first row / column: the general loop
                "s_cmp_le_i32 %[C], %[COL0]\n\t"
                "s_cbranch_scc1 undo_%=\n\t"
                "s_add_i32 %[t1], %[t0], %[t1]\n\t"                // anti-diagonals moved (>= 1)
                "v_alignbit_b32 %[vt], %[x], %[x], 29\n\t"         // typ | run << 3
                "v_cmp_eq_u32 vcc, %[NL], %[lane]\n\t"
                "s_sub_i32 %[P], %[P], %[run]\n\t"
                "s_add_i32 %[NL], %[NL], 1\n\t"
                "s_and_b32 %[NL], %[NL], 63\n\t"
                "v_cndmask_b32 %[rbuf], %[rbuf], %[vt], vcc\n\t"
                "s_cbranch_scc0 flush_%=\n\t"
                "recorded_%=:\n\t"
                "s_cmp_eq_u32 %[t1], %[V2]\n\t"
                "s_cbranch_scc1 below_%=\n\t"
                "s_add_i32 %[t0], %[A], %[C]\n\t"                  // the anti-diagonal landed on (>= 2: the one below exists)
                "v_lshlrev_b32 %[va], 2, %[t0]\n\t"
                "global_load_dword %[ins], %[va], %[IB]\n\t"
                "global_load_dword %[ins2], %[va], %[IB] offset:-4\n\t"
                "v_mad_u32_u24 %[va], %[t0], %[vs4], %[vg]\n\t"
                "global_load_dword %[row], %[va], %[TB]\n\t"
                "v_sub_u32 %[va], %[va], %[vs4]\n\t"
                "global_load_dword %[row2], %[va], %[TB]\n\t"
                "s_mov_b32 %[V2], 1\n\t"
                "s_branch hop_%=\n\t"
                "below_%=:\n\t"                                      // one anti-diagonal down, held: no request
                "v_mov_b32 %[row], %[row2]\n\t"
                "v_mov_b32 %[ins], %[ins2]\n\t"
                "s_mov_b32 %[V2], 0\n\t"
                "s_branch hop_nw_%=\n\t"
                "flush_%=:\n\t"
                "s_lshl_b32 %[t0], %[N64], 2\n\t"                  // 64 runs recorded: store them
                "s_add_i32 %[N64], %[N64], 64\n\t"
                "v_add_u32 %[va], %[t0], %[lane4]\n\t"
                "global_store_dword %[va], %[rbuf], %[RB]\n\t"
                "s_branch recorded_%=\n\t"
                "other_%=:\n\t"
                "s_add_i32 %[t0], %[bc], -1\n\t"
                "s_cmp_ge_u32 %[t0], %[WM2]\n\t"
                "s_cbranch_scc1 out_%=\n\t"                         // band edge or outside the band
                "s_lshr_b32 %[G], %[bc], 6\n\t"                    // the path has crossed into another group: same row again
                "s_lshl_b32 %[t0], %[G], 6\n\t"
                "s_max_i32 %[LO], %[t0], 1\n\t"
                "s_add_i32 %[t0], %[t0], 64\n\t"
                "s_add_i32 %[t1], %[WM2], 1\n\t"
                "s_min_i32 %[t0], %[t0], %[t1]\n\t"
                "s_sub_i32 %[SPAN], %[t0], %[LO]\n\t"
                "s_lshl_b32 %[t0], %[G], 8\n\t"
                "v_add_u32 %[vg], %[t0], %[lane4]\n\t"
                "v_min_u32 %[vg], %[S4M4], %[vg]\n\t"
                "s_add_i32 %[t0], %[A], %[C]\n\t"
                "s_mov_b32 %[V2], 0\n\t"
                "v_mad_u32_u24 %[va], %[t0], %[vs4], %[vg]\n\t"
                "global_load_dword %[row], %[va], %[TB]\n\t"
                "s_branch hop_%=\n\t"
                "undo_%=:\n\t"
                "s_add_i32 %[A], %[A], %[t0]\n\t"
                "s_add_i32 %[C], %[C], %[t1]\n\t"
                "out_%=:\n\t"
                : [A] "+s"(arel), [C] "+s"(a_col), [P] "+s"(pos), [NL] "+s"(nl), [N64] "+s"(n64), [G] "+s"(grp), [LO] "+s"(lo),
                  [SPAN] "+s"(span), [V2] "+s"(v2), [row] "+v"(row), [ins] "+v"(row_ins), [row2] "+v"(row2), [ins2] "+v"(ins2),
                  [rbuf] "+v"(rbuf), [vg] "+v"(vg), [t0] "=&s"(t0), [t1] "=&s"(t1), [bc] "=&s"(bc), [run] "=&s"(run), [x] "=&s"(x),
                  [va] "=&v"(va), [vt] "=&v"(vt)
                : [R] "s"(p.r - d.brk), [WM2] "s"(W - 2), [ROW0] "s"(d.row0 - d.brk), [COL0] "s"(d.col0), [S4M4] "s"(stride4 - 4u),
                  [TB] "s"(tb), [IB] "s"(inss), [RB] "s"(runs), [lane] "v"(lane), [lane4] "v"(lane4), [vs4] "v"(vs4)
                : "vcc", "scc", "memory");
            a_row = arel + d.brk;
            nruns = n64 + nl;
        }
    }
    // two combined tests per ordinary hop; what stopped the loop is sorted out in the reference's order
    // (src/aln.pyx:680-716) where it stops, so the status bits are those of one test per condition
    for (;;) {
        a_row = uni(a_row); a_col = uni(a_col); pos = uni(pos); nruns = uni(nruns);
        const bool live = (a_row > d.row0) | (a_col > d.col0);
        if (!(live & in_chunk(a_row, a_col))) {
            if (live) status |= 16;
            break;
        }
        const int bc = uni(row_ins) - a_row + p.r;     // inss[b] - a_row + r, src/aln.pyx:322-326
        if ((bc <= 0) | (bc >= W - 1)) {
            status |= (bc < 0 || bc >= W) ? 16 : 4;    // band edge: TYP = MAT, RUN = 0 (src/aln.pyx:502-507) -> "run < 1"
            break;
        }
        if ((bc >> 6) != uni(grp)) {
            grp = bc >> 6;
            load_row(a_row + a_col - d.brk, grp);
            continue;
        }
        const uint32_t x = word(bc);
        const int typ = tb_typ(x), run = tb_run(x);     // src/aln.pyx:684-685
        if ((run < 1) | (run > pos) | (typ > T_SHR)) {
            status |= (run < 1) ? 4 : (run > pos) ? 16 : 8;
            break;
        }
        const bool ins = (typ == T_LEN) | (typ == T_INS), del = (typ == T_SHR) | (typ == T_DEL);
        int emit = run;
        if (typ == T_MAT) {
            const int lim = min(a_row - d.row0, a_col - d.col0);
            emit = run < lim ? run : lim;                          // diagonal steps that stay in the chunk
        }
        const int n_row = a_row - (del ? 0 : emit), n_col = a_col - (ins ? 0 : emit);
        // request the next row now
        if (emit == run && (n_row > d.row0 || n_col > d.col0) && in_chunk(n_row, n_col)) load_row(n_row + n_col - d.brk, grp);
        if (emit > 0) {
            rbuf = (lane == (nruns & 63)) ? ((uint32_t)typ | ((uint32_t)emit << 3)) : rbuf;
            nruns++;
            if ((nruns & 63) == 0) runs[nruns - 64 + lane] = rbuf;
        }
        pos -= emit;
        if (emit < run) { status |= 16; break; }
        a_row = n_row;
        a_col = n_col;
    }
    if (lane == 0) {
        p.chunk_len[k] = d.out_cap - pos;
        p.chunk_status[k] = status;
        p.chunk_nruns[k] = nruns;
    }
    if (lane < (nruns & 63)) runs[(nruns & ~63) + lane] = rbuf;
}

// ---------------------------------------------------------------------------
constexpr int GATHER_TILE = 256 * 33;      // output positions per LDS tile of gather_kernel (33 per thread: odd, for the LDS banks)
struct GParams {
    const ChunkDesc *descs;
    const int32_t *read_first_chunk;   // [n_reads+1]
    const uint32_t *chunk_runs;
    const int32_t *chunk_nruns;
    const int32_t *chunk_len, *chunk_status;
    const int32_t *read_status_in;     // prep status per read (bad input)
    const int32_t *counters;
    const uint8_t *seqs, *refs;        // the batch's read / reference bases ('=' vs 'X')
    uint8_t *out;
    const int64_t *out_off;            // [n_reads+1] in the caller's buffer
    int64_t *out_len;
    int32_t *status;
    int64_t read_base;                 // index of this group's first read in the caller's arrays
    int64_t n_reads;                   // reads of this group
    int64_t *chunk_woff;               // [chunks] position of the chunk's ops in its read's string
    int slice_cap;                     // gather_kernel: LDS bytes per staged base slice (0: bases are read from global memory)
};

// wave per read: status, output length, and the position of every chunk's ops in the read's string
__global__ __launch_bounds__(256) void gather_scan_kernel(GParams p)
{
    const int64_t rd = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (rd >= p.n_reads) return;
    const int64_t grd = p.read_base + rd;
    const int c0 = p.read_first_chunk[rd], c1 = p.read_first_chunk[rd + 1];
    int st = p.read_status_in[rd];
    int64_t total = 0;
    for (int cb = c0; cb < c1; cb += 64) {
        const int c = cb + lane;
        const int len = (c < c1) ? p.chunk_len[c] : 0;
        int s_ = (c < c1) ? p.chunk_status[c] : 0;
        int64_t inc = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s_ |= __shfl_xor(s_, o);
        if (c < c1) p.chunk_woff[c] = total + inc - len;
        total += __shfl(inc, 63);
        st |= s_;
    }
    const int64_t cap = p.out_off[grd + 1] - p.out_off[grd];
    if (st & 32) total = -1;
    else if (total > cap) { st |= 64; total = -1; }
    if (lane == 0) { p.out_len[grd] = total; p.status[grd] = st; }
}

// TILE_IN_LDS = false: no dynamic LDS -- every thread stores its ops straight to the output (uncoalesced) and reads
// the bases from global memory: slower on its own, but a workgroup then needs 3 KB of LDS only and runs beside a
// fill-kernel workgroup (the variant for a group whose successor's fill is already on the GPU).
template <bool TILE_IN_LDS>
__global__ __launch_bounds__(256) void gather_kernel(GParams p)
{
    const int c = blockIdx.x;
    if (c >= p.counters[0]) return;
    const ChunkDesc d = p.descs[c];
    const int64_t grd = p.read_base + d.read_id;
    if (p.out_len[grd] < 0) return;
    uint8_t *dst = p.out + p.out_off[grd] + p.chunk_woff[c];
    // Expand the chunk's runs (recorded last run first) into ops.  Run j ends where the ops of the runs before
    // it (in recording order) begin, counted from the chunk's end, and pairs the bases below the cell reached
    // after those runs: three prefix sums in recording order -- ops, read bases, reference bases.  The output is
    // produced in tiles of GATHER_TILE positions: every thread takes an equal, contiguous share of the tile (runs
    // differ wildly in length), finds its first run through per-thread-segment sums and writes its ops into an
    // LDS tile, which the workgroup then stores with consecutive lanes on consecutive bytes; the chunk's two base
    // slices are staged in LDS the same way when they fit (a share per thread means addresses ~33 bytes apart
    // across the lanes: uncoalesced when it goes to memory directly).
    extern __shared__ __attribute__((aligned(16))) uint8_t gl[];
    __shared__ int s_ops[257], s_rows[257], s_cols[257];
    const int T = (int)blockDim.x, t = (int)threadIdx.x;
    const int len = p.chunk_len[c], nr = p.chunk_nruns[c];
    const uint32_t *runs = p.chunk_runs + d.out_off;
    uint8_t *l_ops = gl;
    // bases below the chunk's end cell: seq[row0, row_end), ref[col0, col_end)
    const bool staged = TILE_IN_LDS && p.slice_cap > 0 && d.drows <= p.slice_cap && d.dcols <= p.slice_cap;
    const uint8_t *seq = p.seqs + d.seq_off + d.row0, *ref = p.refs + d.ref_off + d.col0;   // local row / column 0
    if (staged) {
        uint8_t *l_seq = gl + GATHER_TILE, *l_ref = l_seq + p.slice_cap;
        for (int k = t; k < d.drows; k += T) l_seq[k] = seq[k];
        for (int k = t; k < d.dcols; k += T) l_ref[k] = ref[k];
    }
    const int seg = (nr + T - 1) / T;
    const int e0 = min(nr, t * seg), e1 = min(nr, e0 + seg);
    int so = 0, sr = 0, sc = 0;
    for (int e = e0; e < e1; e++) {
        const uint32_t x = runs[e];
        const int typ = (int)(x & 7u), l = (int)(x >> 3);
        so += l;
        sr += (typ == T_DEL || typ == T_SHR) ? 0 : l;
        sc += (typ == T_INS || typ == T_LEN) ? 0 : l;
    }
    s_ops[t + 1] = so; s_rows[t + 1] = sr; s_cols[t + 1] = sc;
    __syncthreads();
    if (t == 0) {
        s_ops[0] = s_rows[0] = s_cols[0] = 0;
        for (int q = 1; q <= T; q++) { s_ops[q] += s_ops[q - 1]; s_rows[q] += s_rows[q - 1]; s_cols[q] += s_cols[q - 1]; }
    }
    __syncthreads();
    auto base_pair_equal = [&](int ri, int ci) -> bool {   // local read row ri, local reference column ci
        if (staged) return gl[GATHER_TILE + ri] == gl[GATHER_TILE + p.slice_cap + ci];
        return seq[ri] == ref[ci];
    };
    for (int U0 = 0; U0 < len; U0 += GATHER_TILE) {
        const int U1 = min(len, U0 + GATHER_TILE);
        // output positions u (0 = the chunk's LAST op) of this thread within the tile
        const int useg = (U1 - U0 + T - 1) / T;
        const int u0 = min(U1, U0 + t * useg), u1 = min(U1, u0 + useg);
        if (u0 < u1) {
            int lo = 0, hi = T;                     // segment sgm with s_ops[sgm] <= u0 < s_ops[sgm + 1]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_ops[mid] <= u0) lo = mid; else hi = mid; }
            int e = min(nr, lo * seg), ao = s_ops[lo], ar = s_rows[lo], ac = s_cols[lo];
            int typ = 0, l = 0;
            for (;; e++) {                          // first run that reaches beyond u0
                const uint32_t x = runs[e];
                typ = (int)(x & 7u); l = (int)(x >> 3);
                if (ao + l > u0) break;
                ao += l;
                ar += (typ == T_DEL || typ == T_SHR) ? 0 : l;
                ac += (typ == T_INS || typ == T_LEN) ? 0 : l;
            }
            for (int u = u0; u < u1; u++) {
                while (u >= ao + l) {               // next run
                    ao += l;
                    ar += (typ == T_DEL || typ == T_SHR) ? 0 : l;
                    ac += (typ == T_INS || typ == T_LEN) ? 0 : l;
                    e++;
                    const uint32_t x = runs[e];
                    typ = (int)(x & 7u); l = (int)(x >> 3);
                }
                uint8_t op;
                if (typ == T_MAT) {                 // '=' / 'X' by comparing the paired bases, src/aln.pyx:732-735
                    const int q = u - ao;
                    op = base_pair_equal(d.drows - ar - q - 1, d.dcols - ac - q - 1) ? '=' : 'X';
                } else {
                    op = (typ == T_INS || typ == T_LEN) ? 'I' : 'D';
                }
                if constexpr (TILE_IN_LDS) l_ops[u - U0] = op;
                else dst[len - 1 - u] = op;
            }
        }
        if constexpr (TILE_IN_LDS) {
            __syncthreads();
            for (int k = t; k < U1 - U0; k += T) dst[len - 1 - U0 - k] = l_ops[k];
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// realign_read's glue on the device (reference src/bam.pyx:59-78 with src/cig.pyx:13-38, 102-192): ONE WAVEFRONT PER READ
// runs the streaming standardisation of std_stream.hpp over the read's traceback runs (chunk after chunk, each chunk's runs
// last-recorded first: that is read order) and writes the collapsed CIGAR text -- digits and 'M' / 'I' / 'D' -- into the
// read's output slot, where gather_kernel would have put the op string.  The stages are sequential by nature (every indel
// run sees the list as the ones before left it), so the state machine is WAVE-UNIFORM -- scalar registers, scalar branches --
// and the 64 lanes serve the two things that are not: the runs come 64 per load (a register buffer, one `v_readlane` per
// run), and how far an indel run slides through the matches in front of it is ONE round trip of 64 compared positions
// (WaveProbe) instead of one per position.  (Rounds 4: one LANE per read -- 63 wavefronts per 4 000 reads, every lane a chain
// of dependent byte loads: 7 ms; the same code with one read per wavefront 3.5 ms: the chain of round trips per read is
// what the time is.)  No LDS, 64 registers: it runs beside the next batch's fill kernel.
// A read align() refused (out_len < 0 from gather_scan_kernel) gets an empty text, like the host path's empty string.
struct StdKParams {
    const ChunkDesc *descs;
    const int32_t *read_first_chunk;   // [n_reads+1]
    const uint32_t *chunk_runs;
    const int32_t *chunk_nruns;
    const uint8_t *refs, *seqs;        // the batch's bases, only compared for equality
    const int64_t *ref_off, *seq_off;  // [n_reads+1], this group's
    uint8_t *out;
    const int64_t *out_off;            // [n_reads+1] in the caller's buffer: 2 bytes per op + 16 always suffice
    int64_t *out_len;                  // in: ops of the alignment (or < 0); out: bytes of text
    int32_t *status;
    int64_t read_base, n_reads;
};

// (every lane of the wave calls it with the same arguments; lane 0 stores)
struct CigarTextSink {
    uint8_t *o, *end;
    bool overflow;
    __device__ __forceinline__ void operator()(uint32_t op, int32_t n)
    {
        uint32_t v = (uint32_t)n;
        int nd = 1;
        for (uint32_t t = v; t >= 10u; t /= 10u) nd++;
        if (o + nd + 1 > end) { overflow = true; return; }
        if (threadIdx.x == 0) {
            for (int k = nd - 1; k >= 0; k--) { o[k] = (uint8_t)('0' + v % 10u); v /= 10u; }
            o[nd] = (uint8_t)(op == SOP_M ? 'M' : op == SOP_I ? 'I' : 'D');
        }
        o += nd + 1;
    }
};

// std_stream.hpp's probe by the 64 lanes of a wavefront: positions p - 1 - t, t = s + lane, 64 at a time (two coalesced
// byte loads per lane and one ballot per round); every lane returns the same count
struct WaveProbe {
    __device__ __forceinline__ int32_t operator()(const uint8_t *s_, int32_t p, int32_t k, int32_t m, int32_t s_len) const
    {
        const int32_t lane = (int32_t)threadIdx.x;
        int32_t s = 0;
        for (;;) {
            const int32_t t = s + lane;
            bool ok = t < m && p - t - 1 + k < s_len;
            if (ok) ok = s_[p - t - 1] == s_[p - t - 1 + k];
            const unsigned long long differ = ~__builtin_amdgcn_ballot_w64(ok);
            const int32_t n = differ ? (int32_t)__builtin_ctzll(differ) : 64;
            s += n;
            if (n < 64) break;
        }
        return s;
    }
};

// Held to 64 vector registers: what the fill kernel's four waves leave on a SIMD (fill_kernel) -- with more it would wait
// for a fill workgroup to leave its CU, i.e. for the end of the launch it is meant to run beside.
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(32))) void standardize_kernel(StdKParams p)
{
    const int64_t rd = (int64_t)blockIdx.x;
    if (rd >= p.n_reads) return;
    const int lane = (int)threadIdx.x;
    const int64_t grd = p.read_base + rd;
    if (p.out_len[grd] < 0) { if (lane == 0) p.out_len[grd] = 0; return; }
    uint8_t *start = p.out + p.out_off[grd];
    CigarTextSink sink{start, p.out + p.out_off[grd + 1], false};
    // (32-bit lengths and positions: a read has fewer than 2^31 ops -- run_core refuses longer ones)
    StdStream<CigarTextSink, int32_t, WaveProbe> st(sink, p.refs + p.ref_off[rd], (int32_t)(p.ref_off[rd + 1] - p.ref_off[rd]),
                                                    p.seqs + p.seq_off[rd], (int32_t)(p.seq_off[rd + 1] - p.seq_off[rd]));
    const int c0 = uni(p.read_first_chunk[rd]), c1 = uni(p.read_first_chunk[rd + 1]);
    // the read's runs in read order (a chunk's last-recorded run first), 64 per load: lane l holds run `hi - l` of the
    // current chunk; the loads of the next 64 are issued when the first of these is handed out
    int c = c0;
    const uint32_t *runs = nullptr;
    int e = -1;                        // next run of the current chunk (counts down)
    uint32_t rbuf = 0u;
    int hi = -1, lo = 0;               // the buffer holds runs lo ... hi
    auto next_run = [&](uint32_t &x) -> bool {
        while (e < 0) {
            if (c >= c1) return false;
            runs = p.chunk_runs + p.descs[c].out_off;
            e = uni(p.chunk_nruns[c]) - 1;
            hi = -1;
            lo = 0;
            c++;
        }
        if (e > hi || e < lo) {
            hi = e;
            lo = e > 63 ? e - 63 : 0;
            rbuf = e - lane >= 0 ? runs[e - lane] : 0u;
        }
        x = (uint32_t)__builtin_amdgcn_readlane((int)rbuf, uni(hi - e));
        e--;
        return true;
    };
    uint32_t x = 0u;
    while (next_run(x)) {
        const int typ = (int)(x & 7u);
        st.feed(typ == T_MAT ? SOP_M : (typ == T_INS || typ == T_LEN) ? SOP_I : SOP_D, (int32_t)(x >> 3));
    }
    st.finish();
    if (lane == 0) {
        if (sink.overflow) {
            p.out_len[grd] = -1;
            p.status[grd] |= 64;          // NPORE_ST_OUT_CAP
        } else {
            p.out_len[grd] = (int64_t)(sink.o - start);
        }
    }
}

// exhaustive check of div_recip on its domain: counts violations of its contract
__global__ void divcheck_kernel(unsigned long long *bad)
{
    const int run = blockIdx.x * blockDim.x + threadIdx.x;   // 0..65535
    unsigned long long b = 0;
    for (int n = 1; n <= MAX_PERIOD; n++) {
        const int q = div_recip(run, recip16(n)), t = run / n;
        b += !(q == t || (run >= 13107 && q == t + 1));   // cell.hpp: exact where it matters
    }
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
