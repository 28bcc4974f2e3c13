// npore_api.cpp -- C ABI (include/npore_amd.h) and host orchestration.
// Compiled with hipcc for gfx950 only.  There is no CPU execution path for the
// DP here: without a gfx950 device npore_ctx_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <future>
#include <mutex>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../include/npore_amd.h"
#include "confusion.hpp"
#include "glue.hpp"
#include "hostio.hpp"
#include "kernels.hpp"
#include "prep_kernels.hpp"
#include "annot_wave.hpp"
#include "unpack_kernels.hpp"

using namespace npore;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
// No C++ exception may cross the C ABI: the entry points that allocate are function-try-blocks ending in one of these.
#define NPORE_CATCH_INT                                                                             \
    catch (const std::bad_alloc &) { return fail(NPORE_E_NOMEM, "out of host memory"); }           \
    catch (const std::exception &e) { return fail(NPORE_E_INVALID, std::string("internal: ") + e.what()); }
#define NPORE_CATCH_PTR                                                                             \
    catch (const std::bad_alloc &) { fail(NPORE_E_NOMEM, "out of host memory"); return nullptr; }  \
    catch (const std::exception &e) { fail(NPORE_E_INVALID, std::string("internal: ") + e.what()); return nullptr; }
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(NPORE_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return NPORE_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        AllocTrace tr("hipMalloc", want);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(NPORE_E_NOMEM, "hipMalloc(" + std::to_string(want) + "): " + hipGetErrorString(e));
        }
        cap = want;
        return NPORE_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    // best effort, no head-room: another buffer of the same role already has this capacity (presize_like)
    void match(const DevBuf &o)
    {
        if (o.cap <= cap) return;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        AllocTrace tr("hipMalloc like", o.cap);
        if (hipMalloc(&p, o.cap) == hipSuccess) cap = o.cap;
        else { p = nullptr; (void)hipGetLastError(); }
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct HostBuf {   // pinned staging
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return NPORE_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        AllocTrace tr("hipHostMalloc", want);
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(NPORE_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        }
        cap = want;
        return NPORE_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace

// One batch on its way through the BAM -> SAM pipeline: host staging (page-locked where it crosses PCIe) and offsets.
// The file pipeline's texts compacted on the device (unpack_kernels.hpp compact_texts_kernel): what a batch's align call needs
// to know about it
struct TextCompact {
    uint8_t *d_ctext;
    unsigned long long *d_cursor;
    int64_t cap;                 // bytes of d_ctext (all slots: it cannot overflow)
    int64_t *h_coff;             // [n_reads] page-locked
    char *h_ctext;               // page-locked
    int64_t h_bytes;             // how much of the compact buffer's front to send with the batch's last group
};

struct npore_batch_slot {
    RawBuf refs{true}, seqs{true}, cigs{true}, alns{true}, finals, sam;
    RawBuf raw{true};            // device pack: the heads of the batch's records (fixed fields ... 4-bit bases), one after the other
    std::vector<int64_t> rawo;   // ... and where each starts
    // device glue: the batch's texts compacted on the device (unpack_kernels.hpp compact_texts_kernel) -- the compact buffer
    // and its cursor there, the copied front of it and the reads' offsets here (page-locked)
    DevBuf d_ctext, d_cursor;
    RawBuf ctext_pin{true}, coff_pin{true};
    int64_t ctext_copied = 0;              // bytes of the compact buffer the batch's last group sent behind its kernels
    RawBuf olen_pin{true}, st_pin{true};   // lengths / status bits of an ASYNCHRONOUS batch land here (page-locked: a copy into
                                           // pageable memory would make the enqueueing call wait for the whole batch)
    hipEvent_t done = nullptr;             // ... behind which this event is recorded (npore_bam_realign_file)
    std::vector<std::shared_ptr<RawBuf>> keep;   // one-pass ingest: the inflated windows the batch's records lie in
    ~npore_batch_slot() { if (done) (void)hipEventDestroy(done); }
    RecFetch rf;                 // the batch's BAM records (streamed handles: inflated for the batch)
    double t_ms[6] = {0, 0, 0, 0, 0, 0};   // npore_bam_realign_file: fetch + pack, align call, standardise, format, write, (spare)
    std::vector<int64_t> ro, so, co, oo, fo, olen, flen;
    int64_t sam_len = 0;
    int64_t m = 0;               // reads of the batch
    int rc = 0;
    std::string err;
};

// Work buffers of one group of reads on its way through the device stages (grow-only, reused).  A context has
// N_SETS of them: while the fill kernel works on one group, the next group is prepared in another set and the previous
// group's traceback / gather drains from a third (run_core).
struct WorkSet {
    DevBuf rd_i32, rd_i64, steps, inss, descs, sched, hist, counters; // path + chunks
    DevBuf tiles, cwoff;                                             // CIGAR tiles; chunk positions in the output
    DevBuf seqw, refw, refl, seql;                                   // annotation
    DevBuf tb, cout_, clen, cstat, cnruns;                           // fill / traceback (cout_: uint32 runs)
    DevBuf dbg;                                                      // experiments build: MAT.VAL per cell (NPORE_DBGMAT=1)
    HostBuf h_cnt;                                                   // counters read back with the group
    // host-buffer entry points: the group's slice of the caller's inputs / outputs on the device, its offset
    // arrays rebased to the slice (page-locked copy for the upload)
    DevBuf in_refs, in_seqs, in_cigs, in_off, out, out_len, status;
    DevBuf in_raw;               // device pack (unpack_kernels.hpp): the group's record heads
    DevBuf coff;                 // compacted texts: where each read of the group begins in the batch's compact buffer
    HostBuf h_off;
    hipEvent_t evc[4] = {};      // H2D start / end, D2H start / end of a staged group
    bool staged = false;
    hipEvent_t ev[6] = {};       // prep start / end, fill start / end, traceback + gather start / end (= group done)
    bool busy = false;           // enqueued, not collected yet
    // An idle set takes the capacities of one that has just been given a group: the groups of a run are alike, so its
    // own first group then finds its buffers in place instead of allocating tens of GB in front of its kernels (with
    // three sets that was the THIRD step of a run -- 0.4 s in a timed region that had two warm-up steps)
    void presize_like(const WorkSet &o)
    {
        DevBuf WorkSet::*const all[] = {&WorkSet::rd_i32, &WorkSet::rd_i64, &WorkSet::steps, &WorkSet::inss, &WorkSet::descs, &WorkSet::sched,
                                        &WorkSet::hist, &WorkSet::counters, &WorkSet::tiles, &WorkSet::cwoff, &WorkSet::seqw, &WorkSet::refw,
                                        &WorkSet::refl, &WorkSet::seql, &WorkSet::tb, &WorkSet::cout_, &WorkSet::clen, &WorkSet::cstat,
                                        &WorkSet::cnruns, &WorkSet::in_refs, &WorkSet::in_seqs, &WorkSet::in_cigs, &WorkSet::in_off,
                                        &WorkSet::out, &WorkSet::out_len, &WorkSet::status, &WorkSet::in_raw, &WorkSet::coff};
        for (auto m : all) (this->*m).match(o.*m);
    }
    int64_t cells = 0, call_id = 0;
    DevBuf *all[28] = {&rd_i32, &rd_i64, &steps, &inss, &descs, &sched, &hist, &counters, &tiles, &cwoff,
                       &seqw, &refw, &refl, &seql, &tb, &cout_, &clen, &cstat, &cnruns,
                       &in_refs, &in_seqs, &in_cigs, &in_off, &out, &out_len, &status, &in_raw, &coff};
};

// Work sets of a context: group k + 1 is prepared while group k is in the fill kernel and group k - 1 in its traceback;
// the third set lets the host enqueue group k + 1's preparation without waiting for group k - 1's traceback to end
// (with two, that wait sits between every pair of groups; measured equal within 1 % either way on this hardware --
// what binds the pipelined r = 30 case is the preparation's own duration beside a running fill kernel, LABNOTES.md).
constexpr int N_SETS = 3;

struct npore_ctx {
    int device = 0;
    int n_cus = 256;
    int max_n = 6, max_l = 100;
    // three non-blocking streams: preparation (also every copy), fill kernels, traceback + gather; events order
    // the stages of a group, the streams let stages of neighbouring groups run side by side
    hipStream_t stream = nullptr, s_fill[2] = {nullptr, nullptr}, s_post = nullptr;
    const TextCompact *pending_compact = nullptr;      // file pipeline: the next align_batch_raw / _host call compacts its texts
    int next_fill = 0;           // the fill stream the next group's fill kernel goes to
    int fill_streams = 2;        // 1: every fill kernel on one stream (npore_ctx_set "fill_streams")
    hipEvent_t ev[8] = {};       // [4..7] H2D / D2H of the host-buffer entry point, [0] the caller's stream
    float *d_sub = nullptr, *d_np = nullptr;   // NULL in an annotation-only context (created without tables)
    WorkSet ws[N_SETS];
    int next_ws = 0;             // set the next group goes into (the oldest of them)
    WorkSet *last_ws = nullptr;  // set of the group enqueued last (npore_debug_fetch)
    int64_t call_id = 0, timing_call = -1;
    int deferred_rc = 0;         // failure found while collecting a group of an asynchronous call
    std::string deferred_err;
    double totals[8] = {};       // like timing[], summed over every group since the context was made
    // tunables
    int64_t tb_budget_mb = 0;   // 0 = auto
    int force_chunks = 0;
    int device_glue = 1;        // BAM -> SAM pipeline: realign_read's glue on the device (0: on the host, from the op strings)
    int coresident = 1;         // kernel shapes that fit beside a fill kernel for a group that overlaps another one's
    int device_pack = 1;        // BAM -> SAM pipeline with the glue on the device: align()'s inputs unpacked from the records on the device
    // device pack: the FASTA of the current run on the device (uploaded once per FASTA), the contig of every BAM reference
    DevBuf d_fasta, d_ctg;
    uint64_t d_fasta_serial = 0;
    size_t d_fasta_bytes = 0;
    int n_ctg = 0;
    bool fill_has_room = false; // the last fill launch left LDS for such kernels on its CUs
    HostBuf h_offs;             // offset arrays of a device-resident batch (npore_align_batch_device)
    // device buffers (grow-only, reused across calls)
    DevBuf in_refs, in_seqs, in_cigs, in_off;                       // raw inputs (host-buffer entry point)
    std::vector<int32_t> regions;                                    // npore_np_regions: positions, then repeat counts
    DevBuf out, out_off, out_len, status;                            // outputs (host-buffer entry point)
    // host staging of the BAM -> SAM pipeline (npore_bam_realign_batch / _file): grow-only, reused across batches and files
    double file_mark[2] = {0, 0};      // totals at the start of npore_bam_realign_file (kernels, PCIe)
    static constexpr int N_SLOTS = 6;
    npore_batch_slot *slots[N_SLOTS] = {};
    // second context of the file pipeline (its own stream and work buffers), so that the transfers, preparation and
    // traceback of one batch run beside the fill kernel of its neighbour; made on first use from the host copy of the tables
    std::vector<float> h_sub, h_np;
    npore_ctx *peer = nullptr;
    double timing[8] = {};
};

namespace {

std::atomic<int> g_live_ctx[16];   // contexts alive per device (they share its memory: run_core's budget)

// waves per chunk: the smallest count whose 64 * nw lanes cover the band (the kernel relies on band
// column 2r lying in the last wave); 0 if the band is too wide
int pick_shape(int r)
{
    const int nw = (2 * r + 1 + 63) / 64;
    return nw <= MAX_WAVES_PER_CHUNK ? nw : 0;
}

int pow2_at_least(int x)
{
    int p = 64;
    while (p < x) p <<= 1;
    return p;
}

// Geometry of a fill launch for band half-width r: LDS sizes, chunks per workgroup and how many workgroups the
// GPU holds at a time.
struct FillGeom {
    int nw = 0, hw = 0, rwin = 0, cmax = 0;
};
bool fill_geometry(int r, FillGeom &g)
{
    g.nw = pick_shape(r);
    if (!g.nw) return false;
    g.hw = 2 * r + 1 + HIST_PAD;
    // reference-L window: the band (2r+1), 96 positions of read-ahead and the 6 positions below the band that the
    // generic SHR path looks back on -- plus 16 of margin, because the first wave of a chunk may run NW - 2
    // anti-diagonals behind the last one, which refills the window
    // (a chunk of ONE wave refills for itself, 32 positions at a time with 8 of slack: kernels.hpp WIN_STEP / WIN_SLACK;
    // r <= 31 then needs 2r + 6 + 8 + 32 <= 128 entries, which leaves the CU 16 KB of LDS at r = 30 -- room for the
    // kernels of the neighbouring batches beside 16 chunks)
    g.rwin = g.nw == 1 ? pow2_at_least(2 * r + 6 + 8 + 32) : pow2_at_least(2 * r + 101 + 16);
    const size_t lds_cap = 160 * 1024 / sizeof(float);
    if (fill_lds_floats(g.nw, 1, g.hw, g.rwin) > lds_cap) return false;
    g.cmax = 1;
    while ((g.cmax + 1) * g.nw * 64 <= 1024 && fill_lds_floats(g.nw, g.cmax + 1, g.hw, g.rwin) <= lds_cap) g.cmax++;
    return true;
}
// workgroups of `chunks` chunks that are resident together: the size of a persistent fill launch
int fill_round_workgroups(const FillGeom &g, int chunks, int n_cus)
{
    const size_t lds = fill_lds_floats(g.nw, chunks, g.hw, g.rwin) * sizeof(float);
    const int wg_per_cu = std::max(1, std::min((int)((160 * 1024) / std::max<size_t>(lds, 1)), 2048 / (64 * g.nw * chunks)));
    return std::max(1, n_cus) * wg_per_cu;
}

// NW waves per chunk, `chunks` chunks per workgroup (they share the LDS score table).
// leave_room: groups of reads overlap on the device (run_core), so the next group's preparation and this one's
// gather will look for room BESIDE fill workgroups: where the fill would take (nearly) all of a CU's LDS -- r = 30:
// 16 chunks = 159.75 KB -- a workgroup takes one chunk less (measured at r = 30, 8 000 reads per batch: 154 k
// instead of 144 k reads/s; the scans and the gather need ~3.5 KB of LDS).
// (NWT = 0: the instantiation that takes its wave count from the launch -- bands of 9 ... 16 waves, one chunk per workgroup)
template <int NWT>
hipError_t launch_fill(KParams kp, int max_chunks, int force_chunks, int n_cus, hipStream_t s, bool leave_room, bool *has_room)
{
    constexpr int MAXT = 1024;
    FillGeom g;
    if (!fill_geometry(kp.r, g) || (NWT ? g.nw != NWT : g.nw <= 8)) return hipErrorInvalidValue;
    const int NW = g.nw;
    kp.hw = g.hw;
    kp.rwin = g.rwin;
    const int cmax = g.cmax;
    // few chunks: spread them over the CUs; many: pack workgroups so that the table is amortised
    int chunks = std::min(cmax, std::max(1, (max_chunks + 255) / 256));
    if (leave_room && chunks > 1 && fill_lds_floats(NW, chunks, kp.hw, kp.rwin) * sizeof(float) + 4096 > 160 * 1024) {
        // ... unless exactly that chunk per workgroup decides whether the batch's full-size chunks (about half of
        // the upper bound: a read's last chunk is a short tail) are resident all at once (r = 30, 4 000 reads per
        // batch: 142 k reads/s with 16 chunks per workgroup, 127 k with 15)
        const int64_t big = (max_chunks + 1) / 2, wgs = fill_round_workgroups(g, chunks, n_cus);
        if (!(big <= wgs * chunks && big > wgs * (chunks - 1))) chunks--;
    }
    if (force_chunks > 0) chunks = std::min(cmax, force_chunks);
    const size_t lds = fill_lds_floats(NW, chunks, kp.hw, kp.rwin) * sizeof(float);
    *has_room = lds + 4096 <= 160 * 1024;      // other kernels' light workgroups fit beside this launch's
    // the kernel addresses its score tables by absolute LDS address (kernels.hpp: lds_abs_f32): it must not
    // own any static LDS, so that the dynamic array starts at address 0
    static const hipError_t no_static_lds = [] {
        hipFuncAttributes at;
        const hipError_t e0 = hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&fill_kernel<NWT, MAXT>));
        return e0 != hipSuccess ? e0 : (at.sharedSizeBytes == 0 ? hipSuccess : hipErrorInvalidDeviceFunction);
    }();
    if (no_static_lds != hipSuccess) return no_static_lds;
    // A persistent launch: as many workgroups as the GPU keeps resident (or fewer, if the batch is small); their
    // groups of NW waves pull the chunks of the schedule (largest first) from a device-side queue (kernels.hpp)
    const int resident = fill_round_workgroups(g, chunks, n_cus);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fill_kernel<NWT, MAXT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((fill_kernel<NWT, MAXT>), dim3(std::min((max_chunks + chunks - 1) / chunks, resident)),
                       dim3(64 * NW * chunks), lds, s, kp);
    return hipGetLastError();
}

struct OutTarget {
    uint8_t *d_out;
    const int64_t *d_out_off;
    int64_t *d_out_len;
    int32_t *d_status;
};

// device pointers to the raw batch + host copies of the three offset arrays
struct AlignArgs {
    int64_t n_reads;
    const uint8_t *d_refs;
    const int64_t *d_ref_off;
    const uint8_t *d_seqs;
    const int64_t *d_seq_off;
    const char *d_cigs;
    const int64_t *d_cig_off;
    const int64_t *h_ref_off, *h_seq_off, *h_cig_off;
    float indel_start, indel_extend;
    int max_b_rows, r;
    // host-buffer entry points (d_* above are NULL then): every group uploads its slice of these, and downloads
    // its slice of the results, around its own kernels -- the copies of one group overlap the kernels of its neighbours
    const uint8_t *h_refs = nullptr, *h_seqs = nullptr;
    const char *h_cigs = nullptr;
    char *h_out = nullptr;
    const int64_t *h_out_off = nullptr;
    int64_t *h_out_len = nullptr;
    int32_t *h_status = nullptr;
    // the output is the collapsed, standardised CIGAR text (realign_read's glue on the device, kernels.hpp
    // standardize_kernel) instead of the op string; out_len = bytes of text
    bool final_text = false;
    // device pack: instead of h_refs / h_seqs / h_cigs the heads of the BAM records; every group uploads its slice and
    // unpacks it on the device (unpack_kernels.hpp)
    const uint8_t *h_raw = nullptr;
    const int64_t *h_raw_off = nullptr;
    const CtgEntry *d_ctg = nullptr;
    int n_ctg = 0;
    // the file pipeline with the device glue: the texts compacted on the device, the used front of the compact buffer and
    // the reads' offsets copied instead of the slots (nullptr: the slots, as the public entry points promise)
    const TextCompact *compact = nullptr;
    bool staged() const { return h_out != nullptr; }
};


int64_t chunk_bound(int64_t cig_len, int max_b_rows)
{
    const int64_t cm1 = max_b_rows - 1;
    return std::max<int64_t>(1, (2 * cig_len + cm1 - 1) / cm1);
}

// Reads [g0,g1): everything from the raw bytes to the gathered output, on stream s,
// without host synchronisation.
int run_group(npore_ctx *ctx, WorkSet *w, const AlignArgs &a, int64_t g0, int64_t g1, const OutTarget &ot, int shape,
              bool overlapping)
{
    // overlapping: another group of this context is on the device.  beside_fill: ... and the fill launch it belongs to
    // left room on its CUs (launch_fill), so kernels of light shapes can run beside it.
    const bool beside_fill = overlapping && ctx->fill_has_room;
    // beside_fill: another group of this context is on the device, most likely in its fill kernel, whose persistent
    // workgroups hold nearly all the LDS and most of the vector registers of every CU until they have emptied their
    // queue.  The preparation and gather kernels of THIS group then run in shapes that find room beside a fill
    // workgroup instead of waiting for it to leave: 256-thread scans (one wave per SIMD), the LDS-free annotation
    // (prep_kernels.hpp) and the gather without its LDS tile (kernels.hpp).
    const unsigned scan_threads = beside_fill ? 256 : 1024;
    hipStream_t s = ctx->stream;      // preparation; the fill and traceback stages go to their own streams below
    const int64_t nr = g1 - g0;
    const int r = a.r;
    const int tbs = tb_stride(r);
    const int64_t cig_bytes = a.h_cig_off[g1] - a.h_cig_off[g0];
    const int64_t S_tot = a.h_seq_off[g1] - a.h_seq_off[g0], R_tot = a.h_ref_off[g1] - a.h_ref_off[g0];
    // longest chunk slice of the group: a slice never exceeds its sequence, nor max_b_rows + 1 bases
    int64_t max_len = 0;
    for (int64_t i = g0; i < g1; i++)
        max_len = std::max({max_len, a.h_seq_off[i + 1] - a.h_seq_off[i], a.h_ref_off[i + 1] - a.h_ref_off[i]});
    int64_t max_chunks = 0;
    for (int64_t k = g0; k < g1; k++) max_chunks += chunk_bound(a.h_cig_off[k + 1] - a.h_cig_off[k], a.max_b_rows);
    if (max_chunks > (1ll << 30)) return fail(NPORE_E_UNSUPPORTED, "too many chunks in one group");
    const int64_t steps_cap = 2 * cig_bytes + 512;
    const int64_t tb_words = (2 * cig_bytes + max_chunks) * tbs;

    // CIGAR tiles (prep_kernels.hpp): every read has at least one
    int64_t max_tiles = 0;
    for (int64_t k = g0; k < g1; k++)
        max_tiles += std::max<int64_t>(1, (a.h_cig_off[k + 1] - a.h_cig_off[k] + CIGAR_TILE - 1) / CIGAR_TILE);
    if (max_tiles > (1ll << 30)) return fail(NPORE_E_UNSUPPORTED, "too many CIGAR tiles in one group");
    if (int rc = w->rd_i32.ensure((size_t)(5 * nr + 16) * 4)) return rc;
    if (int rc = w->tiles.ensure((size_t)max_tiles * 24 + 64)) return rc;
    if (int rc = w->cwoff.ensure((size_t)max_chunks * 8 + 64)) return rc;
    if (int rc = w->rd_i64.ensure((size_t)(nr + 2) * 8)) return rc;
    if (int rc = w->steps.ensure(steps_cap)) return rc;
    if (int rc = w->inss.ensure((size_t)(2 * cig_bytes + nr + 16) * 4)) return rc;
    if (int rc = w->descs.ensure((size_t)max_chunks * sizeof(ChunkDesc))) return rc;
    if (int rc = w->sched.ensure((size_t)max_chunks * 4)) return rc;
    if (int rc = w->hist.ensure((size_t)(a.max_b_rows + 2) * 4)) return rc;
    if (int rc = w->counters.ensure(64)) return rc;
    if (int rc = w->seqw.ensure((size_t)(S_tot + max_chunks + 16) * 4)) return rc;
    if (int rc = w->refw.ensure((size_t)(R_tot + max_chunks + 16) * 16)) return rc;
    if (int rc = w->refl.ensure((size_t)(R_tot + max_chunks + 16) * 8)) return rc;
    if (int rc = w->tb.ensure((size_t)tb_words * 4 + 64)) return rc;
#if defined(NPORE_EXPERIMENTS)
    if (std::getenv("NPORE_DBGMAT") && w->dbg.p) (void)hipMemsetAsync(w->dbg.p, 0, w->dbg.cap, ctx->stream);     // (steps of the compiled path leave zeros)
    if (std::getenv("NPORE_DBGMAT")) { if (int rc = w->dbg.ensure((size_t)tb_words * 4 * (size_t)std::max(1, std::atoi(std::getenv("NPORE_DBGMAT"))) + 64)) return rc; }
#endif
    if (int rc = w->cout_.ensure(((size_t)(S_tot + R_tot) + 64) * 4)) return rc;
    if (int rc = w->cnruns.ensure((size_t)max_chunks * 4 + 64)) return rc;
    if (int rc = w->clen.ensure((size_t)max_chunks * 4 + 64)) return rc;
    if (int rc = w->cstat.ensure((size_t)max_chunks * 4 + 64)) return rc;

    PrepParams pp;
    pp.n_reads = nr;
    pp.refs = a.d_refs; pp.ref_off = a.d_ref_off + g0;
    pp.seqs = a.d_seqs; pp.seq_off = a.d_seq_off + g0;
    pp.cigs = a.d_cigs; pp.cig_off = a.d_cig_off + g0;
    OutTarget got = ot;                  // where the gather writes, and the index of this group's first read in it
    int64_t out_read_base = g0;
    w->staged = a.staged();
    if (a.staged()) {
        // upload the group's slice: bases + CIGAR ops as they lie, the four offset arrays rebased to the slice
        const int64_t out_bytes = a.h_out_off[g1] - a.h_out_off[g0];
        if (int rc = w->h_off.ensure((size_t)5 * (nr + 1) * 8)) return rc;
        if (int rc = w->in_refs.ensure((size_t)R_tot + 64)) return rc;
        if (int rc = w->in_seqs.ensure((size_t)S_tot + 64)) return rc;
        if (int rc = w->in_cigs.ensure((size_t)cig_bytes + 64)) return rc;
        if (int rc = w->in_off.ensure((size_t)5 * (nr + 1) * 8)) return rc;
        if (int rc = w->out.ensure((size_t)out_bytes + 64)) return rc;
        if (int rc = w->out_len.ensure((size_t)nr * 8)) return rc;
        if (int rc = w->status.ensure((size_t)nr * 4)) return rc;
        int64_t *ho = w->h_off.as<int64_t>(), *hro = ho, *hso = ho + (nr + 1), *hco = ho + 2 * (nr + 1), *hoo = ho + 3 * (nr + 1);
        for (int64_t i = 0; i <= nr; i++) {
            hro[i] = a.h_ref_off[g0 + i] - a.h_ref_off[g0]; hso[i] = a.h_seq_off[g0 + i] - a.h_seq_off[g0];
            hco[i] = a.h_cig_off[g0 + i] - a.h_cig_off[g0]; hoo[i] = a.h_out_off[g0 + i] - a.h_out_off[g0];
        }
        const int64_t raw_bytes = a.h_raw ? a.h_raw_off[g1] - a.h_raw_off[g0] : 0;
        if (a.h_raw) {
            int64_t *hwo = ho + 4 * (nr + 1);
            for (int64_t i = 0; i <= nr; i++) hwo[i] = a.h_raw_off[g0 + i] - a.h_raw_off[g0];
            if (int rc = w->in_raw.ensure((size_t)raw_bytes + 64)) return rc;
        }
        HIP_TRY(hipEventRecord(w->evc[0], s));
        if (a.h_raw) {
            HIP_TRY(hipMemcpyAsync(w->in_raw.p, a.h_raw + a.h_raw_off[g0], (size_t)raw_bytes, hipMemcpyHostToDevice, s));
        } else {
            HIP_TRY(hipMemcpyAsync(w->in_refs.p, a.h_refs + a.h_ref_off[g0], (size_t)R_tot, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(w->in_seqs.p, a.h_seqs + a.h_seq_off[g0], (size_t)S_tot, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(w->in_cigs.p, a.h_cigs + a.h_cig_off[g0], (size_t)cig_bytes, hipMemcpyHostToDevice, s));
        }
        HIP_TRY(hipMemcpyAsync(w->in_off.p, ho, (size_t)5 * (nr + 1) * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(w->evc[1], s));
        const int64_t *d_off = w->in_off.as<int64_t>();
        if (a.h_raw) {          // align()'s three inputs from the record heads (unpack_kernels.hpp), where the copies above would have put them
            UnpackParams up;
            up.raw = w->in_raw.as<uint8_t>(); up.raw_off = d_off + 4 * (nr + 1);
            up.ctg = a.d_ctg; up.n_ctg = a.n_ctg;
            up.refs = w->in_refs.as<uint8_t>(); up.ref_off = d_off;
            up.seqs = w->in_seqs.as<uint8_t>(); up.seq_off = d_off + (nr + 1);
            up.cigs = w->in_cigs.as<char>(); up.cig_off = d_off + 2 * (nr + 1);
            up.n_reads = nr;
            hipLaunchKernelGGL(unpack_records_kernel, dim3((unsigned)nr), dim3(256), 0, s, up);
            HIP_TRY(hipGetLastError());
        }
        pp.refs = w->in_refs.as<uint8_t>(); pp.ref_off = d_off;
        pp.seqs = w->in_seqs.as<uint8_t>(); pp.seq_off = d_off + (nr + 1);
        pp.cigs = w->in_cigs.as<char>(); pp.cig_off = d_off + 2 * (nr + 1);
        got = OutTarget{w->out.as<uint8_t>(), d_off + 3 * (nr + 1), w->out_len.as<int64_t>(), w->status.as<int32_t>()};
        out_read_base = 0;
    }
    pp.max_b_rows = a.max_b_rows; pp.r = r; pp.tbstride = tbs; pp.max_n = ctx->max_n; pp.max_l = ctx->max_l;
    pp.max_chunks = (int)max_chunks;
    int32_t *i32 = w->rd_i32.as<int32_t>();
    pp.rd_nsteps = i32;
    pp.rd_nchunks = i32 + nr;
    pp.rd_status = i32 + 2 * nr;
    pp.rd_chunk_first = i32 + 3 * nr;   // nr + 1 entries
    pp.rd_tile_first = i32 + 4 * nr + 4;   // nr + 1 entries
    pp.tile_cnt = w->tiles.as<int4>();
    pp.tile_base = reinterpret_cast<int2 *>(w->tiles.as<char>() + (size_t)max_tiles * 16);
    pp.rd_steps_off = w->rd_i64.as<int64_t>();
    pp.steps = w->steps.as<uint8_t>();
    pp.inss = w->inss.as<int32_t>();
    pp.descs = w->descs.as<ChunkDesc>();
    pp.sched = w->sched.as<int32_t>();
    pp.hist = w->hist.as<int32_t>();
    pp.counters = w->counters.as<int32_t>();
    pp.seqw = w->seqw.as<uint32_t>();
    pp.refw = w->refw.as<uint4>();
    pp.refl = w->refl.as<uint2>();

    if (int rc = w->h_cnt.ensure(64)) return rc;
    HIP_TRY(hipEventRecord(w->ev[0], s));
    HIP_TRY(hipMemsetAsync(pp.hist, 0, (size_t)(a.max_b_rows + 2) * 4, s));
    const unsigned rd_blocks = (unsigned)((nr + 3) / 4), ch_blocks = (unsigned)((max_chunks + 255) / 256);
    const unsigned tile_blocks = (unsigned)((max_tiles + 3) / 4);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(scan_threads), 0, s, pp);
    hipLaunchKernelGGL(cigar_tile_kernel, dim3(tile_blocks), dim3(256), 0, s, pp);
    hipLaunchKernelGGL(cigar_scan_kernel, dim3(rd_blocks), dim3(256), 0, s, pp);
    hipLaunchKernelGGL(read_scan_kernel, dim3(1), dim3(scan_threads), 0, s, pp);
    hipLaunchKernelGGL(expand_path_kernel, dim3(tile_blocks), dim3(256), 0, s, pp);
    hipLaunchKernelGGL(make_chunks_kernel, dim3(ch_blocks), dim3(256), 0, s, pp);
    hipLaunchKernelGGL(chunk_scan_kernel, dim3(1), dim3(scan_threads), 0, s, pp);
    hipLaunchKernelGGL(sched_scatter_kernel, dim3(ch_blocks), dim3(256), 0, s, pp);
    // n-polymer annotation + word packing: one wave per (chunk, sequence), registers only (annot_wave.hpp) -- the same
    // launch whether the GPU is empty or a fill kernel holds the CUs' LDS
    if (ctx->max_n == MAX_PERIOD) hipLaunchKernelGGL(annotate_wave_kernel<true>, dim3((unsigned)(2 * max_chunks)), dim3(64), 0, s, pp);
    else hipLaunchKernelGGL(annotate_wave_kernel<false>, dim3((unsigned)(2 * max_chunks)), dim3(64), 0, s, pp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(w->ev[1], s));
    // ---- fill: behind this group's preparation.  Consecutive groups alternate between two streams: their fill
    // kernels share nothing, so the next one's persistent workgroups move onto the CUs that this one's leave as its
    // last chunks run out -- the tail of one launch is filled by the head of the next (C2, steps back to back:
    // 17.1 ms per step against the 18.0 ms one fill kernel takes alone; one stream: 18.1)
    s = ctx->s_fill[ctx->next_fill];
    if (ctx->fill_streams == 2) ctx->next_fill ^= 1;
    HIP_TRY(hipStreamWaitEvent(s, w->ev[1], 0));
    HIP_TRY(hipEventRecord(w->ev[2], s));

    KParams kp;
    kp.descs = pp.descs;
    kp.sched = pp.sched;
    kp.n_chunks = pp.counters;
    kp.queue = pp.counters + 2;
    kp.steps = pp.steps;
    kp.inss = pp.inss;
    kp.seqw = pp.seqw;
    kp.refw = pp.refw;
    kp.refl = pp.refl;
    kp.tb = w->tb.as<uint32_t>();
    kp.dbg = w->dbg.as<uint32_t>();
    kp.sub_scores = ctx->d_sub;
    kp.np_scores = ctx->d_np;
    kp.max_n = ctx->max_n;
    kp.max_l = ctx->max_l;
    kp.r = r;
    kp.tbstride = tbs;
    kp.indel_start = a.indel_start;
    kp.indel_extend = a.indel_extend;
    hipError_t e = hipSuccess;
    const int mc = (int)max_chunks;
    switch (shape) {
        case 1: e = launch_fill<1>(kp, mc, ctx->force_chunks, ctx->n_cus, s, overlapping, &ctx->fill_has_room); break;
        case 2: e = launch_fill<2>(kp, mc, ctx->force_chunks, ctx->n_cus, s, overlapping, &ctx->fill_has_room); break;
        case 3: e = launch_fill<3>(kp, mc, ctx->force_chunks, ctx->n_cus, s, overlapping, &ctx->fill_has_room); break;
        case 4: e = launch_fill<4>(kp, mc, ctx->force_chunks, ctx->n_cus, s, overlapping, &ctx->fill_has_room); break;
        case 5: e = launch_fill<5>(kp, mc, ctx->force_chunks, ctx->n_cus, s, overlapping, &ctx->fill_has_room); break;
        case 6: e = launch_fill<6>(kp, mc, ctx->force_chunks, ctx->n_cus, s, overlapping, &ctx->fill_has_room); break;
        case 7: e = launch_fill<7>(kp, mc, ctx->force_chunks, ctx->n_cus, s, overlapping, &ctx->fill_has_room); break;
        case 8: e = launch_fill<8>(kp, mc, ctx->force_chunks, ctx->n_cus, s, overlapping, &ctx->fill_has_room); break;
        case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16:
            e = launch_fill<0>(kp, mc, ctx->force_chunks, ctx->n_cus, s, overlapping, &ctx->fill_has_room); break;
        default: return fail(NPORE_E_UNSUPPORTED, "unsupported waves-per-chunk count");
    }
    if (e != hipSuccess) return fail(NPORE_E_HIP, std::string("fill launch: ") + hipGetErrorString(e));
    HIP_TRY(hipEventRecord(w->ev[3], s));
    // ---- traceback + gather: behind this group's fill, beside the next group's
    s = ctx->s_post;
    HIP_TRY(hipStreamWaitEvent(s, w->ev[3], 0));
    HIP_TRY(hipEventRecord(w->ev[4], s));

    TParams tp;
    tp.descs = pp.descs;
    tp.n_chunks = pp.counters;
    tp.tb = kp.tb;
    tp.inss = pp.inss;
    tp.chunk_runs = w->cout_.as<uint32_t>();
    tp.chunk_nruns = w->cnruns.as<int32_t>();
    tp.chunk_len = w->clen.as<int32_t>();
    tp.chunk_status = w->cstat.as<int32_t>();
    tp.r = r;
    tp.tbstride = tbs;
    // (10 kb reads: 0.53 ms at 500 chunk slots, 0.83 ms at 8 000)
    hipLaunchKernelGGL(traceback_rows_kernel, dim3((unsigned)max_chunks), dim3(64), 0, s, tp);
    HIP_TRY(hipGetLastError());

    GParams gp;
    gp.descs = pp.descs;
    gp.read_first_chunk = pp.rd_chunk_first;
    gp.chunk_runs = tp.chunk_runs;
    gp.chunk_nruns = tp.chunk_nruns;
    gp.chunk_len = tp.chunk_len;
    gp.chunk_status = tp.chunk_status;
    gp.read_status_in = pp.rd_status;
    gp.counters = pp.counters;
    gp.seqs = pp.seqs;
    gp.refs = pp.refs;
    gp.out = got.d_out;
    gp.out_off = got.d_out_off;
    gp.out_len = got.d_out_len;
    gp.status = got.d_status;
    gp.read_base = out_read_base;
    gp.n_reads = nr;
    gp.chunk_woff = w->cwoff.as<int64_t>();
    hipLaunchKernelGGL(gather_scan_kernel, dim3(rd_blocks), dim3(256), 0, s, gp);
    if (a.final_text) {
        StdKParams sp;
        sp.descs = pp.descs;
        sp.read_first_chunk = pp.rd_chunk_first;
        sp.chunk_runs = tp.chunk_runs;
        sp.chunk_nruns = tp.chunk_nruns;
        sp.refs = pp.refs; sp.ref_off = pp.ref_off;
        sp.seqs = pp.seqs; sp.seq_off = pp.seq_off;
        sp.out = got.d_out;
        sp.out_off = got.d_out_off;
        sp.out_len = got.d_out_len;
        sp.status = got.d_status;
        sp.read_base = out_read_base;
        sp.n_reads = nr;
        hipLaunchKernelGGL(standardize_kernel, dim3((unsigned)nr), dim3(64), 0, s, sp);      // one wavefront per read
        if (a.compact) {        // the texts to the front of the batch's compact buffer (unpack_kernels.hpp)
            if (int rc = w->coff.ensure((size_t)nr * 8 + 64)) return rc;
            if (g0 == 0) HIP_TRY(hipMemsetAsync(a.compact->d_cursor, 0, 8, s));
            CompactParams cp;
            cp.out = got.d_out;
            cp.out_off = got.d_out_off;
            cp.out_len = got.d_out_len;
            cp.read_base = out_read_base;
            cp.n_reads = nr;
            cp.ctext = a.compact->d_ctext;
            cp.cursor = a.compact->d_cursor;
            cp.coff = w->coff.as<int64_t>();
            cp.cap = a.compact->cap;
            hipLaunchKernelGGL(compact_texts_kernel, dim3((unsigned)nr), dim3(64), 0, s, cp);
        }
    } else
    // LDS of gather_kernel: one tile of ops + (when a chunk's two base slices fit beside it) the slices
    {
        const int64_t rows_max = std::min<int64_t>(max_len, a.max_b_rows) + 1;    // longest slice of any chunk
        if (beside_fill) {
            gp.slice_cap = 0;
            hipLaunchKernelGGL(gather_kernel<false>, dim3((unsigned)max_chunks), dim3(256), 0, s, gp);
        } else {
            gp.slice_cap = rows_max <= 24 * 1024 ? (int)((rows_max + 15) & ~(int64_t)15) : 0;
            const size_t glds = (size_t)GATHER_TILE + 2 * (size_t)gp.slice_cap;
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&gather_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));
            hipLaunchKernelGGL(gather_kernel<true>, dim3((unsigned)max_chunks), dim3(256), glds, s, gp);
        }
    }
    HIP_TRY(hipGetLastError());
    if (a.staged()) {                   // download the group's slice of the results behind its gather
        HIP_TRY(hipEventRecord(w->evc[2], s));
        if (a.final_text && a.compact) {
            HIP_TRY(hipMemcpyAsync(a.compact->h_coff + g0, w->coff.p, (size_t)nr * 8, hipMemcpyDeviceToHost, s));
            if (g1 == a.n_reads && a.compact->h_bytes > 0)       // the batch's last group: the front of the compact buffer
                HIP_TRY(hipMemcpyAsync(a.compact->h_ctext, a.compact->d_ctext, (size_t)a.compact->h_bytes, hipMemcpyDeviceToHost, s));
        } else {
            HIP_TRY(hipMemcpyAsync(a.h_out + a.h_out_off[g0], w->out.p, (size_t)(a.h_out_off[g1] - a.h_out_off[g0]), hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipMemcpyAsync(a.h_out_len + g0, w->out_len.p, (size_t)nr * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(a.h_status + g0, w->status.p, (size_t)nr * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(w->evc[3], s));
    }
    HIP_TRY(hipMemcpyAsync(w->h_cnt.p, w->counters.p, 8, hipMemcpyDeviceToHost, s));    // chunk count, overflow flag
    HIP_TRY(hipEventRecord(w->ev[5], s));
    return NPORE_OK;
}

// Wait for a group that was enqueued into `w`, add its stage times to the context's timing and check its counters.
int collect_group(npore_ctx *ctx, WorkSet *w)
{
    if (!w->busy) return NPORE_OK;
    w->busy = false;
    HIP_TRY(hipEventSynchronize(w->ev[5]));
    if (w->call_id != ctx->timing_call) {       // first group of a newer call: npore_last_timing starts over
        std::fill(ctx->timing, ctx->timing + 8, 0.0);
        ctx->timing_call = w->call_id;
    }
    float ms = 0;
    for (int k = 0; k < 3; k++) {
        HIP_TRY(hipEventElapsedTime(&ms, w->ev[2 * k], w->ev[2 * k + 1]));
        ctx->timing[k] += ms;
        ctx->totals[k] += ms;
    }
    if (w->staged)
        for (int k = 0; k < 2; k++) {
            HIP_TRY(hipEventElapsedTime(&ms, w->evc[2 * k], w->evc[2 * k + 1]));
            ctx->timing[3 + k] += ms;
            ctx->totals[3 + k] += ms;
        }
    ctx->timing[6] += (double)w->cells; ctx->totals[6] += (double)w->cells;
    ctx->timing[7] += 1; ctx->totals[7] += 1;
    if (w->h_cnt.as<int32_t>()[1]) return fail(NPORE_E_HIP, "internal: chunk bound exceeded");
    return NPORE_OK;
}

// Everything this context has in flight (asynchronous calls): collected oldest first.  Returns the first failure,
// including one found earlier while a work set was being recycled.
int quiesce(npore_ctx *ctx)
{
    int rc = ctx->deferred_rc;
    std::string err = ctx->deferred_err;
    for (int k = 0; k < N_SETS; k++) {
        WorkSet *w = &ctx->ws[(ctx->next_ws + k) % N_SETS];
        const int r2 = collect_group(ctx, w);
        if (r2 && !rc) { rc = r2; err = g_err; }
    }
    ctx->deferred_rc = 0;
    ctx->deferred_err.clear();
    return rc ? fail(rc, err) : NPORE_OK;
}

// The batch, group by group, through the three-stage pipeline: the groups rotate through the N_SETS work sets, so
// that group k+1 is prepared and group k-1 traced back while the fill kernel works on group k.  `user` (may be
// NULL) is the caller's stream: the batch is ordered behind what it holds now.  sync = false returns once the
// last group is enqueued (results complete when npore_ctx_wait returns, or for work put on `user` afterwards).
int run_core(npore_ctx *ctx, const AlignArgs &a, const OutTarget &ot, hipStream_t user, bool sync)
{
    if (a.n_reads < 0) return fail(NPORE_E_INVALID, "n_reads < 0");
    if (!ctx->d_sub || !ctx->d_np) return fail(NPORE_E_INVALID, "this context was created without penalty tables (annotation only)");
    if (a.r < 1) return fail(NPORE_E_INVALID, "r must be >= 1");
    if (a.max_b_rows < 2) return fail(NPORE_E_INVALID, "max_b_rows must be >= 2");
    if (a.max_b_rows > 60000)
        return fail(NPORE_E_UNSUPPORTED, "max_b_rows > 60000: run lengths are kept in 16 bits");
    const int shape = pick_shape(a.r);
    if (!shape) return fail(NPORE_E_UNSUPPORTED, "band half-width r > 511");
    if (a.n_reads == 0) return NPORE_OK;
    ctx->call_id++;
    if (user) {
        HIP_TRY(hipEventRecord(ctx->ev[0], user));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev[0], 0));
    }

    // groups of consecutive reads whose traceback words fit the budget.  The automatic budget is this context's
    // share of the device (contexts of one device run side by side: the file pipeline's peer, bench --inflight),
    // divided by its N_SETS work sets: 60 % of the memory divided by the live contexts, and never more than what is
    // free now plus what the context already holds.
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const int live = std::max(1, g_live_ctx[ctx->device & 15].load());
    size_t held = 0;
    for (const auto &w0 : ctx->ws) held += w0.tb.cap;
    const int64_t budget = ctx->tb_budget_mb > 0
                               ? ctx->tb_budget_mb * (int64_t)1048576
                               : (int64_t)(std::min(0.6 * (double)total_b / live, 0.9 * (double)(free_b + held)) / N_SETS);
    const int tbs = tb_stride(a.r);
    int64_t g0 = 0;
    int64_t max_group = a.n_reads;       // halved when a group's buffers do not fit after all
    WorkSet *last = nullptr;
    while (g0 < a.n_reads) {
        int64_t g1 = g0, acc = 0, cells = 0;
        while (g1 < a.n_reads && g1 - g0 < max_group) {
            const int64_t cl = a.h_cig_off[g1 + 1] - a.h_cig_off[g1];
            // traceback words + the per-step / per-base side arrays (steps, inss, refw, refl, seqw, runs: < 48 B per op)
            const int64_t need = (2 * cl + chunk_bound(cl, a.max_b_rows)) * tbs * 4 + 48 * cl;
            if (g1 > g0 && acc + need > budget) break;
            acc += need;
            g1++;
        }
        // a group that is not the last one holds a whole number of launch-fulls of full-size chunks (about one per
        // read): its fill kernel then ends on full chains instead of a sparse tail
        if (g1 < a.n_reads) {
            FillGeom fg;
            if (fill_geometry(a.r, fg)) {
                const int64_t full = (int64_t)fill_round_workgroups(fg, fg.cmax, ctx->n_cus) * fg.cmax;
                if (g1 - g0 > full) g1 = g0 + (g1 - g0) / full * full;
            }
        }
        cells = (a.h_seq_off[g1] - a.h_seq_off[g0] + a.h_ref_off[g1] - a.h_ref_off[g0] + (g1 - g0)) * (2 * a.r + 1);
        WorkSet *w = &ctx->ws[ctx->next_ws];
        if (int rc = collect_group(ctx, w)) {        // the set's previous group (N_SETS groups back) has to be through
            if (!ctx->deferred_rc) { ctx->deferred_rc = rc; ctx->deferred_err = g_err; }
        }
        // (the other work set still busy: its group is in the fill or traceback stage while this one is prepared,
        // and this group's gather will most likely run while the next one's fill is on the GPU)
        bool beside = false;
        for (int k = 1; k < N_SETS; k++) beside |= ctx->coresident && ctx->ws[(ctx->next_ws + k) % N_SETS].busy;
        if (int rc = run_group(ctx, w, a, g0, g1, ot, shape, beside)) {
            // drain what is in flight; a failure found there (an earlier group of this call, or of a previous
            // sync = 0 call) is the older one and must not be lost: it stays deferred / is what the call returns
            const std::string this_err = g_err;
            const int older = quiesce(ctx);
            if (rc == NPORE_E_NOMEM && g1 - g0 > 1) {          // another context got there first: smaller groups
                if (older) { ctx->deferred_rc = older; ctx->deferred_err = g_err; }
                max_group = (g1 - g0) / 2;
                continue;
            }
            return older ? older : fail(rc, this_err);
        }
        for (auto &o : ctx->ws)
            if (&o != w && !o.busy && o.tb.cap < w->tb.cap) o.presize_like(*w);
        w->busy = true;
        w->cells = cells;
        w->call_id = ctx->call_id;
        ctx->last_ws = last = w;
        ctx->next_ws = (ctx->next_ws + 1) % N_SETS;
        g0 = g1;
    }
    if (sync) return quiesce(ctx);
    if (user && last) HIP_TRY(hipStreamWaitEvent(user, last->ev[5], 0));
    return NPORE_OK;
}

}  // namespace

extern "C" {

int npore_abi_version(void) { return NPORE_ABI_VERSION; }
#if defined(NPORE_EXPERIMENTS) && defined(NPORE_STATS)
extern "C" int npore_debug_stats(unsigned long long *out, int reset)
{
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(npore::g_npore_stats), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(npore::g_npore_stats), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif
const char *npore_last_error(void) { return g_err.c_str(); }

int npore_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, i) == hipSuccess && std::strncmp(pr.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

npore_ctx *npore_ctx_create(const float *sub_scores, const float *np_scores, int max_n, int max_l, int device_id)
try {
    const bool tables = sub_scores && np_scores;
    if ((!tables && (sub_scores || np_scores)) || max_n < 1 || max_n > MAX_PERIOD || max_l < 2 || max_l > 127) {   // repeat counts travel in 7-bit fields (layout.hpp, annotate planes)
        fail(NPORE_E_INVALID, "npore_ctx_create: need both tables (or neither: annotation-only context), 1 <= max_n <= 6, 2 <= max_l <= 127");
        return nullptr;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0 || device_id < 0 || device_id >= n) {
        fail(NPORE_E_NODEVICE, "npore_ctx_create: no HIP device " + std::to_string(device_id) +
                                   " (this library has no CPU path)");
        return nullptr;
    }
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device_id) != hipSuccess || std::strncmp(pr.gcnArchName, "gfx950", 6) != 0) {
        fail(NPORE_E_NODEVICE, std::string("npore_ctx_create: device is not gfx950: ") + pr.gcnArchName);
        return nullptr;
    }
    auto *ctx = new npore_ctx();
    g_live_ctx[device_id & 15]++;
    ctx->device = device_id;
    ctx->n_cus = pr.multiProcessorCount;
    if (const char *e = std::getenv("NPORE_DEVICE_GLUE")) ctx->device_glue = std::atoi(e) != 0;      // (A/B of the BAM -> SAM pipeline: scripts/bench_realign.py)
    ctx->max_n = max_n;
    ctx->max_l = max_l;
    const size_t np_elems = (size_t)max_n * (max_l + 1) * (max_l + 1);
    bool ok = hipSetDevice(device_id) == hipSuccess && hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&ctx->s_fill[0], hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&ctx->s_fill[1], hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&ctx->s_post, hipStreamNonBlocking) == hipSuccess;
    if (tables) {
        ctx->h_sub.assign(sub_scores, sub_scores + 25);
        ctx->h_np.assign(np_scores, np_scores + np_elems);
        ok = ok && hipMalloc((void **)&ctx->d_sub, 25 * sizeof(float)) == hipSuccess &&
             hipMalloc((void **)&ctx->d_np, np_elems * sizeof(float)) == hipSuccess &&
             hipMemcpy(ctx->d_sub, sub_scores, 25 * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(ctx->d_np, np_scores, np_elems * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    }
    for (auto &e : ctx->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    for (auto &w : ctx->ws) {
        for (auto &e : w.ev) ok = ok && hipEventCreate(&e) == hipSuccess;
        for (auto &e : w.evc) ok = ok && hipEventCreate(&e) == hipSuccess;
    }
    if (!ok) {
        fail(NPORE_E_HIP, "npore_ctx_create: HIP initialisation failed");
        npore_ctx_destroy(ctx);
        return nullptr;
    }
    return ctx;
}
NPORE_CATCH_PTR

void npore_ctx_destroy(npore_ctx *ctx)
{
    if (!ctx) return;
    g_live_ctx[ctx->device & 15]--;
    npore_ctx_destroy(ctx->peer);
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();       // nothing of this context may still be running
    for (DevBuf *b : {&ctx->in_refs, &ctx->in_seqs, &ctx->in_cigs, &ctx->in_off, &ctx->out, &ctx->out_off, &ctx->out_len, &ctx->status,
                      &ctx->d_fasta, &ctx->d_ctg})
        b->release();
    for (auto &w : ctx->ws) {
        for (DevBuf *b : w.all) b->release();
        w.h_cnt.release();
        w.h_off.release();
        for (auto &e : w.ev)
            if (e) (void)hipEventDestroy(e);
        for (auto &e : w.evc)
            if (e) (void)hipEventDestroy(e);
    }
    ctx->h_offs.release();
    if (ctx->d_sub) (void)hipFree(ctx->d_sub);
    if (ctx->d_np) (void)hipFree(ctx->d_np);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {ctx->stream, ctx->s_fill[0], ctx->s_fill[1], ctx->s_post})
        if (st) (void)hipStreamDestroy(st);
    for (auto *sp : ctx->slots) delete sp;
    delete ctx;
}

static int align_batch_host(npore_ctx *ctx, int64_t n_reads, const uint8_t *refs, const int64_t *ref_off,
                            const uint8_t *seqs, const int64_t *seq_off, const char *cigars, const int64_t *cig_off,
                            float indel_start, float indel_extend, int max_b_rows, int r, char *out,
                            const int64_t *out_off, int64_t *out_len, int32_t *status, bool sync, bool final_text = false)
{
    if (!ctx) return fail(NPORE_E_INVALID, "null context");
    if (n_reads < 0) return fail(NPORE_E_INVALID, "n_reads < 0");
    if (n_reads == 0) return NPORE_OK;
    if (!refs || !seqs || !cigars || !out || !ref_off || !seq_off || !cig_off || !out_off || !out_len || !status)
        return fail(NPORE_E_INVALID, "null argument");
    if (out_off[n_reads] < out_off[0]) return fail(NPORE_E_INVALID, "out_off not ascending");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->deferred_rc) return quiesce(ctx);   // a group of an earlier asynchronous call failed
    // every group of reads uploads its own slice and downloads its own results (run_group): the copies of one
    // group run beside the kernels of its neighbours, and the caller's arrays are used as they are
    AlignArgs a{n_reads, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ref_off, seq_off, cig_off,
                indel_start, indel_extend, max_b_rows, r};
    a.h_refs = refs; a.h_seqs = seqs; a.h_cigs = cigars;
    a.h_out = out; a.h_out_off = out_off; a.h_out_len = out_len; a.h_status = status;
    a.final_text = final_text;
    if (final_text) { a.compact = ctx->pending_compact; ctx->pending_compact = nullptr; }
    return run_core(ctx, a, OutTarget{nullptr, nullptr, nullptr, nullptr}, nullptr, sync);
}

// The same with align()'s inputs still inside BAM records: `raw` holds the heads of the records (fixed fields ... 4-bit
// bases) one after the other, raw_off[n + 1] where each starts; the three offset arrays are the sizes pack_sizes_of
// found.  Every group uploads its slice of `raw` and unpacks it on the device (unpack_kernels.hpp); the contigs are the
// context's device copy of the FASTA (device_fasta).  Asynchronous, final CIGAR text out: the file pipeline's call.
static int align_batch_raw(npore_ctx *ctx, int64_t n_reads, const uint8_t *raw, const int64_t *raw_off, const int64_t *ref_off,
                           const int64_t *seq_off, const int64_t *cig_off, float indel_start, float indel_extend, int max_b_rows,
                           int r, char *out, const int64_t *out_off, int64_t *out_len, int32_t *status)
{
    if (!ctx) return fail(NPORE_E_INVALID, "null context");
    if (n_reads <= 0) return n_reads < 0 ? fail(NPORE_E_INVALID, "n_reads < 0") : NPORE_OK;
    if (!raw || !raw_off || !out || !ref_off || !seq_off || !cig_off || !out_off || !out_len || !status || !ctx->d_ctg.p)
        return fail(NPORE_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->deferred_rc) return quiesce(ctx);
    AlignArgs a{n_reads, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ref_off, seq_off, cig_off,
                indel_start, indel_extend, max_b_rows, r};
    a.h_raw = raw; a.h_raw_off = raw_off;
    a.d_ctg = ctx->d_ctg.as<CtgEntry>(); a.n_ctg = ctx->n_ctg;
    a.h_out = out; a.h_out_off = out_off; a.h_out_len = out_len; a.h_status = status;
    a.final_text = true;
    a.compact = ctx->pending_compact;
    ctx->pending_compact = nullptr;
    return run_core(ctx, a, OutTarget{nullptr, nullptr, nullptr, nullptr}, nullptr, false);
}

// The FASTA on the device (once per FASTA and context) and, per BAM reference, where its contig lies there.
static int device_fasta(npore_ctx *ctx, const npore_bam *b, const npore_fasta *fa, const int32_t *fasta_of_ref)
{
    const size_t bytes = fa->off.empty() ? 0 : (size_t)fa->off.back();
    if (ctx->d_fasta_serial != fa->serial || ctx->d_fasta_bytes != bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (int rc = ctx->d_fasta.ensure(bytes + 64)) return rc;
        HIP_TRY(hipMemcpy(ctx->d_fasta.p, fa->bases.p, bytes, hipMemcpyHostToDevice));
        ctx->d_fasta_serial = fa->serial;
        ctx->d_fasta_bytes = bytes;
    }
    const size_t nref = b->ref_names.size();
    std::vector<CtgEntry> tab(std::max<size_t>(1, nref), CtgEntry{nullptr, 0});
    for (size_t k = 0; k < nref; k++) {
        const int fi = fasta_of_ref[k];
        if (fi >= 0 && fi < (int)fa->names.size()) tab[k] = CtgEntry{ctx->d_fasta.as<char>() + fa->off[(size_t)fi], fa->len((size_t)fi)};
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));       // (no group of an earlier run still reads the table)
    if (int rc = ctx->d_ctg.ensure(tab.size() * sizeof(CtgEntry))) return rc;
    HIP_TRY(hipMemcpy(ctx->d_ctg.p, tab.data(), tab.size() * sizeof(CtgEntry), hipMemcpyHostToDevice));
    ctx->n_ctg = (int)nref;
    return NPORE_OK;
}

int npore_align_batch(npore_ctx *ctx, int64_t n_reads, const uint8_t *refs, const int64_t *ref_off,
                      const uint8_t *seqs, const int64_t *seq_off, const char *cigars, const int64_t *cig_off,
                      float indel_start, float indel_extend, int max_b_rows, int r, char *out,
                      const int64_t *out_off, int64_t *out_len, int32_t *status)
try {
    return align_batch_host(ctx, n_reads, refs, ref_off, seqs, seq_off, cigars, cig_off, indel_start, indel_extend,
                            max_b_rows, r, out, out_off, out_len, status, true);
}
NPORE_CATCH_INT

int npore_align_batch_async(npore_ctx *ctx, int64_t n_reads, const uint8_t *refs, const int64_t *ref_off,
                            const uint8_t *seqs, const int64_t *seq_off, const char *cigars, const int64_t *cig_off,
                            float indel_start, float indel_extend, int max_b_rows, int r, char *out,
                            const int64_t *out_off, int64_t *out_len, int32_t *status)
try {
    return align_batch_host(ctx, n_reads, refs, ref_off, seqs, seq_off, cigars, cig_off, indel_start, indel_extend,
                            max_b_rows, r, out, out_off, out_len, status, false);
}
NPORE_CATCH_INT

int npore_align_batch_cigars(npore_ctx *ctx, int64_t n_reads, const uint8_t *refs, const int64_t *ref_off,
                             const uint8_t *seqs, const int64_t *seq_off, const char *cigars, const int64_t *cig_off,
                             float indel_start, float indel_extend, int max_b_rows, int r, char *out,
                             const int64_t *out_off, int64_t *out_len, int32_t *status)
try {
    return align_batch_host(ctx, n_reads, refs, ref_off, seqs, seq_off, cigars, cig_off, indel_start, indel_extend,
                            max_b_rows, r, out, out_off, out_len, status, true, true);
}
NPORE_CATCH_INT

int npore_align_batch_device(npore_ctx *ctx, int64_t n_reads, const uint8_t *d_refs, const int64_t *d_ref_off,
                             const uint8_t *d_seqs, const int64_t *d_seq_off, const char *d_cigars,
                             const int64_t *d_cig_off, float indel_start, float indel_extend, int max_b_rows,
                             int r, char *d_out, const int64_t *d_out_off, int64_t *d_out_len,
                             int32_t *d_status, void *stream, int sync)
try {
    if (!ctx) return fail(NPORE_E_INVALID, "null context");
    if (n_reads < 0) return fail(NPORE_E_INVALID, "n_reads < 0");
    if (n_reads == 0) return NPORE_OK;
    if (!d_ref_off || !d_seq_off || !d_cig_off || !d_out_off || !d_out_len || !d_status)
        return fail(NPORE_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->deferred_rc) return quiesce(ctx);   // a group of an earlier asynchronous call failed
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    // the host only needs the three offset arrays (24 bytes per read) to size work buffers
    const int64_t n = n_reads;
    // (into page-locked memory: three truly asynchronous copies and one wait instead of three staged ones)
    if (int rc = ctx->h_offs.ensure(3 * (size_t)(n + 1) * 8)) return rc;
    int64_t *offs = ctx->h_offs.as<int64_t>();
    HIP_TRY(hipMemcpyAsync(offs, d_ref_off, (n + 1) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(offs + (n + 1), d_seq_off, (n + 1) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(offs + 2 * (n + 1), d_cig_off, (n + 1) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    AlignArgs a{n, d_refs, d_ref_off, d_seqs, d_seq_off, d_cigars, d_cig_off,
                offs, offs + (n + 1), offs + 2 * (n + 1),
                indel_start, indel_extend, max_b_rows, r};
    OutTarget ot{reinterpret_cast<uint8_t *>(d_out), d_out_off, d_out_len, d_status};
    return run_core(ctx, a, ot, (hipStream_t)stream, sync != 0);
}
NPORE_CATCH_INT

int npore_ctx_wait(npore_ctx *ctx)
try {
    if (!ctx) return fail(NPORE_E_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return quiesce(ctx);
}
NPORE_CATCH_INT

// get_np_info() of the sequences in ctx->in_seqs (host offsets `off`, n of them) by one wave per segment (annot_wave.hpp
// np_info_wave_kernel): raw (L, L_IDX) values into out32, or the byte planes of the region kernels into planes
static int launch_np_info(npore_ctx *ctx, hipStream_t s, const int64_t *off, int64_t n, int32_t *out32, uint8_t *planes)
{
    NpInfoParams q;
    q.max_n = ctx->max_n;
    q.max_l = ctx->max_l;
    q.seg = 16384;
    int warm = 0;
    for (int k = 1; k <= ctx->max_n; k++) warm += (ctx->max_l + 2) * k;
    q.warm = (warm + 63) & ~63;
    std::vector<int2> work;
    for (int64_t k = 0; k < n; k++)
        for (int64_t g = 0; g * q.seg < off[k + 1] - off[k]; g++) work.push_back(make_int2((int)k, (int)g));
    if (work.empty()) return NPORE_OK;
    if (int rc = ctx->in_off.ensure((size_t)(n + 1) * 8)) return rc;
    if (int rc = ctx->ws[0].rd_i32.ensure(work.size() * 8 + 16)) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->in_off.p, off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ctx->ws[0].rd_i32.p, work.data(), work.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));                // (`work` is pageable host memory that dies with this call)
    q.seqs = ctx->in_seqs.as<uint8_t>();
    q.seq_off = ctx->in_off.as<int64_t>();
    q.work = ctx->ws[0].rd_i32.as<int2>();
    q.n_work = (int)work.size();
    q.out32 = out32;
    q.planes = planes;
    const dim3 grid((unsigned)work.size()), block(64);
    const bool alln = ctx->max_n == MAX_PERIOD;
    if (out32) {
        if (alln) hipLaunchKernelGGL((np_info_wave_kernel<ANNOT_RAW, true>), grid, block, 0, s, q);
        else hipLaunchKernelGGL((np_info_wave_kernel<ANNOT_RAW, false>), grid, block, 0, s, q);
    } else {
        if (alln) hipLaunchKernelGGL((np_info_wave_kernel<ANNOT_PLANES, true>), grid, block, 0, s, q);
        else hipLaunchKernelGGL((np_info_wave_kernel<ANNOT_PLANES, false>), grid, block, 0, s, q);
    }
    HIP_TRY(hipGetLastError());
    return NPORE_OK;
}

int npore_get_np_info(npore_ctx *ctx, const uint8_t *seq, int64_t len, int32_t *out)
try {
    if (!ctx || (len > 0 && (!seq || !out))) return fail(NPORE_E_INVALID, "null argument");
    if (len <= 0) return NPORE_OK;
    if (len > (1ll << 30)) return fail(NPORE_E_UNSUPPORTED, "sequence too long");
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = quiesce(ctx)) return rc;
    const int mn = ctx->max_n;
    const size_t out_bytes = (size_t)len * 2 * mn * 4;
    // work buffers of the align path are reused (nothing else runs on this context meanwhile): grow-only, kept
    if (int rc = ctx->in_seqs.ensure((size_t)len + 16)) return rc;
    if (int rc = ctx->out.ensure(out_bytes)) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->in_seqs.p, seq, (size_t)len, hipMemcpyHostToDevice, s));
    const int64_t one_off[2] = {0, len};
    int32_t *L = ctx->out.as<int32_t>();
    if (int rc = launch_np_info(ctx, s, one_off, 1, L, nullptr)) return rc;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, L, out_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return NPORE_OK;
}
NPORE_CATCH_INT

int npore_np_regions(npore_ctx *ctx, const uint8_t *seqs, const int64_t *seq_off, int64_t n_slices, int64_t *counts,
                     const int32_t **pos, const int32_t **reps, int64_t *total)
try {
    if (!ctx || n_slices < 0 || (n_slices > 0 && (!seqs || !seq_off || !counts)) || !pos || !reps || !total)
        return fail(NPORE_E_INVALID, "null argument");
    *pos = *reps = nullptr;
    *total = 0;
    if (n_slices == 0) return NPORE_OK;
    if (n_slices > (1 << 24)) return fail(NPORE_E_UNSUPPORTED, "too many slices in one call");
    const int64_t bases = seq_off[n_slices] - seq_off[0];
    for (int64_t k = 0; k < n_slices; k++) {
        const int64_t l = seq_off[k + 1] - seq_off[k];
        if (l < 0 || l >= (1ll << 30)) return fail(NPORE_E_INVALID, "slice length out of range");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = quiesce(ctx)) return rc;
    const int mn = ctx->max_n;
    const size_t m = (size_t)mn * n_slices;
    // work buffers of the align path are reused (nothing else runs on this context meanwhile)
    if (int rc = ctx->in_seqs.ensure((size_t)bases + 16)) return rc;
    if (int rc = ctx->in_off.ensure((size_t)(n_slices + 1) * 8)) return rc;
    if (int rc = ctx->ws[0].seql.ensure((size_t)bases * mn + 64)) return rc;
    if (int rc = ctx->ws[0].rd_i64.ensure((m + 2) * 8)) return rc;
    std::vector<int64_t> off((size_t)n_slices + 1);
    for (int64_t k = 0; k <= n_slices; k++) off[k] = seq_off[k] - seq_off[0];
    hipStream_t s = ctx->stream;
    HIP_TRY(hipMemcpyAsync(ctx->in_seqs.p, seqs + seq_off[0], (size_t)bases, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ctx->in_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, s));
    RegionParams rp;
    rp.seqs = ctx->in_seqs.as<uint8_t>();
    rp.seq_off = ctx->in_off.as<int64_t>();
    rp.n_slices = (int)n_slices;
    rp.max_n = mn;
    rp.max_l = ctx->max_l;
    rp.planes = ctx->ws[0].seql.as<uint8_t>();
    rp.counts = ctx->ws[0].rd_i64.as<int64_t>();
    rp.out_pos = rp.out_reps = nullptr;
    if (int rc = launch_np_info(ctx, s, off.data(), n_slices, nullptr, ctx->ws[0].seql.as<uint8_t>())) return rc;
    hipLaunchKernelGGL(region_count_kernel, dim3((unsigned)n_slices), dim3(1024), 0, s, rp);
    hipLaunchKernelGGL(region_scan_kernel, dim3(1), dim3(1024), 0, s, rp);
    HIP_TRY(hipGetLastError());
    std::vector<int64_t> offs(m + 1);
    HIP_TRY(hipMemcpyAsync(offs.data(), rp.counts, (m + 1) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const int64_t tot = offs[m];
    for (size_t k = 0; k < m; k++) counts[k] = offs[k + 1] - offs[k];
    ctx->regions.resize((size_t)tot * 2);
    if (tot > 0) {
        if (int rc = ctx->out.ensure((size_t)tot * 8)) return rc;
        rp.out_pos = ctx->out.as<int32_t>();
        rp.out_reps = rp.out_pos + tot;
        hipLaunchKernelGGL(region_emit_kernel, dim3((unsigned)n_slices, (unsigned)mn), dim3(256), 0, s, rp);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(ctx->regions.data(), rp.out_pos, (size_t)tot * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    *pos = ctx->regions.data();
    *reps = ctx->regions.data() + tot;
    *total = tot;
    return NPORE_OK;
}
NPORE_CATCH_INT

int npore_last_timing(npore_ctx *ctx, double *ms, int n)
{
    if (!ctx || !ms) return fail(NPORE_E_INVALID, "null argument");
    for (int i = 0; i < n && i < 8; i++) ms[i] = ctx->timing[i];
    return NPORE_OK;
}

int npore_total_timing(npore_ctx *ctx, double *ms, int n)
{
    if (!ctx || !ms) return fail(NPORE_E_INVALID, "null argument");
    for (int i = 0; i < n && i < 8; i++) ms[i] = ctx->totals[i];
    return NPORE_OK;
}

int64_t npore_round_chunks(npore_ctx *ctx, int r)
{
    FillGeom g;
    if (!ctx || !fill_geometry(r, g)) return 0;
    return (int64_t)fill_round_workgroups(g, g.cmax, ctx->n_cus) * g.cmax;
}

int npore_fill_shape(npore_ctx *ctx, int r, int32_t *out, int n)
{
    FillGeom g;
    if (!ctx || !out) return fail(NPORE_E_INVALID, "null argument");
    if (!fill_geometry(r, g)) return fail(NPORE_E_UNSUPPORTED, "band half-width r > 511");
    const size_t lds = fill_lds_floats(g.nw, g.cmax, g.hw, g.rwin) * sizeof(float);
    const int wg_per_cu = std::max(1, std::min((int)((160 * 1024) / std::max<size_t>(lds, 1)), 2048 / (64 * g.nw * g.cmax)));
    const int32_t v[5] = {g.nw, g.cmax, wg_per_cu, fill_round_workgroups(g, g.cmax, ctx->n_cus), (int32_t)lds};
    for (int i = 0; i < n && i < 5; i++) out[i] = v[i];
    return NPORE_OK;
}

int npore_ctx_set(npore_ctx *ctx, const char *key, int64_t value)
try {
    if (!ctx || !key) return fail(NPORE_E_INVALID, "null argument");
    const std::string k(key);
    if (k == "tb_budget_mb") ctx->tb_budget_mb = value;
    else if (k == "force_chunks") ctx->force_chunks = (int)value;
    else if (k == "coresident") ctx->coresident = value != 0;
    else if (k == "device_glue") ctx->device_glue = value != 0;
    else if (k == "device_pack") ctx->device_pack = value != 0;
    else if (k == "fill_streams") { if (value < 1 || value > 2) return fail(NPORE_E_INVALID, "fill_streams: 1 or 2"); ctx->fill_streams = (int)value; }
    else return fail(NPORE_E_INVALID, "unknown key " + k);
    return NPORE_OK;
}
NPORE_CATCH_INT

static int standardize_batch_impl(bool expanded, int64_t n_reads, const char *alns, const int64_t *aln_off, const uint8_t *refs,
                            const int64_t *ref_off, const uint8_t *seqs, const int64_t *seq_off, char *out,
                            const int64_t *out_off, int64_t *out_len, int threads)
try {
    if (n_reads < 0 || (n_reads > 0 && (!alns || !aln_off || !ref_off || !seq_off || !out || !out_off || !out_len)))
        return fail(NPORE_E_INVALID, "null argument");
    const int nt = threads > 0 ? threads : (int)std::max(1u, std::thread::hardware_concurrency());
    std::atomic<int64_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&] {
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n_reads) break;
            if (expanded) {       // never longer than the align() string
                if (aln_off[i + 1] - aln_off[i] > out_off[i + 1] - out_off[i]) { out_len[i] = -1; bad++; continue; }
                out_len[i] = standardize_expanded(alns + aln_off[i], aln_off[i + 1] - aln_off[i], refs + ref_off[i],
                                                  ref_off[i + 1] - ref_off[i], seqs + seq_off[i],
                                                  seq_off[i + 1] - seq_off[i], out + out_off[i]);
                continue;
            }
            const std::string c = standardize_collapsed(alns + aln_off[i], aln_off[i + 1] - aln_off[i],
                                                        refs + ref_off[i], ref_off[i + 1] - ref_off[i],
                                                        seqs + seq_off[i], seq_off[i + 1] - seq_off[i]);
            if ((int64_t)c.size() > out_off[i + 1] - out_off[i]) { out_len[i] = -1; bad++; continue; }
            std::memcpy(out + out_off[i], c.data(), c.size());
            out_len[i] = (int64_t)c.size();
        }
    };
    if (nt <= 1 || n_reads <= 1) work();
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < std::min<int64_t>(nt, n_reads); t++) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    return bad ? fail(NPORE_E_INVALID, "output slot too small") : NPORE_OK;
}
NPORE_CATCH_INT

int npore_standardize_batch(int64_t n_reads, const char *alns, const int64_t *aln_off, const uint8_t *refs,
                            const int64_t *ref_off, const uint8_t *seqs, const int64_t *seq_off, char *out,
                            const int64_t *out_off, int64_t *out_len, int threads)
{
    return standardize_batch_impl(false, n_reads, alns, aln_off, refs, ref_off, seqs, seq_off, out, out_off, out_len, threads);
}

int npore_standardize_ops_batch(int64_t n_reads, const char *alns, const int64_t *aln_off, const uint8_t *refs,
                                const int64_t *ref_off, const uint8_t *seqs, const int64_t *seq_off, char *out,
                                const int64_t *out_off, int64_t *out_len, int threads)
{
    return standardize_batch_impl(true, n_reads, alns, aln_off, refs, ref_off, seqs, seq_off, out, out_off, out_len, threads);
}

// ---- confusion matrices from pileup text (confusion.hpp) -------------------------------------------
int npore_confusion_counts(const char *lines, const int64_t *line_off, int64_t n_lines, const uint8_t *ref_codes,
                           int64_t n_ref, const char *ref_text, int64_t ref_text_len, const int32_t *np_info,
                           int64_t np_len, int max_n, int max_l, int64_t *subs, int64_t *nps, int64_t *inss,
                           int64_t *dels, int64_t *bad_lines, int threads)
try {
    if (n_lines < 0 || max_n < 1 || max_l < 1 || !subs || !nps || !inss || !dels ||
        (n_lines > 0 && (!lines || !line_off || !ref_codes || !ref_text || !np_info)))
        return fail(NPORE_E_INVALID, "bad argument");
    const size_t dim = (size_t)max_l + 1, n_nps = (size_t)max_n * dim * dim;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads > 0 ? threads : (int)std::thread::hardware_concurrency(),
                                                                 (n_lines + 4095) / 4096));
    // per-thread matrices (25 + max_n (max_l+1)^2 + 2 (max_l+1) counters), summed at the end
    std::vector<std::vector<int64_t>> acc((size_t)nt, std::vector<int64_t>(25 + n_nps + 2 * dim, 0));
    std::vector<int64_t> bad((size_t)nt, 0);
    std::atomic<int64_t> next{0};
    auto work = [&](int t) {
        int64_t *a = acc[(size_t)t].data();
        const ConfusionOut o{a, a + 25, a + 25 + n_nps, a + 25 + n_nps + dim};
        for (;;) {
            const int64_t b0 = next.fetch_add(1024);
            if (b0 >= n_lines) break;
            for (int64_t k = b0; k < std::min(n_lines, b0 + 1024); k++)
                if (!confusion_count_line(lines + line_off[k], line_off[k + 1] - line_off[k], k, ref_codes, n_ref, ref_text,
                                          ref_text_len, np_info, np_len, max_n, max_l, o))
                    bad[(size_t)t]++;
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++) pool.emplace_back(work, t);
        for (auto &th : pool) th.join();
    }
    int64_t nbad = 0;
    for (int t = 0; t < nt; t++) {
        const int64_t *a = acc[(size_t)t].data();
        for (size_t k = 0; k < 25; k++) subs[k] += a[k];
        for (size_t k = 0; k < n_nps; k++) nps[k] += a[25 + k];
        for (size_t k = 0; k < dim; k++) { inss[k] += a[25 + n_nps + k]; dels[k] += a[25 + n_nps + dim + k]; }
        nbad += bad[(size_t)t];
    }
    if (bad_lines) *bad_lines = nbad;
    return NPORE_OK;
}
NPORE_CATCH_INT

// debug / self-test entries (used by tests -m gpu)
int npore_debug_inflate(const uint8_t *in, int64_t in_len, uint8_t *out, int64_t out_len, int force)
{
    if (!in || !out || in_len < 0 || out_len < 0) return fail(NPORE_E_INVALID, "null argument");
    return inflate_block(in, (size_t)in_len, out, (size_t)out_len, force) ? 1 : 0;
}

int npore_debug_inflate_pair(const uint8_t *in_a, int64_t in_len_a, uint8_t *out_a, int64_t out_len_a, const uint8_t *in_b, int64_t in_len_b,
                             uint8_t *out_b, int64_t out_len_b, int force)
{
    if (!in_a || !out_a || !in_b || !out_b || in_len_a < 0 || out_len_a < 0 || in_len_b < 0 || out_len_b < 0) return fail(NPORE_E_INVALID, "null argument");
    // (through the readers' own entry: a list of blocks for one thread; the pair three times over, so that more lanes than
    // two -- NPORE_INFLATE_LANES -- are exercised too: every copy must agree)
    std::vector<uint8_t> ca((size_t)out_len_a * 2 + 1), cb((size_t)out_len_b * 2 + 1);
    const FastInflate::Job jobs[6] = {{in_a, (size_t)in_len_a, out_a, (size_t)out_len_a}, {in_b, (size_t)in_len_b, out_b, (size_t)out_len_b},
                                      {in_a, (size_t)in_len_a, ca.data(), (size_t)out_len_a}, {in_b, (size_t)in_len_b, cb.data(), (size_t)out_len_b},
                                      {in_b, (size_t)in_len_b, cb.data() + out_len_b, (size_t)out_len_b}, {in_a, (size_t)in_len_a, ca.data() + out_len_a, (size_t)out_len_a}};
    bool ok[6] = {false, false, false, false, false, false};
    if (force != 2) inflate_raw_fast_many(jobs, 6, ok);
    if (force != 1)
        for (int k = 0; k < 6; k++)
            if (!ok[k]) ok[k] = inflate_block(jobs[k].in, jobs[k].in_len, jobs[k].out, jobs[k].out_len, 2);
    const bool oa = ok[0], ob = ok[1];
    if (ok[2] != oa || ok[5] != oa || ok[3] != ob || ok[4] != ob) return fail(NPORE_E_INVALID, "copies of one stream disagree");
    if (oa && (std::memcmp(out_a, ca.data(), (size_t)out_len_a) || std::memcmp(out_a, ca.data() + out_len_a, (size_t)out_len_a)))
        return fail(NPORE_E_INVALID, "copies of stream a differ");
    if (ob && (std::memcmp(out_b, cb.data(), (size_t)out_len_b) || std::memcmp(out_b, cb.data() + out_len_b, (size_t)out_len_b)))
        return fail(NPORE_E_INVALID, "copies of stream b differ");
    return (oa ? 1 : 0) | (ob ? 2 : 0);
}

int64_t npore_debug_crc32(const uint8_t *p, int64_t n, uint32_t crc)
{
    if ((!p && n > 0) || n < 0) return fail(NPORE_E_INVALID, "null argument");
    return (int64_t)crc32_fast(crc, p, (size_t)n);
}

int npore_debug_dpp(uint32_t *out128)
{
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 128 * 4));
    hipLaunchKernelGGL(dpp_selftest_kernel, dim3(1), dim3(64), 0, 0, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out128, d, 128 * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d));
    return NPORE_OK;
}

int npore_debug_divcheck(int64_t *mismatches)
{
    unsigned long long *d = nullptr, h = 0;
    HIP_TRY(hipMalloc((void **)&d, 8));
    HIP_TRY(hipMemset(d, 0, 8));
    hipLaunchKernelGGL(divcheck_kernel, dim3(256), dim3(256), 0, 0, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d));
    *mismatches = (int64_t)h;
    return NPORE_OK;
}

// Copies the device-prepared arrays of the LAST group of the last align call to the host
// (what: 0 steps, 1 inss, 2 descs, 3 seqw, 4 refw, 5 refl, 6 sched, 7 counters).
int npore_debug_fetch(npore_ctx *ctx, int what, void *dst, int64_t bytes)
{
    if (!ctx || !dst) return fail(NPORE_E_INVALID, "null argument");
    if (int rc = quiesce(ctx)) return rc;
    WorkSet *w = ctx->last_ws ? ctx->last_ws : &ctx->ws[0];
    DevBuf *b[] = {&w->steps, &w->inss, &w->descs, &w->seqw, &w->refw, &w->refl, &w->sched, &w->counters};
    if (what < 0 || what > 7) return fail(NPORE_E_INVALID, "bad selector");
    if ((size_t)bytes > b[what]->cap) return fail(NPORE_E_INVALID, "more bytes than the buffer holds");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(dst, b[what]->p, bytes, hipMemcpyDeviceToHost));
    return NPORE_OK;
}

// Debug: `bytes` of the last group's traceback words from byte `offset` on (a chunk's words start at 4 * tb_off of its
// descriptor, one row of tb_stride(r) words per anti-diagonal)
#if defined(NPORE_EXPERIMENTS)
extern "C" int npore_debug_fetch_dbg(npore_ctx *ctx, int64_t offset, void *dst, int64_t bytes)
{
    if (!ctx || !dst || offset < 0 || bytes < 0) return fail(NPORE_E_INVALID, "bad argument");
    if (int rc = quiesce(ctx)) return rc;
    WorkSet *w = ctx->last_ws ? ctx->last_ws : &ctx->ws[0];
    if ((size_t)(offset + bytes) > w->dbg.cap) return fail(NPORE_E_INVALID, "beyond the debug buffer");
    HIP_TRY(hipMemcpy(dst, static_cast<const char *>(w->dbg.p) + offset, bytes, hipMemcpyDeviceToHost));
    return NPORE_OK;
}
#endif
int npore_debug_fetch_tb(npore_ctx *ctx, int64_t offset, void *dst, int64_t bytes)
{
    if (!ctx || !dst || offset < 0 || bytes < 0) return fail(NPORE_E_INVALID, "bad argument");
    if (int rc = quiesce(ctx)) return rc;
    WorkSet *w = ctx->last_ws ? ctx->last_ws : &ctx->ws[0];
    if ((size_t)(offset + bytes) > w->tb.cap) return fail(NPORE_E_INVALID, "beyond the traceback buffer");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(dst, static_cast<const char *>(w->tb.p) + offset, bytes, hipMemcpyDeviceToHost));
    return NPORE_OK;
}


// ---- BAM ingest / SAM emit (hostio.hpp) ---------------------------------------------------------
namespace {
// header of a BAM stream at d[0 .. N): text, references; *hdr_end = offset of the first record.  -1: more bytes needed
// (streamed mode reads on), 0: corrupt, 1: ok
int bam_parse_header(npore_bam *b, const uint8_t *d, size_t N, size_t *hdr_end)
{
    if (N < 12) return -1;
    if (std::memcmp(d, "BAM\1", 4) != 0) return 0;
    size_t p = 4;
    const int64_t l_text = rdi32(&d[p]);
    p += 4;
    if (l_text < 0) return 0;
    if (p + (size_t)l_text + 4 > N) return -1;
    b->text.assign(reinterpret_cast<const char *>(&d[p]), (size_t)l_text);
    while (!b->text.empty() && b->text.back() == '\0') b->text.pop_back();
    p += (size_t)l_text;
    const int32_t n_ref = rdi32(&d[p]);
    p += 4;
    if (n_ref < 0) return 0;
    b->ref_names.clear();
    b->ref_lens.clear();
    for (int32_t k = 0; k < n_ref; k++) {
        if (p + 4 > N) return -1;
        const int32_t l_name = rdi32(&d[p]);
        if (l_name < 1) return 0;
        if (p + 8 + (size_t)l_name > N) return -1;
        b->ref_names.emplace_back(reinterpret_cast<const char *>(&d[p + 4]), (size_t)l_name - 1);
        b->ref_lens.push_back(rdi32(&d[p + 4 + (size_t)l_name]));
        p += 8 + (size_t)l_name;
    }
    *hdr_end = p;
    return 1;
}

// what selection needs of the records in d[first record .. ): validation + metadata on all cores.  `offs` = offsets of
// the records' block_size fields relative to d; the metadata is appended to the handle's arrays.
bool bam_index_records(npore_bam *b, const uint8_t *d, const std::vector<int64_t> &offs, int64_t global_base, int threads)
{
    const int64_t n = (int64_t)offs.size(), at = (int64_t)b->rec_off.size();
    b->rec_off.resize((size_t)(at + n));
    b->m_ref.resize((size_t)(at + n));
    b->m_pos.resize((size_t)(at + n));
    b->m_span.resize((size_t)(at + n));
    b->m_flag.resize((size_t)(at + n));
    std::atomic<int> corrupt{0};
    const int64_t per = 256;
    parallel_for((n + per - 1) / per, threads, [&](int64_t blk) {
        for (int64_t i = blk * per; i < std::min(n, (blk + 1) * per); i++) {
            // the variable-length parts the accessors will walk must lie inside the record
            const uint8_t *q = d + offs[(size_t)i];
            const int32_t bs = rdi32(q);
            const uint8_t *f = q + 4;
            const int64_t l_rn = f[8], n_cig = rd16(f + 12), l_seq = rdi32(f + 16);
            if (l_rn < 1 || l_seq < 0 || 32 + l_rn + 4 * n_cig + (l_seq + 1) / 2 + l_seq > bs || f[32 + l_rn - 1] != 0) { corrupt++; return; }
            const RecView r = rec_view(q);
            const int64_t span = rec_ref_len(r);
            b->rec_off[(size_t)(at + i)] = global_base + offs[(size_t)i];
            b->m_ref[(size_t)(at + i)] = r.ref_id();
            b->m_pos[(size_t)(at + i)] = r.pos();
            b->m_span[(size_t)(at + i)] = (int32_t)std::min<int64_t>(span, INT32_MAX);
            b->m_flag[(size_t)(at + i)] = (uint16_t)r.flag();
        }
    });
    return corrupt == 0;
}

// per-reference record lists and the shortcuts of npore_bam_select, from the per-record metadata
void bam_finish_index(npore_bam *b)
{
    const int32_t n_ref = (int32_t)b->ref_names.size();
    b->ref_has_reads.assign((size_t)n_ref, 0);
    b->by_ref.assign((size_t)n_ref, {});
    b->ref_sorted.assign((size_t)n_ref, 1);
    b->ref_max_len.assign((size_t)n_ref, 0);
    std::vector<int64_t> last_pos((size_t)n_ref, -1);
    const int64_t n_rec = (int64_t)b->rec_off.size();
    for (int64_t i = 0; i < n_rec; i++) {
        const int32_t rid = b->m_ref[(size_t)i];
        if (rid >= 0 && rid < n_ref) {
            const int64_t pos = b->m_pos[(size_t)i];
            b->ref_has_reads[(size_t)rid] = 1;
            b->by_ref[(size_t)rid].push_back(i);
            if (pos < last_pos[(size_t)rid]) b->ref_sorted[(size_t)rid] = 0;
            last_pos[(size_t)rid] = pos;
            b->ref_max_len[(size_t)rid] = std::max<int64_t>(b->ref_max_len[(size_t)rid], b->m_span[(size_t)i]);
        }
    }
}

// STREAMED open: block table, then the stream in windows of `win_blocks` BGZF blocks (inflated on all cores, walked,
// dropped); a record that straddles two windows is carried over.  Resident: one window + 22 bytes per record.
npore_bam *bam_open_streamed(const char *path, int threads, std::unique_ptr<PreadFile> file, const char *index_path)
{
    std::unique_ptr<npore_bam> hold(new npore_bam());
    npore_bam *b = hold.get();
    b->streamed = true;
    b->file = std::move(file);
    uint64_t total = 0;
    if (!bgzf_scan(*b->file, b->blocks, total)) { fail(NPORE_E_INVALID, std::string("'") + path + "' is not a BGZF file"); return nullptr; }
    b->data_size = (size_t)total;
    if (index_path && *index_path) {        // another process of this node has made the record index already
        MappedFile ix;
        const uint8_t *q = nullptr;
        if (ix.open(index_path) && ix.n >= 32 && std::memcmp(ix.p, "NPOREIX1", 8) == 0) q = ix.p;
        if (q) {
            uint64_t n_rec, hdr_len, tot;
            std::memcpy(&n_rec, q + 8, 8); std::memcpy(&hdr_len, q + 16, 8); std::memcpy(&tot, q + 24, 8);
            size_t hdr_end = 0;
            const size_t need = 32 + hdr_len + n_rec * 22;
            if (tot == total && ix.n >= need && bam_parse_header(b, q + 32, (size_t)hdr_len, &hdr_end) == 1) {
                const uint8_t *a = q + 32 + hdr_len;
                b->rec_off.resize(n_rec); b->m_ref.resize(n_rec); b->m_pos.resize(n_rec); b->m_span.resize(n_rec); b->m_flag.resize(n_rec);
                std::memcpy(b->rec_off.data(), a, n_rec * 8); a += n_rec * 8;
                std::memcpy(b->m_ref.data(), a, n_rec * 4); a += n_rec * 4;
                std::memcpy(b->m_pos.data(), a, n_rec * 4); a += n_rec * 4;
                std::memcpy(b->m_span.data(), a, n_rec * 4); a += n_rec * 4;
                std::memcpy(b->m_flag.data(), a, n_rec * 2);
                bam_finish_index(b);
                return hold.release();
            }
        }
        // (an unusable index file: fall through and index the file here)
    }
    size_t win_blocks = 4096;               // <= 256 MB of inflated stream per window
    if (const char *e = std::getenv("NPORE_BAM_WINDOW_BLOCKS")) win_blocks = (size_t)std::max(1, std::atoi(e));
    RawBuf win;
    std::vector<uint8_t> carry;             // the incomplete tail of the previous window
    uint64_t carry_at = 0;                  // stream offset of carry[0]
    bool have_header = false;
    std::vector<int64_t> offs;
    for (size_t b0 = 0; b0 < b->blocks.size();) {
        const size_t b1 = std::min(b->blocks.size(), b0 + win_blocks);
        const uint64_t w0 = b->blocks[b0].out_off, w1 = b->blocks[b1 - 1].out_off + b->blocks[b1 - 1].out_len;
        if (!win.ensure(carry.size() + (size_t)(w1 - w0) + 8)) { fail(NPORE_E_NOMEM, "BAM window"); return nullptr; }
        uint8_t *d = reinterpret_cast<uint8_t *>(win.p);
        if (!carry.empty()) std::memcpy(d, carry.data(), carry.size());
        if (!bgzf_inflate_range(*b->file, b->blocks, b0, b1, d + carry.size(), threads)) {
            fail(NPORE_E_INVALID, std::string("'") + path + "': corrupt BGZF block");
            return nullptr;
        }
        const uint64_t base = carry.empty() ? w0 : carry_at;      // stream offset of d[0]
        const size_t N = carry.size() + (size_t)(w1 - w0);
        size_t p = 0;
        if (!have_header) {
            size_t hdr_end = 0;
            const int rc = bam_parse_header(b, d, N, &hdr_end);
            if (rc == 0) { fail(NPORE_E_INVALID, std::string("'") + path + "' is not a BAM file"); return nullptr; }
            if (rc < 0) {                                         // the header does not end in this window: read on
                if (b1 == b->blocks.size()) { fail(NPORE_E_INVALID, "truncated BAM header"); return nullptr; }
                carry.assign(d, d + N);
                carry_at = base;
                b0 = b1;
                continue;
            }
            have_header = true;
            p = hdr_end;
        }
        offs.clear();
        while (p + 4 <= N) {
            const int32_t bs = rdi32(&d[p]);
            if (bs < 32) { fail(NPORE_E_INVALID, "truncated BAM record"); return nullptr; }
            if (p + 4 + (size_t)bs > N) break;
            offs.push_back((int64_t)p);
            p += 4 + (size_t)bs;
        }
        if (!bam_index_records(b, d, offs, (int64_t)base, threads)) { fail(NPORE_E_INVALID, "corrupt BAM record"); return nullptr; }
        carry.assign(d + p, d + N);
        carry_at = base + p;
        b0 = b1;
    }
    if (!have_header) { fail(NPORE_E_INVALID, std::string("'") + path + "' is not a BAM file"); return nullptr; }
    if (!carry.empty()) { fail(NPORE_E_INVALID, "truncated BAM record"); return nullptr; }
    bam_finish_index(b);
    return hold.release();
}
}  // namespace

// ONE-PASS open (mode 3): the BGZF block table and the BAM header, nothing else -- no record is looked at until
// npore_bam_realign_sequential walks the stream.  Such a handle has no record index: npore_bam_select finds nothing.
npore_bam *bam_open_header_only(const char *path, int threads, std::unique_ptr<PreadFile> file)
{
    std::unique_ptr<npore_bam> hold(new npore_bam());
    npore_bam *b = hold.get();
    b->streamed = true;
    b->file = std::move(file);
    uint64_t total = 0;
    if (!bgzf_scan(*b->file, b->blocks, total)) { fail(NPORE_E_INVALID, std::string("'") + path + "' is not a BGZF file"); return nullptr; }
    b->data_size = (size_t)total;
    RawBuf head;
    for (size_t b1 = std::min<size_t>(b->blocks.size(), 16);; b1 = std::min(b->blocks.size(), b1 * 4)) {
        if (b1 == 0) { fail(NPORE_E_INVALID, std::string("'") + path + "' is not a BAM file"); return nullptr; }
        const size_t n = (size_t)(b->blocks[b1 - 1].out_off + b->blocks[b1 - 1].out_len);
        if (!head.ensure(n + 8) || !bgzf_inflate_range(*b->file, b->blocks, 0, b1, reinterpret_cast<uint8_t *>(head.p), threads)) {
            fail(NPORE_E_INVALID, std::string("'") + path + "': corrupt BGZF block");
            return nullptr;
        }
        size_t hdr_end = 0;
        const int rc = bam_parse_header(b, reinterpret_cast<const uint8_t *>(head.p), n, &hdr_end);
        if (rc == 1) break;
        if (rc == 0 || b1 == b->blocks.size()) { fail(NPORE_E_INVALID, std::string("'") + path + "' is not a BAM file"); return nullptr; }
    }
    bam_finish_index(b);                                 // (empty per-reference lists)
    // which contigs have reads (get_bam_regions' default keeps only those, src/util.py:16-93 with bam.count() > 0): from
    // the file's .bai when there is one (a reference with bins or linear-index entries has records) -- otherwise unknown
    // without a pass over the records: every contig is assumed to have some
    b->ref_has_reads.assign(b->ref_names.size(), 1);
    {
        const std::string p0 = std::string(path) + ".bai";
        std::string p1 = path;
        const size_t dot = p1.rfind('.');
        if (dot != std::string::npos) p1 = p1.substr(0, dot) + ".bai";
        std::vector<uint64_t> offs;
        std::vector<uint8_t> has;
        for (const std::string &cand : {p0, p1})
            if (bai_linear_offsets(cand.c_str(), offs, &has) && has.size() == b->ref_names.size()) { b->ref_has_reads = has; break; }
    }
    return hold.release();
}

// mode 0: automatic (streamed when the file is BGZF and larger than NPORE_BAM_STREAM_MB, default 1024 MB), 1: whole file
// resident, 2: streamed, 3: one-pass (header only; the reads through npore_bam_realign_sequential).  index_path (may be NULL): a record index saved by npore_bam_save_index for this very file --
// a streamed handle then skips its indexing pass (one process of a node indexes, the others load).
npore_bam *npore_bam_open_mode(const char *path, int threads, int mode, const char *index_path)
try {
    if (!path) { fail(NPORE_E_INVALID, "null path"); return nullptr; }
    if (mode != 1) {
        std::unique_ptr<PreadFile> pf(new PreadFile());
        if (!pf->open(path)) { fail(NPORE_E_INVALID, std::string("BAM file '") + path + "' not found"); return nullptr; }
        uint8_t magic[4] = {0, 0, 0, 0};
        const bool gz = pf->size >= 28 && pf->read(0, magic, 4) && magic[0] == 31 && magic[1] == 139;
        uint64_t limit_mb = 1024;
        if (const char *e = std::getenv("NPORE_BAM_STREAM_MB")) limit_mb = (uint64_t)std::max(0ll, std::atoll(e));
        if (gz && mode == 3) return bam_open_header_only(path, threads, std::move(pf));
        if (mode == 3) { fail(NPORE_E_INVALID, std::string("'") + path + "' is not a BGZF file (one-pass mode)"); return nullptr; }
        if (gz && (mode == 2 || pf->size > limit_mb * 1048576ull)) return bam_open_streamed(path, threads, std::move(pf), index_path);
        if (mode == 2) { fail(NPORE_E_INVALID, std::string("'") + path + "' is not a BGZF file (streamed mode)"); return nullptr; }
    }
    std::unique_ptr<MappedFile> mfp(new MappedFile());
    MappedFile &mf = *mfp;
    if (!mf.open(path)) { fail(NPORE_E_INVALID, std::string("BAM file '") + path + "' not found"); return nullptr; }
    const ByteSpan raw{mf.p, mf.n};
    std::unique_ptr<npore_bam> hold(new npore_bam());
    npore_bam *b = hold.get();
    std::string err;
    if (mf.n >= 12 && std::memcmp(mf.p, "BAM\1", 4) == 0) {
        // an inflated BAM stream (npore_bam_dump_inflated: one rank of a node inflates, the others map its copy)
        b->data = mf.p;
        b->data_size = mf.n;
        b->raw_map = std::move(mfp);
    } else {
        if (!bgzf_inflate(raw, threads, b->data_buf, b->data_size, err) || b->data_size < 12 || std::memcmp(b->data_buf.p, "BAM\1", 4) != 0) {
            fail(NPORE_E_INVALID, std::string("'") + path + "' is not a BAM file" + (err.empty() ? "" : " (" + err + ")"));
            return nullptr;
        }
        b->data = reinterpret_cast<const uint8_t *>(b->data_buf.p);
    }
    const uint8_t *d = b->data;
    const size_t N = b->data_size;
    size_t p = 0;
    if (bam_parse_header(b, d, N, &p) != 1) { fail(NPORE_E_INVALID, "truncated BAM header"); return nullptr; }
    // records: offsets (one hop per record), then validation + metadata on all cores, then the per-reference lists
    std::vector<int64_t> offs;
    while (p + 4 <= N) {
        const int32_t bs = rdi32(&d[p]);
        if (bs < 32 || p + 4 + (size_t)bs > N) { fail(NPORE_E_INVALID, "truncated BAM record"); return nullptr; }
        offs.push_back((int64_t)p);
        p += 4 + (size_t)bs;
    }
    if (!bam_index_records(b, d, offs, 0, threads)) { fail(NPORE_E_INVALID, "corrupt BAM record"); return nullptr; }
    bam_finish_index(b);
    return hold.release();
}
NPORE_CATCH_PTR
npore_bam *npore_bam_open(const char *path, int threads) { return npore_bam_open_mode(path, threads, 0, nullptr); }
int npore_bam_is_streamed(const npore_bam *b) { return b && b->streamed ? 1 : 0; }

// The record index of a handle (header + per-record offsets and metadata: 22 bytes per record), complete or not there
// at all, for npore_bam_open_mode(..., index_path) in the other processes of a node.
int npore_bam_save_index(const npore_bam *b, const char *path)
try {
    if (!b || !path) return fail(NPORE_E_INVALID, "null argument");
    // the header as a BAM stream prefix, so that the loader parses it with the same code
    std::string hdr("BAM\1", 4);
    auto put32 = [&](int32_t v) { hdr.append(reinterpret_cast<const char *>(&v), 4); };
    put32((int32_t)b->text.size());
    hdr += b->text;
    put32((int32_t)b->ref_names.size());
    for (size_t k = 0; k < b->ref_names.size(); k++) {
        put32((int32_t)b->ref_names[k].size() + 1);
        hdr.append(b->ref_names[k].c_str(), b->ref_names[k].size() + 1);
        put32((int32_t)b->ref_lens[k]);
    }
    const std::string tmp = std::string(path) + ".tmp" + std::to_string((long long)::getpid());
    FILE *fh = std::fopen(tmp.c_str(), "wb");
    if (!fh) return fail(NPORE_E_INVALID, "cannot create '" + tmp + "'");
    const uint64_t n_rec = b->rec_off.size(), hdr_len = hdr.size(), tot = b->data_size;
    bool ok = std::fwrite("NPOREIX1", 1, 8, fh) == 8 && std::fwrite(&n_rec, 8, 1, fh) == 1 && std::fwrite(&hdr_len, 8, 1, fh) == 1 &&
              std::fwrite(&tot, 8, 1, fh) == 1 && std::fwrite(hdr.data(), 1, hdr.size(), fh) == hdr.size();
    auto put = [&](const void *p, size_t bytes) { if (ok && bytes) ok = std::fwrite(p, 1, bytes, fh) == bytes; };
    put(b->rec_off.data(), n_rec * 8); put(b->m_ref.data(), n_rec * 4); put(b->m_pos.data(), n_rec * 4);
    put(b->m_span.data(), n_rec * 4); put(b->m_flag.data(), n_rec * 2);
    if (std::fclose(fh) != 0 || !ok || std::rename(tmp.c_str(), path) != 0) {
        std::remove(tmp.c_str());
        return fail(NPORE_E_INVALID, std::string("cannot write '") + path + "'");
    }
    return NPORE_OK;
}
NPORE_CATCH_INT
void npore_bam_close(npore_bam *b) { delete b; }

int64_t npore_bam_inflated_size(const npore_bam *b) { return b ? (int64_t)b->data_size : 0; }

int npore_bam_dump_inflated(const npore_bam *b, const char *path)
try {
    if (!b || !path) return fail(NPORE_E_INVALID, "null argument");
    if (b->streamed) return fail(NPORE_E_UNSUPPORTED, "a streamed BAM handle holds no inflated stream (share its index: npore_bam_save_index)");
    const std::string tmp = std::string(path) + ".tmp" + std::to_string((long long)::getpid());
    FILE *fh = std::fopen(tmp.c_str(), "wb");
    if (!fh) return fail(NPORE_E_INVALID, "cannot create '" + tmp + "'");
    const bool ok = std::fwrite(b->data, 1, b->data_size, fh) == b->data_size;
    if (std::fclose(fh) != 0 || !ok || std::rename(tmp.c_str(), path) != 0) {       // complete, or not there at all
        std::remove(tmp.c_str());
        return fail(NPORE_E_INVALID, std::string("cannot write '") + path + "'");
    }
    return NPORE_OK;
}
NPORE_CATCH_INT
int64_t npore_bam_n_records(const npore_bam *b) { return b ? (int64_t)b->rec_off.size() : 0; }
int npore_bam_n_refs(const npore_bam *b) { return b ? (int)b->ref_names.size() : 0; }
const char *npore_bam_ref_name(const npore_bam *b, int i) { return (b && i >= 0 && i < (int)b->ref_names.size()) ? b->ref_names[(size_t)i].c_str() : ""; }
int64_t npore_bam_ref_len(const npore_bam *b, int i) { return (b && i >= 0 && i < (int)b->ref_lens.size()) ? b->ref_lens[(size_t)i] : -1; }
int npore_bam_ref_has_reads(const npore_bam *b, int i) { return (b && i >= 0 && i < (int)b->ref_has_reads.size()) ? b->ref_has_reads[(size_t)i] : 0; }

int64_t npore_bam_select(const npore_bam *b, int n_regions, const int32_t *ref_id, const int64_t *start, const int64_t *stop,
                         int64_t max_reads, int64_t *out_idx, int64_t cap)
{
    if (!b || (n_regions > 0 && (!ref_id || !start || !stop)) || (cap > 0 && !out_idx)) return fail(NPORE_E_INVALID, "null argument");
    int64_t kept = 0;
    for (int g = 0; g < n_regions; g++) {
        if (ref_id[g] < 0 || ref_id[g] >= (int32_t)b->by_ref.size()) continue;
        const std::vector<int64_t> &recs = b->by_ref[(size_t)ref_id[g]];     // file order
        size_t first = 0;
        if (b->ref_sorted[(size_t)ref_id[g]]) {
            // coordinate-sorted (the usual case): skip everything that ends before the region can start
            const int64_t lo = start[g] - b->ref_max_len[(size_t)ref_id[g]];
            first = (size_t)(std::lower_bound(recs.begin(), recs.end(), lo,
                                              [&](int64_t i, int64_t v) { return (int64_t)b->m_pos[(size_t)i] < v; }) - recs.begin());
        }
        for (size_t q = first; q < recs.size(); q++) {
            const int64_t i = recs[q];
            const int64_t pos = b->m_pos[(size_t)i], rl = b->m_span[(size_t)i];
            if (b->ref_sorted[(size_t)ref_id[g]] && pos >= stop[g]) break;
            if (!(pos < stop[g] && pos + rl > start[g])) continue;                 // overlaps [start, stop)
            if (max_reads > 0 && kept >= max_reads) return kept;                   // src/bam.pyx:29-30
            if (b->m_flag[(size_t)i] & (0x100 | 0x800 | 0x4)) continue;           // secondary / supplementary / unmapped, :31-32
            if (kept < cap) out_idx[kept] = i;
            kept++;
        }
    }
    return kept;
}

npore_fasta *npore_fasta_open(const char *path)
try {
    if (!path) { fail(NPORE_E_INVALID, "null path"); return nullptr; }
    MappedFile mf;
    if (!mf.open(path)) { fail(NPORE_E_INVALID, std::string("could not open FASTA '") + path + "'"); return nullptr; }
    std::unique_ptr<npore_fasta> hold(new npore_fasta());
    static std::atomic<uint64_t> next_serial{1};
    hold->serial = next_serial.fetch_add(1);
    npore_fasta *f = hold.get();
    if (!fasta_parse(ByteSpan{mf.p, mf.n}, 0, *f)) { fail(NPORE_E_NOMEM, "FASTA: out of memory"); return nullptr; }
    return hold.release();
}
NPORE_CATCH_PTR
void npore_fasta_close(npore_fasta *f) { delete f; }
int npore_fasta_n(const npore_fasta *f) { return f ? (int)f->names.size() : 0; }
const char *npore_fasta_name(const npore_fasta *f, int i) { return (f && i >= 0 && i < (int)f->names.size()) ? f->names[(size_t)i].c_str() : ""; }
const char *npore_fasta_seq(const npore_fasta *f, int i) { return (f && i >= 0 && i < (int)f->names.size()) ? f->seq((size_t)i) : nullptr; }
int64_t npore_fasta_len(const npore_fasta *f, int i) { return (f && i >= 0 && i < (int)f->names.size()) ? f->len((size_t)i) : -1; }

namespace {
bool pack_args_ok(const npore_bam *b, const int64_t *idx, int64_t n)
{
    if (!b || n < 0 || (n > 0 && !idx)) return false;
    for (int64_t k = 0; k < n; k++)
        if (idx[k] < 0 || idx[k] >= (int64_t)b->rec_off.size()) return false;
    return true;
}
}  // namespace

namespace {
int fetch_records(const npore_bam *b, const int64_t *idx, int64_t n, int threads, RecFetch &rf)
{
    std::string err;
    if (!bam_fetch(*b, idx, n, threads, rf, err)) return fail(NPORE_E_INVALID, "BAM records: " + err);
    return NPORE_OK;
}
void pack_sizes_of(const RecFetch &rf, int64_t n, int64_t *ref_off, int64_t *seq_off, int64_t *cig_off, int threads = 0)
{
    // per read (a 10 kb read has thousands of CIGAR operations: all cores), then the prefix sums
    ref_off[0] = seq_off[0] = cig_off[0] = 0;
    const int64_t per = 64;
    parallel_for((n + per - 1) / per, threads, [&](int64_t t) {
        for (int64_t k = t * per; k < std::min(n, (t + 1) * per); k++) {
            const RecView r = rec_of(rf, k);
            int64_t lead, trail, ops = 0, rl = 0;
            rec_clips(r, lead, trail);
            const int nc = r.n_cigar();
            for (int c = 0; c < nc; c++) {
                const uint32_t w = r.cig(c), op = w & 15u, len = w >> 4;
                if (op != 4 && op != 5) ops += len;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += len;      // (rec_ref_len: M D N = X)
            }
            ref_off[k + 1] = rl;
            seq_off[k + 1] = std::max<int64_t>(0, (int64_t)r.l_seq() - lead - trail);
            cig_off[k + 1] = ops;
        }
    });
    for (int64_t k = 0; k < n; k++) { ref_off[k + 1] += ref_off[k]; seq_off[k + 1] += seq_off[k]; cig_off[k + 1] += cig_off[k]; }
}
}  // namespace

int npore_bam_pack_sizes(const npore_bam *b, const int64_t *idx, int64_t n, int64_t *ref_off, int64_t *seq_off, int64_t *cig_off)
try {
    if (!pack_args_ok(b, idx, n) || !ref_off || !seq_off || !cig_off) return fail(NPORE_E_INVALID, "bad argument");
    RecFetch rf;          // (local to the call: the handle is const here, and two threads may size / pack from one handle)
    if (int rc = fetch_records(b, idx, n, 0, rf)) return rc;
    pack_sizes_of(rf, n, ref_off, seq_off, cig_off, 0);
    return NPORE_OK;
}
NPORE_CATCH_INT

namespace {
int pack_records(const npore_bam *b, const RecFetch &rf, const npore_fasta *fa, const int32_t *fasta_of_ref, int64_t n,
                 uint8_t *refs, const int64_t *ref_off, uint8_t *seqs, const int64_t *seq_off, char *cigs,
                 const int64_t *cig_off, int threads, bool for_upload = false);
}
int npore_bam_pack(const npore_bam *b, const npore_fasta *fa, const int32_t *fasta_of_ref, const int64_t *idx, int64_t n,
                   uint8_t *refs, const int64_t *ref_off, uint8_t *seqs, const int64_t *seq_off, char *cigs,
                   const int64_t *cig_off, int threads)
try {
    if (!pack_args_ok(b, idx, n) || !fa || !fasta_of_ref || !ref_off || !seq_off || !cig_off ||
        (n > 0 && (!refs || !seqs || !cigs)))
        return fail(NPORE_E_INVALID, "bad argument");
    RecFetch rf;
    if (int rc = fetch_records(b, idx, n, threads, rf)) return rc;
    return pack_records(b, rf, fa, fasta_of_ref, n, refs, ref_off, seqs, seq_off, cigs, cig_off, threads);
}
NPORE_CATCH_INT

namespace {
int pack_records(const npore_bam *b, const RecFetch &rf, const npore_fasta *fa, const int32_t *fasta_of_ref, int64_t n,
                 uint8_t *refs, const int64_t *ref_off, uint8_t *seqs, const int64_t *seq_off, char *cigs,
                 const int64_t *cig_off, int threads, bool for_upload)
{
    std::atomic<int> bad{0};
    parallel_for(n, threads, [&](int64_t k) {
        const RecView r = rec_of(rf, k);
        const int32_t rid = r.ref_id();
        const int fi = (rid >= 0 && rid < (int32_t)b->ref_names.size()) ? fasta_of_ref[rid] : -1;
        if (fi < 0 || fi >= (int)fa->names.size()) { bad++; return; }
        // reference bases: FASTA slice [pos, pos + reference_length), what pysam rebuilds from MD (src/bam.pyx:45)
        const char *ctg = fa->seq((size_t)fi);
        const int64_t ctg_len = fa->len((size_t)fi);
        const int64_t rl = ref_off[k + 1] - ref_off[k], pos = r.pos();
        uint8_t *ro = refs + ref_off[k];
        {
            const int64_t q0 = std::min(rl, std::max<int64_t>(0, -pos)), q1 = std::max(q0, std::min(rl, ctg_len - pos));
            std::memset(ro, 0, (size_t)q0);
            base_codes(ctg + pos + q0, ro + q0, q1 - q0);
            std::memset(ro + q1, 0, (size_t)(rl - q1));
        }
        // query bases without the soft clips (src/bam.pyx:42)
        int64_t lead, trail;
        rec_clips(r, lead, trail);
        uint8_t *so = seqs + seq_off[k];
        const int64_t sl = seq_off[k + 1] - seq_off[k];
        nibble_codes(r.seq(), lead, so, sl);
        // expanded CIGAR without S and H (src/bam.pyx:59)
        char *co = cigs + cig_off[k];
        const int nc = r.n_cigar();
        for (int c = 0; c < nc; c++) {
            const uint32_t w = r.cig(c), op = w & 15u, len = w >> 4;
            if (op == 4 || op == 5) continue;
            const char ch = op < 10 ? CIGOPS[op] : '?';
            if (len <= 8) { for (uint32_t q = 0; q < len; q++) co[q] = ch; }      // (most runs are a few ops long)
            else std::memset(co, ch, len);
            co += len;
        }
        if (for_upload) {       // page-locked staging about to cross PCIe: out of this core's cache first (hostio.hpp)
            cache_writeback(ro, (size_t)rl);
            cache_writeback(so, (size_t)sl);
            cache_writeback(cigs + cig_off[k], (size_t)(cig_off[k + 1] - cig_off[k]));
        }
    });
    return bad ? fail(NPORE_E_INVALID, "a selected read lies on a contig that is not in the FASTA") : NPORE_OK;
}

#if defined(__x86_64__)
// "=ACMGRSVTWYHKDBN"[nibble] for 16 packed bytes at a time (high nibble first); returns the packed bytes done (a multiple of 16)
__attribute__((target("ssse3"))) static int64_t nibbles_to_text_ssse3(const uint8_t *src, char *dst, int64_t n_bytes)
{
    const __m128i lut = _mm_loadu_si128(reinterpret_cast<const __m128i *>(SEQ16));
    const __m128i low = _mm_set1_epi8(0x0F);
    int64_t j = 0;
    for (; j + 16 <= n_bytes; j += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + j));
        const __m128i hi = _mm_shuffle_epi8(lut, _mm_and_si128(_mm_srli_epi16(v, 4), low));
        const __m128i lo = _mm_shuffle_epi8(lut, _mm_and_si128(v, low));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + 2 * j), _mm_unpacklo_epi8(hi, lo));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + 2 * j + 16), _mm_unpackhi_epi8(hi, lo));
    }
    return j;
}
#endif

int format_sam_into(const npore_bam *b, const RecFetch &rf, int64_t n, const char *finals, const int64_t *final_off,
                    const int64_t *final_len, const int32_t *status, int threads, RawBuf &out, int64_t *sam_len)
{
    if (!b || n < 0 || !sam_len || (n > 0 && (!finals || !final_off || !final_len || !status)))
        return fail(NPORE_E_INVALID, "bad argument");
    // pass 1: line sizes; pass 2: fill (both parallel over reads)
    std::vector<int64_t> off((size_t)n + 1, 0);
    // decimal text of v at dst (dst == nullptr: only the length), no terminator
    auto put_int = [](char *dst, long long v) -> int {
        char tmp[24];
        int nd = 0;
        unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        do { tmp[nd++] = (char)('0' + u % 10); u /= 10; } while (u);
        const int neg = v < 0;
        if (dst) {
            if (neg) *dst++ = '-';
            for (int q = 0; q < nd; q++) dst[q] = tmp[nd - 1 - q];
        }
        return nd + neg;
    };
#if defined(__x86_64__)
    static const bool have_ssse3 = __builtin_cpu_supports("ssse3");
#endif
    static const struct PairTab {      // two bases of the 4-bit packed sequence per lookup
        uint16_t pair[256];
        PairTab() { for (int v = 0; v < 256; v++) pair[v] = (uint16_t)((uint8_t)SEQ16[v >> 4] | ((uint8_t)SEQ16[v & 15] << 8)); }
    } seqtab;
    auto line = [&](int64_t k, char *dst) -> int64_t {    // returns the length; writes when dst != nullptr
        if (status[k] & NPORE_ST_BAD_INPUT) return 0;    // refused reads are not written
        const RecView r = rec_of(rf, k);
        int64_t lead, trail;
        rec_clips(r, lead, trail);
        const int64_t sl = std::max<int64_t>(0, (int64_t)r.l_seq() - lead - trail);
        const int32_t rid = r.ref_id();
        const std::string &rn = (rid >= 0 && rid < (int32_t)b->ref_names.size()) ? b->ref_names[(size_t)rid] : std::string("*");
        const bool noq = r.l_seq() == 0 || r.qual()[0] == 0xFF;
        const size_t nl = std::strlen(r.name());
        const long long flag = r.flag(), pos1 = (long long)r.pos() + 1, mapq = r.mapq(), hp = (long long)rec_hp(r),
                        reflen = (long long)rec_ref_len(r);
        if (!dst)       // name \t flag \t rname \t pos \t mapq \t cigar \t * \t 0 \t tlen \t seq \t qual \t HP:i:n \n
            return (int64_t)nl + 1 + put_int(nullptr, flag) + 1 + (int64_t)rn.size() + 1 + put_int(nullptr, pos1) + 1 +
                   put_int(nullptr, mapq) + 1 + final_len[k] + 5 + put_int(nullptr, reflen) + 1 + sl + 1 + (noq ? 1 : sl) + 6 +
                   put_int(nullptr, hp) + 1;
        char *o = dst;
        std::memcpy(o, r.name(), nl); o += nl;
        *o++ = '\t'; o += put_int(o, flag); *o++ = '\t';
        std::memcpy(o, rn.data(), rn.size()); o += rn.size();
        *o++ = '\t'; o += put_int(o, pos1); *o++ = '\t'; o += put_int(o, mapq); *o++ = '\t';
        std::memcpy(o, finals + final_off[k], (size_t)final_len[k]); o += final_len[k];
        std::memcpy(o, "\t*\t0\t", 5); o += 5;
        o += put_int(o, reflen); *o++ = '\t';
        {
            const uint8_t *sq = r.seq();
            int64_t q = 0, t = lead;
            if (q < sl && (t & 1)) { o[q++] = SEQ16[sq[t >> 1] & 15]; t++; }
            const uint8_t *src = sq + (t >> 1);
            const int64_t pairs = (sl - q) >> 1;
            char *po = o + q;
            int64_t j = 0;
#if defined(__x86_64__)
            if (have_ssse3) j = nibbles_to_text_ssse3(src, po, pairs);          // 16 packed bytes -> 32 letters per step
#endif
            for (; j < pairs; j++) { const uint16_t v = seqtab.pair[src[j]]; std::memcpy(po + 2 * j, &v, 2); }
            q += 2 * pairs;
            if (q < sl) { o[q] = SEQ16[src[pairs] >> 4]; q++; }
            o += sl;
        }
        *o++ = '\t';
        if (noq) *o++ = '*';
        else {
            const uint8_t *__restrict ql = r.qual() + lead;
            char *__restrict qo = o;
            for (int64_t q = 0; q < sl; q++) qo[q] = (char)(33 + ql[q]);
            o += sl;
        }
        std::memcpy(o, "\tHP:i:", 6); o += 6;
        o += put_int(o, hp); *o++ = '\n';
        return (int64_t)(o - dst);
    };
    parallel_for(n, threads, [&](int64_t k) { off[(size_t)k + 1] = line(k, nullptr); });
    for (int64_t k = 0; k < n; k++) off[(size_t)k + 1] += off[(size_t)k];
    if (!out.ensure((size_t)off[(size_t)n] + 1)) return fail(NPORE_E_NOMEM, "SAM text buffer");
    parallel_for(n, threads, [&](int64_t k) { if (off[(size_t)k + 1] > off[(size_t)k]) line(k, out.p + off[(size_t)k]); });
    *sam_len = off[(size_t)n];
    return NPORE_OK;
}
}  // namespace

int npore_bam_format_sam(npore_bam *b, const int64_t *idx, int64_t n, const char *finals, const int64_t *final_off,
                         const int64_t *final_len, const int32_t *status, int threads, const char **sam, int64_t *sam_len)
try {
    if (!pack_args_ok(b, idx, n) || !sam) return fail(NPORE_E_INVALID, "bad argument");
    if (int rc = fetch_records(b, idx, n, threads, b->api_fetch)) return rc;
    const int rc = format_sam_into(b, b->api_fetch, n, finals, final_off, final_len, status, threads, b->sam, sam_len);
    *sam = b->sam.p;
    return rc;
}
NPORE_CATCH_INT

namespace {
// pack the selected records into the slot (inputs of npore_align_batch) and size its output buffers
// (the records are in s.rf already)
int slot_pack_records(const npore_bam *b, const npore_fasta *fa, const int32_t *fasta_of_ref, int64_t n, int threads, npore_batch_slot &s,
                      bool device_glue = false, bool compact = false)
{
    for (auto *v : {&s.ro, &s.so, &s.co, &s.oo, &s.fo}) v->assign((size_t)n + 1, 0);
    s.olen.assign((size_t)n, 0);
    s.flen.assign((size_t)n, 0);
    pack_sizes_of(s.rf, n, s.ro.data(), s.so.data(), s.co.data(), threads);
    if (!s.refs.ensure((size_t)s.ro[(size_t)n] + 64) || !s.seqs.ensure((size_t)s.so[(size_t)n] + 64) || !s.cigs.ensure((size_t)s.co[(size_t)n] + 64))
        return fail(NPORE_E_NOMEM, "batch buffers");
    if (!fa || !fasta_of_ref) return fail(NPORE_E_INVALID, "bad argument");
    if (int rc = pack_records(b, s.rf, fa, fasta_of_ref, n, reinterpret_cast<uint8_t *>(s.refs.p), s.ro.data(),
                              reinterpret_cast<uint8_t *>(s.seqs.p), s.so.data(), s.cigs.p, s.co.data(), threads, true))
        return rc;
    for (int64_t k = 0; k < n; k++) {
        const int64_t cap = (s.ro[(size_t)k + 1] - s.ro[(size_t)k]) + (s.so[(size_t)k + 1] - s.so[(size_t)k]);
        // (device glue: the slot receives the collapsed CIGAR text instead of the op string -- 2 bytes per op + 16 always suffice)
        s.oo[(size_t)k + 1] = s.oo[(size_t)k] + (device_glue ? 2 * cap + 16 : cap);
        s.fo[(size_t)k + 1] = s.fo[(size_t)k] + 2 * cap + 16;
    }
    // (compact: the texts come back compacted -- file_pipeline --, no page-locked copy of the slots is needed)
    if ((!compact && !s.alns.ensure((size_t)s.oo[(size_t)n] + 64)) || (!device_glue && !s.finals.ensure((size_t)s.fo[(size_t)n] + 64)))
        return fail(NPORE_E_NOMEM, "batch buffers");
    return NPORE_OK;
}
// device pack: the sizes as above, and instead of the three arrays the HEADS of the records (block_size word, fixed fields,
// name, CIGAR words, 4-bit bases -- about half of a record; qualities and tags are not needed on the device) copied one
// after the other into the slot's page-locked buffer; unpack_kernels.hpp does the rest per group.  Device glue only
// (the host glue reads the base arrays).
int slot_pack_raw(const npore_bam *b, const int32_t *fasta_of_ref, int n_fasta, int64_t n, int threads, npore_batch_slot &s, bool compact = false)
{
    for (auto *v : {&s.ro, &s.so, &s.co, &s.oo, &s.fo}) v->assign((size_t)n + 1, 0);
    s.rawo.assign((size_t)n + 1, 0);
    s.olen.assign((size_t)n, 0);
    s.flen.assign((size_t)n, 0);
    pack_sizes_of(s.rf, n, s.ro.data(), s.so.data(), s.co.data(), threads);
    for (int64_t k = 0; k < n; k++) {
        const RecView r = rec_of(s.rf, k);
        const int32_t rid = r.ref_id();
        const int fi = (rid >= 0 && rid < (int32_t)b->ref_names.size()) ? fasta_of_ref[rid] : -1;
        if (fi < 0 || fi >= n_fasta) return fail(NPORE_E_INVALID, "a selected read lies on a contig that is not in the FASTA");
        s.rawo[(size_t)k + 1] = s.rawo[(size_t)k] + (int64_t)(r.qual() - r.p) + 4;
        const int64_t cap = (s.ro[(size_t)k + 1] - s.ro[(size_t)k]) + (s.so[(size_t)k + 1] - s.so[(size_t)k]);
        s.oo[(size_t)k + 1] = s.oo[(size_t)k] + 2 * cap + 16;
    }
    if (!s.raw.ensure((size_t)s.rawo[(size_t)n] + 64) || (!compact && !s.alns.ensure((size_t)s.oo[(size_t)n] + 64))) return fail(NPORE_E_NOMEM, "batch buffers");
    const int64_t per = 16;
    parallel_for((n + per - 1) / per, threads, [&](int64_t t) {
        for (int64_t k = t * per; k < std::min(n, (t + 1) * per); k++) {
            char *dst = s.raw.p + s.rawo[(size_t)k];
            const size_t len = (size_t)(s.rawo[(size_t)k + 1] - s.rawo[(size_t)k]);
            std::memcpy(dst, s.rf.ptr[(size_t)k], len);
            cache_writeback(dst, len);         // page-locked staging about to cross PCIe: out of this core's cache first (hostio.hpp)
        }
    });
    return NPORE_OK;
}
int slot_pack(const npore_bam *b, const npore_fasta *fa, const int32_t *fasta_of_ref, const int64_t *idx, int64_t n, int threads,
              npore_batch_slot &s)
{
    if (!pack_args_ok(b, idx, n)) return fail(NPORE_E_INVALID, "bad argument");
    if (int rc = fetch_records(b, idx, n, threads, s.rf)) return rc;
    return slot_pack_records(b, fa, fasta_of_ref, n, threads, s);
}
int slot_align(npore_ctx *ctx, int64_t n, float indel_start, float indel_extend, int max_b_rows, int r, int32_t *status,
               npore_batch_slot &s)
{
    return npore_align_batch(ctx, n, reinterpret_cast<uint8_t *>(s.refs.p), s.ro.data(), reinterpret_cast<uint8_t *>(s.seqs.p),
                             s.so.data(), s.cigs.p, s.co.data(), indel_start, indel_extend, max_b_rows, r, s.alns.p, s.oo.data(),
                             s.olen.data(), status);
}
// realign_read's glue (src/bam.pyx:65-78) and the SAM lines; reads refused by align() have no string and get an empty CIGAR
int slot_post(const npore_bam *b, const int64_t *idx, int64_t n, const int32_t *status, int threads, npore_batch_slot &s,
              double *ms_std = nullptr, bool device_glue = false)
{
    if (device_glue) {          // the slots hold the final CIGAR text already (standardize_kernel)
        if (ms_std) *ms_std = 0.0;
        for (int64_t k = 0; k < n; k++) s.flen[(size_t)k] = s.olen[(size_t)k] > 0 ? s.olen[(size_t)k] : 0;
        if (s.ctext_copied > 0) {          // ... compacted (file_pipeline): the front of the compact buffer is here, the rest is fetched now
            const int64_t *coff = reinterpret_cast<const int64_t *>(s.coff_pin.p);
            int64_t extent = 0;
            for (int64_t k = 0; k < n; k++) {
                if (s.flen[(size_t)k] > 0 && coff[k] < 0) return fail(NPORE_E_HIP, "internal: compact text buffer overflow");
                if (s.flen[(size_t)k] > 0) extent = std::max(extent, coff[k] + s.flen[(size_t)k]);
            }
            if (extent > s.ctext_copied) {        // (texts far longer than usual: 0.5 bytes per base were sent with the batch)
                if (!s.ctext_pin.ensure((size_t)extent + 64)) return fail(NPORE_E_NOMEM, "batch buffers");
                HIP_TRY(hipMemcpy(s.ctext_pin.p, s.d_ctext.p, (size_t)extent, hipMemcpyDeviceToHost));
                s.ctext_copied = extent;
            }
            return format_sam_into(b, s.rf, n, s.ctext_pin.p, coff, s.flen.data(), status, threads, s.sam, &s.sam_len);
        }
        return format_sam_into(b, s.rf, n, s.alns.p, s.oo.data(), s.flen.data(), status, threads, s.sam, &s.sam_len);
    }
    const auto t0 = std::chrono::steady_clock::now();
    const uint8_t *refs = reinterpret_cast<const uint8_t *>(s.refs.p), *seqs = reinterpret_cast<const uint8_t *>(s.seqs.p);
    parallel_for(n, threads, [&](int64_t k) {
        const int64_t l = s.olen[(size_t)k] > 0 ? s.olen[(size_t)k] : 0;
        s.flen[(size_t)k] = standardize_collapsed_into(s.alns.p + s.oo[(size_t)k], l, refs + s.ro[(size_t)k],     // 2 bytes per op + 16 always suffice
                                                       s.ro[(size_t)k + 1] - s.ro[(size_t)k], seqs + s.so[(size_t)k],
                                                       s.so[(size_t)k + 1] - s.so[(size_t)k], s.finals.p + s.fo[(size_t)k]);
    });
    if (ms_std) *ms_std = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    (void)idx;
    return format_sam_into(b, s.rf, n, s.finals.p, s.fo.data(), s.flen.data(), status, threads, s.sam, &s.sam_len);
}
}  // namespace


int npore_bam_realign_batch(npore_ctx *ctx, npore_bam *b, const npore_fasta *fa, const int32_t *fasta_of_ref, const int64_t *idx,
                            int64_t n, float indel_start, float indel_extend, int max_b_rows, int r, int threads,
                            const char **sam, int64_t *sam_len, int32_t *status)
try {
    if (!ctx || !b || !status || !sam || !sam_len) return fail(NPORE_E_INVALID, "null argument");
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->slots[0]) ctx->slots[0] = new npore_batch_slot();
    npore_batch_slot &s = *ctx->slots[0];
    auto t0 = clk::now();
    if (int rc = slot_pack(b, fa, fasta_of_ref, idx, n, threads, s)) return rc;
    b->stage_ms[0] = ms_since(t0);
    t0 = clk::now();
    if (int rc = slot_align(ctx, n, indel_start, indel_extend, max_b_rows, r, status, s)) return rc;
    b->stage_ms[1] = ms_since(t0);
    t0 = clk::now();
    double ms_std = 0.0;
    const int rc = slot_post(b, idx, n, status, threads, s, &ms_std);
    b->stage_ms[2] = ms_std;
    b->stage_ms[3] = ms_since(t0) - ms_std;
    *sam = s.sam.p;
    *sam_len = s.sam_len;
    return rc;
}
NPORE_CATCH_INT

}  // extern "C"

namespace {
// The BAM -> SAM pipeline over a stream of batches.  Stages: records + pack (helper threads, two batches ahead) | the
// device path of ONE context through its asynchronous entry point -- a batch is enqueued the moment it is packed and
// the call returns at once, so up to N_SETS batches are on the device, the upload / preparation of one and the
// traceback / download of another beside the fill kernel of a third (run_core) | standardise + SAM text + ordered write
// (helper threads, each behind the event recorded behind its batch's download).  A slot carries a batch through all
// stages; six slots cover two being packed, three on the device and one being written.
//   acquire(k, slot) -> the records of batch k in slot.rf: their number, 0 = no further batch, < 0 = failure (fail() called);
//     serial_acquire: called in batch order (the one-pass reader), else from the packing threads as they come;
//   on_status(k, m, status bits): in batch order, in front of the batch's text.
template <class Acquire, class OnStatus>
int file_pipeline(npore_ctx *ctx, npore_bam *b, const npore_fasta *fa, const int32_t *fasta_of_ref, float indel_start, float indel_extend,
                  int max_b_rows, int r, int threads, FILE *fh, bool serial_acquire, Acquire acquire, OnStatus on_status)
{
    for (auto &sp : ctx->slots)
        if (!sp) sp = new npore_batch_slot();
    const auto wall0 = std::chrono::steady_clock::now();
    for (auto &sp : ctx->slots) { std::fill(sp->t_ms, sp->t_ms + 6, 0.0); sp->rc = 0; sp->m = 0; }
    ctx->file_mark[0] = ctx->totals[0] + ctx->totals[1] + ctx->totals[2];
    ctx->file_mark[1] = ctx->totals[3] + ctx->totals[4];
    constexpr int S = npore_ctx::N_SLOTS;
    // Several batches are in the host stages at once (two being packed, up to two being written): each stage gets half
    // of the CPUs, so that the process does not run five times as many busy threads as it has CPUs -- under a cgroup quota
    // that ends in the whole process being throttled for the rest of the scheduling period
    const int half_threads = std::max(1, host_threads(threads) / 2);
    // (experiments: NPORE_PACK_THREADS / NPORE_POST_THREADS override the split)
    const int pack_threads = std::getenv("NPORE_PACK_THREADS") ? std::max(1, std::atoi(std::getenv("NPORE_PACK_THREADS"))) : half_threads;
    const int post_threads = std::getenv("NPORE_POST_THREADS") ? std::max(1, std::atoi(std::getenv("NPORE_POST_THREADS"))) : half_threads;
    const bool glue = ctx->device_glue != 0;      // realign_read's glue on the device: the slots receive the final CIGAR text
    // ... and align()'s inputs unpacked from the records on the device (the host glue needs the base arrays on the host)
    const bool dpack = glue && ctx->device_pack != 0 && !(std::getenv("NPORE_DEVICE_PACK") && std::atoi(std::getenv("NPORE_DEVICE_PACK")) == 0);
    if (dpack)
        if (int rc = device_fasta(ctx, b, fa, fasta_of_ref)) return rc;
    const int n_fasta = (int)fa->names.size();
    // NPORE_PIPE_TRACE=1: one line per batch and stage boundary on stderr (ms since the call began)
    const bool trace = std::getenv("NPORE_PIPE_TRACE") != nullptr;
    std::mutex trace_m;
    auto mark = [&](const char *what, int64_t k) {
        if (!trace) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        std::lock_guard<std::mutex> lk(trace_m);
        std::fprintf(stderr, "pipe %8.2f ms  batch %3lld  %s\n", ms, (long long)k, what);
    };
    std::vector<std::future<void>> packed, posted;
    // batches leave in input order: `written` counts the batches that are through (written, or given up on);
    // `acquired` the batches whose records have been taken from a serial source
    std::mutex gate_m;
    std::condition_variable gate_cv;
    int64_t written = 0, acquired = 0;
    auto wait_for = [&](int64_t &counter, int64_t upto) {
        std::unique_lock<std::mutex> lk(gate_m);
        gate_cv.wait(lk, [&] { return counter >= upto; });
    };
    auto bump = [&](int64_t &counter) {
        { std::lock_guard<std::mutex> lk(gate_m); counter++; }
        gate_cv.notify_all();
    };
    auto start_pack = [&](int64_t k) {
        packed.push_back(std::async(std::launch::async, [&, k] {
            npore_batch_slot &s = *ctx->slots[(size_t)(k % S)];
            if (k >= S) wait_for(written, k - S + 1);          // the slot's previous batch has been written
            if (serial_acquire) wait_for(acquired, k);
            const auto t0 = std::chrono::steady_clock::now();
            mark("acquire begins", k);
            s.keep.clear();
            s.rc = 0;
            s.m = 0;
            bool bumped = !serial_acquire;
            try {                                       // (no exception may leave a task unseen: the batch fails instead)
                const int64_t m = acquire(k, s);
                mark("acquired", k);
                if (serial_acquire) { bump(acquired); bumped = true; }
                s.m = m > 0 ? m : 0;
                if (m < 0) s.rc = (int)m;
                else if (m > 0) s.rc = dpack ? slot_pack_raw(b, fasta_of_ref, n_fasta, m, pack_threads, s, glue)
                                             : slot_pack_records(b, fa, fasta_of_ref, m, pack_threads, s, glue, glue);
                if (s.rc) s.err = npore_last_error();
            } catch (const std::exception &e) {
                s.rc = NPORE_E_NOMEM;
                s.err = std::string("batch preparation: ") + e.what();
                s.m = 0;
            }
            if (!bumped) bump(acquired);                // (the batches behind this one must not wait for ever)
            s.t_ms[0] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            mark("packed", k);
        }));
    };
    auto start_post = [&](int64_t k) {
        posted.push_back(std::async(std::launch::async, [&, k] {
            npore_batch_slot &t = *ctx->slots[(size_t)(k % S)];
            struct Through { decltype(wait_for) &w; decltype(bump) &m; int64_t &c; int64_t k; ~Through() { w(c, k); m(c); } } through{wait_for, bump, written, k};
            if (t.rc) return;
            (void)hipSetDevice(ctx->device);
            auto t0 = std::chrono::steady_clock::now();
            if (hipEventSynchronize(t.done) != hipSuccess) { t.rc = NPORE_E_HIP; t.err = "waiting for a batch failed"; return; }
            mark("device done", k);
            t.t_ms[1] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            const int64_t m = t.m;
            std::memcpy(t.olen.data(), t.olen_pin.p, (size_t)m * 8);
            const int32_t *st = reinterpret_cast<const int32_t *>(t.st_pin.p);
            t0 = std::chrono::steady_clock::now();
            double ms_std = 0.0;
            try {
                t.rc = slot_post(b, nullptr, m, st, post_threads, t, &ms_std, glue);
            } catch (const std::exception &e) {
                fail(NPORE_E_NOMEM, std::string("SAM text of a batch: ") + e.what());
                t.rc = NPORE_E_NOMEM;
            }
            const double ms_post = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            t.t_ms[2] += ms_std;
            t.t_ms[3] += ms_post - ms_std;
            if (t.rc) { t.err = npore_last_error(); return; }
            mark("text made", k);
            wait_for(written, k);                              // records in input order
            on_status(k, m, st);
            t0 = std::chrono::steady_clock::now();
            if (std::fwrite(t.sam.p, 1, (size_t)t.sam_len, fh) != (size_t)t.sam_len) { t.rc = NPORE_E_INVALID; t.err = "short write"; }
            t.t_ms[4] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            mark("written", k);
            t.keep.clear();
        }));
    };
    int rc = NPORE_OK;
    std::string err;
    start_pack(0);
    start_pack(1);
    for (int64_t k = 0; rc == NPORE_OK; k++) {
        npore_batch_slot &s = *ctx->slots[(size_t)(k % S)];
        packed[(size_t)k].wait();
        if (s.rc) { rc = s.rc; err = s.err; break; }
        if (s.m == 0) break;                                   // the source is exhausted
        start_pack(k + 2);
        const int64_t m = s.m;
        if (!s.olen_pin.ensure((size_t)m * 8 + 64) || !s.st_pin.ensure((size_t)m * 4 + 64)) { rc = NPORE_E_NOMEM; err = "batch buffers"; break; }
        if (!s.done && hipEventCreateWithFlags(&s.done, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) { rc = NPORE_E_HIP; err = "hipEventCreate"; break; }
        // (returns once the batch's groups are enqueued; waits only when all work sets of the context are still busy)
        mark("enqueue begins", k);
        // device glue: the texts come back compacted -- the front of the batch's compact buffer (0.5 bytes per base are
        // sent with the batch's last group; what usual reads need is a quarter of that) and the reads' offsets
        TextCompact cmp{};
        s.ctext_copied = 0;
        if (glue) {
            // (a text takes whole 16-byte granules of the compact buffer: at most 15 bytes more than its slot)
            const int64_t slots = s.oo[(size_t)m] + 16 * m, bound = std::min(slots, slots / 4 + 4096);
            if (s.d_ctext.ensure((size_t)slots + 64) || s.d_cursor.ensure(64) || !s.ctext_pin.ensure((size_t)bound + 64) ||
                !s.coff_pin.ensure((size_t)m * 8 + 64)) { rc = NPORE_E_NOMEM; err = "batch buffers"; break; }
            cmp = TextCompact{s.d_ctext.as<uint8_t>(), s.d_cursor.as<unsigned long long>(), slots,
                              reinterpret_cast<int64_t *>(s.coff_pin.p), s.ctext_pin.p, bound};
            s.ctext_copied = bound;
            ctx->pending_compact = &cmp;
        }
        char *const out_host = glue ? s.ctext_pin.p : s.alns.p;      // (with the compaction nothing is copied there)
        if (dpack)
            s.rc = align_batch_raw(ctx, m, reinterpret_cast<uint8_t *>(s.raw.p), s.rawo.data(), s.ro.data(), s.so.data(), s.co.data(),
                                   indel_start, indel_extend, max_b_rows, r, out_host, s.oo.data(),
                                   reinterpret_cast<int64_t *>(s.olen_pin.p), reinterpret_cast<int32_t *>(s.st_pin.p));
        else
            s.rc = align_batch_host(ctx, m, reinterpret_cast<uint8_t *>(s.refs.p), s.ro.data(), reinterpret_cast<uint8_t *>(s.seqs.p),
                                    s.so.data(), s.cigs.p, s.co.data(), indel_start, indel_extend, max_b_rows, r, out_host,
                                    s.oo.data(), reinterpret_cast<int64_t *>(s.olen_pin.p), reinterpret_cast<int32_t *>(s.st_pin.p), false, glue);
        if (s.rc) { rc = s.rc; err = npore_last_error(); s.err = err; break; }
        if (hipEventRecord(s.done, ctx->s_post) != hipSuccess) { rc = NPORE_E_HIP; err = "hipEventRecord"; s.rc = rc; s.err = err; break; }
        mark("enqueued", k);
        start_post(k);
    }
    for (auto &f : packed) if (f.valid()) f.wait();
    for (auto &f : posted) if (f.valid()) f.wait();
    {
        const int rcw = npore_ctx_wait(ctx);                   // stage clocks of every group; a failure found while a work set was recycled
        if (rcw && rc == NPORE_OK) { rc = rcw; err = npore_last_error(); }
    }
    for (auto &sp : ctx->slots) {
        if (rc == NPORE_OK && sp->rc) { rc = sp->rc; err = sp->err; }
        sp->keep.clear();
    }
    // stage clocks of this call: sums over the batches of the time each stage's thread spent (stages of
    // neighbouring batches overlap, so the sums exceed the wall time), the wall time, and the GPU's share
    std::fill(b->file_ms, b->file_ms + 8, 0.0);
    for (auto &sp : ctx->slots)
        for (int q = 0; q < 5; q++) b->file_ms[q] += sp->t_ms[q];
    b->file_ms[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    b->file_ms[6] = ctx->totals[0] + ctx->totals[1] + ctx->totals[2] - ctx->file_mark[0];     // stage clocks of the device path (prep + fill + traceback;
    b->file_ms[7] = ctx->totals[3] + ctx->totals[4] - ctx->file_mark[1];                     // H2D + D2H): SUMS over groups that run beside each other
    return rc == NPORE_OK ? NPORE_OK : fail(rc, err);
}

// the variable-length parts the record accessors will walk lie inside the record
bool record_is_sound(const uint8_t *q)
{
    const int32_t bs = rdi32(q);
    const uint8_t *f = q + 4;
    const int64_t l_rn = f[8], n_cig = rd16(f + 12), l_seq = rdi32(f + 16);
    return !(l_rn < 1 || l_seq < 0 || 32 + l_rn + 4 * n_cig + (l_seq + 1) / 2 + l_seq > bs || f[32 + l_rn - 1] != 0);
}
}  // namespace

extern "C" {

int npore_bam_set_share(npore_bam *b, int rank, int world, const char *bai_path)
try {
    if (!b || world < 1 || rank < 0 || rank >= world) return fail(NPORE_E_INVALID, "bad argument");
    if (!b->file || b->blocks.empty()) return fail(NPORE_E_INVALID, "a share needs a handle opened on a BGZF file (modes 2 and 3)");
    b->has_share = false;
    b->share_begin = 0;
    b->share_end = UINT64_MAX;
    b->share_block = 0;
    if (world == 1) return NPORE_OK;
    std::vector<uint64_t> cuts;
    if (!bai_path || !bai_linear_offsets(bai_path, cuts) || cuts.empty())
        return fail(NPORE_E_UNSUPPORTED, "no usable .bai linear index: the record stream cannot be dealt without a pass over it");
    // the first record of the stretch whose compressed offset lies at or behind k / world of the file
    const uint64_t c0 = b->blocks.front().in_off, c1 = b->blocks.back().in_off + b->blocks.back().in_len;
    auto cut_of = [&](int k) -> uint64_t {
        if (k <= 0) return 0;
        if (k >= world) return UINT64_MAX;
        const uint64_t target = c0 + (uint64_t)((long double)(c1 - c0) * k / world);
        auto it = std::lower_bound(cuts.begin(), cuts.end(), target << 16);
        return it == cuts.end() ? UINT64_MAX : *it;
    };
    // virtual offset -> block of the table and offset in the inflated stream (false: the index is not this file's)
    auto locate = [&](uint64_t v, size_t &blk, uint64_t &abs) -> bool {
        const uint64_t coff = v >> 16, uoff = v & 0xFFFFu;
        // a block's payload begins a gzip header's length behind the block: the block that starts at `coff` is the first
        // one whose payload offset lies behind it
        size_t lo = 0, hi = b->blocks.size();
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (b->blocks[mid].in_off > coff) hi = mid; else lo = mid + 1; }
        if (lo == b->blocks.size() || b->blocks[lo].in_off - coff > 4096 || uoff >= b->blocks[lo].out_len) return false;
        blk = lo;
        abs = b->blocks[lo].out_off + uoff;
        return true;
    };
    const uint64_t v0 = cut_of(rank), v1 = cut_of(rank + 1);
    if (v0 == UINT64_MAX) { b->share_begin = b->share_end = UINT64_MAX; }       // nothing left for this rank
    else if (v0 != 0) {
        if (!locate(v0, b->share_block, b->share_begin)) return fail(NPORE_E_UNSUPPORTED, "the .bai index does not belong to this BAM file");
    }
    if (v1 != UINT64_MAX && v0 != UINT64_MAX) {
        size_t blk;
        if (!locate(v1, blk, b->share_end)) return fail(NPORE_E_UNSUPPORTED, "the .bai index does not belong to this BAM file");
    }
    b->has_share = true;
    return NPORE_OK;
}
NPORE_CATCH_INT

int npore_bam_share_info(const npore_bam *b, int64_t *out4)
{
    if (!b || !out4) return fail(NPORE_E_INVALID, "null argument");
    out4[0] = b->has_share ? 1 : 0;
    out4[1] = b->share_begin == UINT64_MAX ? -1 : (int64_t)b->share_begin;
    out4[2] = b->share_end == UINT64_MAX ? -1 : (int64_t)b->share_end;
    out4[3] = (int64_t)b->share_block;
    return NPORE_OK;
}

int npore_bam_realign_file(npore_ctx *ctx, npore_bam *b, const npore_fasta *fa, const int32_t *fasta_of_ref, const int64_t *idx,
                           int64_t n, int64_t batch_reads, float indel_start, float indel_extend, int max_b_rows, int r,
                           int threads, const char *out_path, int32_t *status)
try {
    if (!ctx || !b || !fa || !fasta_of_ref || !out_path || (n > 0 && (!idx || !status)) || batch_reads < 1) return fail(NPORE_E_INVALID, "bad argument");
    if (!pack_args_ok(b, idx, n)) return fail(NPORE_E_INVALID, "bad argument");
    FILE *fh = std::fopen(out_path, "ab");
    if (!fh) return fail(NPORE_E_INVALID, std::string("cannot open '") + out_path + "' for appending");
    HIP_TRY(hipSetDevice(ctx->device));
    const int64_t nb = (n + batch_reads - 1) / batch_reads;
    int rc = file_pipeline(ctx, b, fa, fasta_of_ref, indel_start, indel_extend, max_b_rows, r, threads, fh, false,
                           [&](int64_t k, npore_batch_slot &s) -> int64_t {
                               if (k >= nb) return 0;
                               const int64_t m = std::min(batch_reads, n - k * batch_reads);
                               const int rcf = fetch_records(b, idx + k * batch_reads, m, threads, s.rf);
                               return rcf ? (int64_t)rcf : m;
                           },
                           [&](int64_t k, int64_t m, const int32_t *st) { std::memcpy(status + k * batch_reads, st, (size_t)m * 4); });
    if (std::fclose(fh) != 0 && rc == NPORE_OK) rc = fail(NPORE_E_INVALID, "close failed");
    return rc;
}
NPORE_CATCH_INT

int npore_bam_realign_sequential(npore_ctx *ctx, npore_bam *b, const npore_fasta *fa, const int32_t *fasta_of_ref, int n_regions,
                                 const int32_t *ref_id, const int64_t *start, const int64_t *stop, int64_t max_reads,
                                 int64_t batch_reads, float indel_start, float indel_extend, int max_b_rows, int r, int threads,
                                 const char *out_path, int64_t *counts, int64_t *bad_ord, int32_t *bad_status, int64_t bad_cap)
try {
    if (!ctx || !b || !fa || !fasta_of_ref || !out_path || !counts || batch_reads < 1 || n_regions < 0 ||
        (n_regions > 0 && (!ref_id || !start || !stop)) || (bad_cap > 0 && (!bad_ord || !bad_status)))
        return fail(NPORE_E_INVALID, "bad argument");
    if (!b->file || b->blocks.empty()) return fail(NPORE_E_INVALID, "one-pass ingest needs a handle opened on a BGZF file (modes 2 and 3)");
    for (int g = 0; g < n_regions; g++)
        if (ref_id[g] < 0 || ref_id[g] >= (int32_t)b->ref_names.size() || (g > 0 && ref_id[g] <= ref_id[g - 1]))
            return fail(NPORE_E_UNSUPPORTED, "one-pass ingest takes at most one region per contig, in the order of the BAM header");
    counts[0] = counts[1] = counts[2] = 0;
    FILE *fh = std::fopen(out_path, "ab");
    if (!fh) return fail(NPORE_E_INVALID, std::string("cannot open '") + out_path + "' for appending");
    HIP_TRY(hipSetDevice(ctx->device));
    // the reader: the stream in windows of consecutive BGZF blocks, each inflated ONCE on all cores; a record that
    // straddles two windows is carried to the front of the next one, so every record lies in one window buffer, which
    // lives as long as a batch points into it
    size_t win_blocks = 1024;                // <= 64 MB of inflated stream per window
    if (const char *e = std::getenv("NPORE_BAM_WINDOW_BLOCKS")) win_blocks = (size_t)std::max(1, std::atoi(e));
    std::shared_ptr<RawBuf> win;
    const uint8_t *wd = nullptr;             // the window's first byte (the carried tail sits in front of the inflated blocks)
    size_t next_block = 0, N = 0, p = 0;
    bool have_header = false, done = n_regions == 0;
    // several processes on one file (npore_bam_set_share): this one walks the records that START in [share_begin,
    // share_end) of the inflated stream -- a stretch that begins at a record (a virtual offset of the .bai linear index);
    // the header was read when the handle was opened
    uint64_t abs0 = 0;                       // offset of the window's first byte in the inflated stream
    bool seek_share = false;
    if (b->has_share) {
        if (max_reads > 0) { std::fclose(fh); return fail(NPORE_E_UNSUPPORTED, "one-pass ingest: max_reads needs one process (the ranks cannot know how many reads the others keep)"); }
        if (b->share_begin == UINT64_MAX) done = true;
        else if (b->share_begin > 0) { next_block = b->share_block; have_header = true; seek_share = true; }
    }
    int g = 0;
    int64_t kept = 0, n_bad = 0, ordinal0 = 0;
    int32_t last_rid = -1;
    // the next window is inflated (on all cores) while the records of the current one are walked and packed: into its
    // buffer behind HEAD bytes of room, where the carried tail of the current window is put once the walk has reached it
    constexpr size_t HEAD = 4u << 20;
    struct Pending { std::shared_ptr<RawBuf> buf; size_t bytes = 0, b0 = 0, b1 = 0; bool ok = false; };
    // The inflater: ONE thread that takes the windows in file order, each on all cores (bgzf_inflate_range), and keeps up to
    // `depth` of them ready -- the inflation then runs whenever CPUs are free instead of one window ahead of the walk (rounds
    // 4 - 5a: the acquisitions of consecutive batches took 12 ... 72 ms, the long ones waiting for a window that the SAM text
    // and packing threads had slowed down, with 12 of the lease's 16 CPUs busy on average).
    // (a process that walks only its stretch of the file stops taking windows where the stretch ends; the inflater is at most
    // `depth` windows further on then -- the last record of a stretch may run on into any number of blocks, so it is not
    // cut off by block count)
    const size_t pf_end = b->blocks.size();
    int pf_depth = 3;
    if (const char *e = std::getenv("NPORE_BAM_WINDOWS_AHEAD")) pf_depth = std::max(1, std::atoi(e));
    int inflate_threads = threads;                     // (experiments: how many of the lease's CPUs the inflater may take at once)
    if (const char *e = std::getenv("NPORE_INFLATE_THREADS")) inflate_threads = std::max(1, std::atoi(e));
    std::mutex pf_m;
    std::condition_variable pf_cv;
    std::deque<Pending> pf_ready;
    bool pf_stop = false, pf_done = false, pf_started = false;
    std::thread pf_thread;
    auto pf_start = [&](size_t first_block) {
        pf_started = true;
        pf_thread = std::thread([&, first_block] {
            size_t b0 = first_block;
            for (;;) {
                {
                    std::unique_lock<std::mutex> lk(pf_m);
                    pf_cv.wait(lk, [&] { return pf_stop || (int)pf_ready.size() < pf_depth; });
                    if (pf_stop || b0 >= pf_end) { pf_done = true; pf_cv.notify_all(); return; }
                }
                Pending pd;
                pd.b0 = b0;
                pd.b1 = std::min(pf_end, b0 + win_blocks);
                const uint64_t w0 = b->blocks[b0].out_off, w1 = b->blocks[pd.b1 - 1].out_off + b->blocks[pd.b1 - 1].out_len;
                pd.bytes = (size_t)(w1 - w0);
                try {                                       // (no exception may leave a thread: the walk sees ok == false)
                    pd.buf = std::make_shared<RawBuf>();
                    pd.ok = pd.buf->ensure(HEAD + pd.bytes + 8) &&
                            bgzf_inflate_range(*b->file, b->blocks, b0, pd.b1, reinterpret_cast<uint8_t *>(pd.buf->p) + HEAD, inflate_threads);
                } catch (...) {
                    pd.ok = false;
                }
                b0 = pd.b1;
                {
                    std::lock_guard<std::mutex> lk(pf_m);
                    pf_ready.push_back(std::move(pd));
                }
                pf_cv.notify_all();
            }
        });
    };
    auto pf_finish = [&] {
        if (!pf_started) return;
        { std::lock_guard<std::mutex> lk(pf_m); pf_stop = true; }
        pf_cv.notify_all();
        if (pf_thread.joinable()) pf_thread.join();
    };
    struct PfGuard { decltype(pf_finish) &f; ~PfGuard() { f(); } } pf_guard{pf_finish};
    // -1: failure (fail() called); 0: end of the stream; 1: a new window is in place
    auto load_window = [&]() -> int {
        const size_t c = (win && p < N) ? N - p : 0;             // carried tail of the current window
        if (next_block >= pf_end) {
            if (c || !have_header) { fail(NPORE_E_INVALID, have_header ? "truncated BAM record" : "not a BAM file"); return -1; }
            return 0;
        }
        if (!pf_started) pf_start(next_block);
        const uint64_t win_off = b->blocks[next_block].out_off;
        Pending pd;
        {
            std::unique_lock<std::mutex> lk(pf_m);
            pf_cv.wait(lk, [&] { return !pf_ready.empty() || pf_done; });
            if (pf_ready.empty()) { fail(NPORE_E_INVALID, "BAM window reader ended early"); return -1; }
            pd = std::move(pf_ready.front());
            pf_ready.pop_front();
        }
        pf_cv.notify_all();
        if (!pd.ok || pd.b0 != next_block) { fail(NPORE_E_INVALID, "corrupt BGZF block (or out of memory)"); return -1; }
        std::shared_ptr<RawBuf> nw = pd.buf;
        uint8_t *d0 = reinterpret_cast<uint8_t *>(nw->p) + HEAD;
        if (c > HEAD) {                                          // a record longer than the room in front: copy once
            auto big = std::make_shared<RawBuf>();
            if (!big->ensure(c + pd.bytes + 8)) { fail(NPORE_E_NOMEM, "BAM window"); return -1; }
            std::memcpy(big->p + c, d0, pd.bytes);
            nw = big;
            d0 = reinterpret_cast<uint8_t *>(nw->p) + c;
        }
        if (c) std::memcpy(d0 - c, wd + p, c);
        win = nw;
        wd = d0 - c;
        abs0 = win_off - c;
        N = c + pd.bytes;
        p = 0;
        if (seek_share) { p = (size_t)(b->share_begin - abs0); seek_share = false; }      // (the first window of a share that begins mid-file)
        next_block = pd.b1;
        return 1;
    };
    auto acquire = [&](int64_t, npore_batch_slot &s) -> int64_t {
        s.rf.ptr.clear();
        while ((int64_t)s.rf.ptr.size() < batch_reads && !done) {
            const uint8_t *d = wd;
            if (!have_header) {
                npore_bam scratch;
                size_t hdr_end = 0;
                const int hrc = win ? bam_parse_header(&scratch, d, N, &hdr_end) : -1;
                if (hrc == 0) return fail(NPORE_E_INVALID, "not a BAM file");
                if (hrc < 0) {                                   // the header does not end in what is inflated so far
                    p = 0;
                    const int lw = load_window();
                    if (lw < 0) return lw;
                    if (lw == 0) return fail(NPORE_E_INVALID, "truncated BAM header");
                    continue;
                }
                have_header = true;
                p = hdr_end;
                continue;
            }
            int32_t bs = 0;
            if (p + 4 > N || (bs = rdi32(d + p)) < 32 || p + 4 + (size_t)bs > N) {
                if (p + 4 <= N && bs < 32) return fail(NPORE_E_INVALID, "truncated BAM record");
                const int lw = load_window();
                if (lw < 0) return lw;
                if (lw == 0) done = true;
                continue;
            }
            if (abs0 + p >= b->share_end) { done = true; break; }                // the next process's stretch begins here
            const uint8_t *q = d + p;
            p += 4 + (size_t)bs;
            if (!record_is_sound(q)) return fail(NPORE_E_INVALID, "corrupt BAM record");
            const RecView rv = rec_view(q);
            const int32_t rid = rv.ref_id();
            if (rid < 0) continue;                               // unplaced
            if (rid < last_rid) return fail(NPORE_E_UNSUPPORTED, "the BAM is not sorted by reference: one-pass ingest needs a coordinate-sorted file");
            last_rid = rid;
            while (g < n_regions && ref_id[g] < rid) g++;
            if (g == n_regions) { done = true; break; }
            if (ref_id[g] != rid) continue;
            const int64_t pos = rv.pos(), rl = rec_ref_len(rv);
            if (!(pos < stop[g] && pos + rl > start[g])) continue;               // overlaps [start, stop)
            if (max_reads > 0 && kept >= max_reads) { done = true; break; }      // src/bam.pyx:29-30
            if (rv.flag() & (0x100 | 0x800 | 0x4)) continue;                     // secondary / supplementary / unmapped, :31-32
            s.rf.ptr.push_back(q);
            if (s.keep.empty() || s.keep.back() != win) s.keep.push_back(win);
            kept++;
        }
        return (int64_t)s.rf.ptr.size();
    };
    auto on_status = [&](int64_t, int64_t m, const int32_t *st) {
        for (int64_t i = 0; i < m; i++)
            if (st[i]) {
                counts[(st[i] & NPORE_ST_BAD_INPUT) ? 1 : 2]++;
                if (n_bad < bad_cap) { bad_ord[n_bad] = ordinal0 + i; bad_status[n_bad] = st[i]; }
                n_bad++;
            }
        ordinal0 += m;
    };
    int rc = file_pipeline(ctx, b, fa, fasta_of_ref, indel_start, indel_extend, max_b_rows, r, threads, fh, true, acquire, on_status);
    pf_finish();                                                 // (windows inflated ahead of a stream that ended early)
    counts[0] = ordinal0;
    if (std::fclose(fh) != 0 && rc == NPORE_OK) rc = fail(NPORE_E_INVALID, "close failed");
    return rc;
}
NPORE_CATCH_INT

int npore_bam_file_timing(const npore_bam *b, double *ms, int n)
{
    if (!b || !ms) return fail(NPORE_E_INVALID, "null argument");
    for (int k = 0; k < n; k++) ms[k] = k < 8 ? b->file_ms[k] : 0.0;
    return NPORE_OK;
}

int npore_bam_last_timing(const npore_bam *b, double *ms, int n)
{
    if (!b || !ms) return fail(NPORE_E_INVALID, "null argument");
    for (int k = 0; k < n; k++) ms[k] = k < 4 ? b->stage_ms[k] : 0.0;
    return NPORE_OK;
}

}  // extern "C"
