// npore_api.cpp -- C ABI (include/npore_amd.h) and host orchestration.
// Compiled with hipcc for gfx950 only.  There is no CPU execution path for the
// DP here: without a gfx950 device npore_ctx_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/npore_amd.h"
#include "kernels.hpp"
#include "prep.hpp"

using namespace npore;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
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

struct npore_ctx {
    int device = 0;
    int max_n = 6, max_l = 100;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    float *d_sub = nullptr, *d_np = nullptr;
    std::vector<float> h_sub, h_np;
    // tunables
    int64_t tb_budget_mb = 0;   // 0 = auto
    int force_ng = 0;
    int force_chunks = 0;
    int force_nw = 0;
    int host_threads = 0;
    // device buffers
    DevBuf refs, seqs, steps, inss, seqw, refw, refl, descs, sched, rfc, rstat, tb, cout_, clen, cstat;
    DevBuf out, out_off, out_len, status;   // used by the host-buffer entry point
    HostBuf h_stage;
    double timing[8] = {};
};

namespace {

// (waves per chunk) * (columns per lane) * 64 must cover the band; returns nw*16 + ng, 0 if impossible
int pick_shape(int r, int force_nw, int force_ng)
{
    const int W = 2 * r + 1;
    for (int cover : {1, 2, 4, 8}) {
        if (64 * cover < W) continue;
        int nw = cover, ng = 1;                 // default: one column per lane, more waves per chunk
        if (force_ng > 0 && cover % force_ng == 0) { ng = force_ng; nw = cover / ng; }
        if (force_nw > 0 && cover % force_nw == 0) { nw = force_nw; ng = cover / nw; }
        return nw * 16 + ng;
    }
    return 0;
}

int pow2_at_least(int x)
{
    int p = 64;
    while (p < x) p <<= 1;
    return p;
}

// NW waves per chunk, `chunks` chunks per workgroup (they share the LDS score table).
template <int NW, int NG, int MAXT>
hipError_t launch_fill(KParams kp, int n_chunks, int force_chunks, hipStream_t s)
{
    const int W = 2 * kp.r + 1;
    kp.lstr = (W + NG - 1) / NG;
    kp.rwin = pow2_at_least(2 * kp.r + 101);
    kp.n_chunks = n_chunks;
    const size_t lds_cap = 160 * 1024 / sizeof(float);
    if (fill_lds_floats(NW, NG, 1, kp.lstr, kp.rwin) > lds_cap) return hipErrorInvalidValue;
    int cmax = 1;
    while ((cmax + 1) * NW * 64 <= MAXT && fill_lds_floats(NW, NG, cmax + 1, kp.lstr, kp.rwin) <= lds_cap) cmax++;
    // few chunks: spread them over the CUs; many: pack workgroups so that the table is amortised
    int chunks = std::min(cmax, std::max(1, (n_chunks + 255) / 256));
    if (force_chunks > 0) chunks = std::min(cmax, force_chunks);
    const size_t lds = fill_lds_floats(NW, NG, chunks, kp.lstr, kp.rwin) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fill_kernel<NW, NG, MAXT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((fill_kernel<NW, NG, MAXT>), dim3((n_chunks + chunks - 1) / chunks), dim3(64 * NW * chunks),
                       lds, s, kp);
    return hipGetLastError();
}

struct ReadPlan {
    ReadPath path;
    bool ok = false;
    int n_chunks = 0;
};

struct OutTarget {
    uint8_t *d_out;
    const int64_t *d_out_off;
    int64_t *d_out_len;
    int32_t *d_status;
};

struct AlignArgs {
    int64_t n_reads;
    const uint8_t *refs;
    const int64_t *ref_off;
    const uint8_t *seqs;
    const int64_t *seq_off;
    const char *cigs;
    const int64_t *cig_off;
    float indel_start, indel_extend;
    int max_b_rows, r;
};

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class F>
void parallel_for(int64_t n, int threads, F f)
{
    if (threads <= 1 || n <= 1) {
        for (int64_t i = 0; i < n; i++) f(i);
        return;
    }
    std::atomic<int64_t> next{0};
    std::vector<std::thread> pool;
    const int nt = (int)std::min<int64_t>(threads, n);
    for (int t = 0; t < nt; t++)
        pool.emplace_back([&] {
            for (;;) {
                int64_t i = next.fetch_add(1);
                if (i >= n) break;
                f(i);
            }
        });
    for (auto &t : pool) t.join();
}

// Runs reads [g0,g1) (one group whose traceback fits the budget).
int run_group(npore_ctx *ctx, const AlignArgs &a, int64_t g0, int64_t g1, std::vector<ReadPlan> &plans,
              const OutTarget &ot, hipStream_t s, int ng)
{
    const int64_t nr = g1 - g0;
    const int r = a.r;
    const int tbs = tb_stride(r);
    const int threads = ctx->host_threads > 0 ? ctx->host_threads
                                              : (int)std::max(1u, std::thread::hardware_concurrency());
    const double t_prep0 = now_ms();

    // ---- sizes and offsets (serial, cheap)
    int64_t n_chunks = 0, steps_tot = 0, inss_tot = 0, seqw_tot = 0, refw_tot = 0, out_tot = 0, tb_tot = 0;
    int64_t bases_ref0 = a.ref_off[g0], bases_seq0 = a.seq_off[g0];
    std::vector<int32_t> rfc(nr + 1), rstat(nr);
    std::vector<ChunkDesc> descs;
    for (int64_t k = 0; k < nr; k++) {
        ReadPlan &pl = plans[g0 + k];
        rfc[k] = (int32_t)n_chunks;
        rstat[k] = pl.ok ? 0 : NPORE_ST_BAD_INPUT;
        if (!pl.ok) continue;
        const auto &pa = pl.path;
        const int64_t steps_off = steps_tot, inss_off = inss_tot;
        steps_tot += (int64_t)pa.steps.size();
        inss_tot += (int64_t)pa.inss.size();
        for (size_t c = 0; c + 1 < pa.breaks.size(); c++) {
            const int64_t brk = pa.breaks[c], nxt = pa.breaks[c + 1];
            ChunkDesc d;
            std::memset(&d, 0, sizeof d);
            d.read_id = (int32_t)k;
            d.brk = (int32_t)brk;
            d.nrows = (int32_t)(nxt - brk + 1);
            d.row0 = pa.inss[brk];
            d.col0 = (int32_t)(brk - pa.inss[brk]);
            d.drows = pa.inss[nxt] - d.row0;
            d.dcols = (int32_t)(nxt - pa.inss[nxt]) - d.col0;
            d.out_cap = d.drows + d.dcols;
            d.steps_off = steps_off;
            d.inss_off = inss_off;
            d.seqw_off = seqw_tot;
            d.refw_off = refw_tot;
            d.tb_off = tb_tot;
            d.out_off = out_tot;
            d.seq_off = a.seq_off[g0 + k] - bases_seq0;
            d.ref_off = a.ref_off[g0 + k] - bases_ref0;
            seqw_tot += d.drows + 1;
            refw_tot += d.dcols + 1;
            tb_tot += (int64_t)d.nrows * tbs;
            out_tot += d.out_cap;
            descs.push_back(d);
            n_chunks++;
        }
    }
    rfc[nr] = (int32_t)n_chunks;
    std::vector<int32_t> sched(n_chunks);
    std::iota(sched.begin(), sched.end(), 0);
    std::stable_sort(sched.begin(), sched.end(),
                     [&](int32_t x, int32_t y) { return descs[x].nrows > descs[y].nrows; });

    // ---- staging layout (one pinned block)
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_steps = 0;
    const size_t o_inss = al(o_steps + steps_tot + 256);
    const size_t o_seqw = al(o_inss + inss_tot * 4);
    const size_t o_refw = al(o_seqw + seqw_tot * 4);
    const size_t o_refl = al(o_refw + refw_tot * 8);
    const size_t o_desc = al(o_refl + refw_tot * 8);
    const size_t o_sched = al(o_desc + n_chunks * sizeof(ChunkDesc));
    const size_t o_rfc = al(o_sched + n_chunks * 4);
    const size_t o_rstat = al(o_rfc + (nr + 1) * 4);
    const size_t stage_bytes = al(o_rstat + nr * 4);
    if (int rc = ctx->h_stage.ensure(stage_bytes)) return rc;
    uint8_t *hs = ctx->h_stage.as<uint8_t>();
    std::memset(hs + o_steps + steps_tot, 0, 256);
    std::memcpy(hs + o_desc, descs.data(), n_chunks * sizeof(ChunkDesc));
    std::memcpy(hs + o_sched, sched.data(), n_chunks * 4);
    std::memcpy(hs + o_rfc, rfc.data(), (nr + 1) * 4);
    std::memcpy(hs + o_rstat, rstat.data(), nr * 4);

    // ---- per-read packing (parallel)
    const int max_n = ctx->max_n, max_l = ctx->max_l;
    parallel_for(nr, threads, [&](int64_t k) {
        ReadPlan &pl = plans[g0 + k];
        if (!pl.ok) return;
        const auto &pa = pl.path;
        const int c0 = rfc[k], c1 = rfc[k + 1];
        if (c1 == c0) return;
        const ChunkDesc &d0 = descs[c0];
        std::memcpy(hs + o_steps + d0.steps_off, pa.steps.data(), pa.steps.size());
        std::memcpy(hs + o_inss + d0.inss_off * 4, pa.inss.data(), pa.inss.size() * 4);
        const uint8_t *seq = a.seqs + a.seq_off[g0 + k];
        const uint8_t *ref = a.refs + a.ref_off[g0 + k];
        const int64_t S = a.seq_off[g0 + k + 1] - a.seq_off[g0 + k];
        const int64_t R = a.ref_off[g0 + k + 1] - a.ref_off[g0 + k];
        std::vector<int32_t> scratch;
        for (int c = c0; c < c1; c++) {
            const ChunkDesc &d = descs[c];
            const int slen = (int)(std::min<int64_t>(d.row0 + d.drows + 1, S) - d.row0);
            const int rlen = (int)(std::min<int64_t>(d.col0 + d.dcols + 1, R) - d.col0);
            pack_chunk_words(seq + d.row0, slen, d.drows, ref + d.col0, rlen, d.dcols, max_n, max_l,
                             reinterpret_cast<uint32_t *>(hs + o_seqw) + d.seqw_off,
                             reinterpret_cast<uint32_t *>(hs + o_refw) + 2 * d.refw_off,
                             hs + o_refl + 8 * d.refw_off, scratch);
        }
    });
    ctx->timing[5] += now_ms() - t_prep0;

    // ---- device buffers
    const int64_t nref_bytes = a.ref_off[g1] - bases_ref0, nseq_bytes = a.seq_off[g1] - bases_seq0;
    if (int rc = ctx->refs.ensure(nref_bytes + 16)) return rc;
    if (int rc = ctx->seqs.ensure(nseq_bytes + 16)) return rc;
    if (int rc = ctx->steps.ensure(stage_bytes)) return rc;   // whole staging block lands in one device block
    if (int rc = ctx->tb.ensure((size_t)tb_tot * 4 + 64)) return rc;
    if (int rc = ctx->cout_.ensure(out_tot + 64)) return rc;
    if (int rc = ctx->clen.ensure(n_chunks * 4 + 64)) return rc;
    if (int rc = ctx->cstat.ensure(n_chunks * 4 + 64)) return rc;

    HIP_TRY(hipEventRecord(ctx->ev[0], s));
    HIP_TRY(hipMemcpyAsync(ctx->steps.p, hs, stage_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ctx->refs.p, a.refs + bases_ref0, nref_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ctx->seqs.p, a.seqs + bases_seq0, nseq_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(ctx->ev[1], s));

    uint8_t *db = ctx->steps.as<uint8_t>();
    if (n_chunks > 0) {
        KParams kp;
        kp.descs = reinterpret_cast<const ChunkDesc *>(db + o_desc);
        kp.sched = reinterpret_cast<const int32_t *>(db + o_sched);
        kp.steps = db + o_steps;
        kp.inss = reinterpret_cast<const int32_t *>(db + o_inss);
        kp.seqw = reinterpret_cast<const uint32_t *>(db + o_seqw);
        kp.refw = reinterpret_cast<const uint2 *>(db + o_refw);
        kp.refl = reinterpret_cast<const uint2 *>(db + o_refl);
        kp.tb = ctx->tb.as<uint32_t>();
        kp.sub_scores = ctx->d_sub;
        kp.np_scores = ctx->d_np;
        kp.max_n = ctx->max_n;
        kp.max_l = ctx->max_l;
        kp.r = r;
        kp.tbstride = tbs;
        kp.indel_start = a.indel_start;
        kp.indel_extend = a.indel_extend;
        hipError_t e = hipSuccess;
        switch (ng) {
            case 1 * 16 + 1: e = launch_fill<1, 1, 1024>(kp, (int)n_chunks, ctx->force_chunks, s); break;
            case 1 * 16 + 2: e = launch_fill<1, 2, 512>(kp, (int)n_chunks, ctx->force_chunks, s); break;
            case 1 * 16 + 4: e = launch_fill<1, 4, 256>(kp, (int)n_chunks, ctx->force_chunks, s); break;
            case 1 * 16 + 8: e = launch_fill<1, 8, 128>(kp, (int)n_chunks, ctx->force_chunks, s); break;
            case 2 * 16 + 1: e = launch_fill<2, 1, 1024>(kp, (int)n_chunks, ctx->force_chunks, s); break;
            case 2 * 16 + 2: e = launch_fill<2, 2, 512>(kp, (int)n_chunks, ctx->force_chunks, s); break;
            case 2 * 16 + 4: e = launch_fill<2, 4, 256>(kp, (int)n_chunks, ctx->force_chunks, s); break;
            case 4 * 16 + 1: e = launch_fill<4, 1, 1024>(kp, (int)n_chunks, ctx->force_chunks, s); break;
            case 4 * 16 + 2: e = launch_fill<4, 2, 512>(kp, (int)n_chunks, ctx->force_chunks, s); break;
            case 8 * 16 + 1: e = launch_fill<8, 1, 1024>(kp, (int)n_chunks, ctx->force_chunks, s); break;
            default: return fail(NPORE_E_UNSUPPORTED, "unsupported waves-per-chunk / columns-per-lane combination");
        }
        if (e != hipSuccess) return fail(NPORE_E_HIP, std::string("fill launch: ") + hipGetErrorString(e));
        HIP_TRY(hipEventRecord(ctx->ev[2], s));

        TParams tp;
        tp.descs = kp.descs;
        tp.n_chunks = (int)n_chunks;
        tp.inss = kp.inss;
        tp.tb = kp.tb;
        tp.seqs = ctx->seqs.as<uint8_t>();
        tp.refs = ctx->refs.as<uint8_t>();
        tp.chunk_out = ctx->cout_.as<uint8_t>();
        tp.chunk_len = ctx->clen.as<int32_t>();
        tp.chunk_status = ctx->cstat.as<int32_t>();
        tp.r = r;
        tp.tbstride = tbs;
        hipLaunchKernelGGL(traceback_kernel, dim3((unsigned)((n_chunks + 63) / 64)), dim3(64), 0, s, tp);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(hipEventRecord(ctx->ev[2], s));
    }
    HIP_TRY(hipEventRecord(ctx->ev[3], s));

    GParams gp;
    gp.descs = reinterpret_cast<const ChunkDesc *>(db + o_desc);
    gp.read_first_chunk = reinterpret_cast<const int32_t *>(db + o_rfc);
    gp.chunk_out = ctx->cout_.as<uint8_t>();
    gp.chunk_len = ctx->clen.as<int32_t>();
    gp.chunk_status = ctx->cstat.as<int32_t>();
    gp.read_status_in = reinterpret_cast<const int32_t *>(db + o_rstat);
    gp.out = ot.d_out;
    gp.out_off = ot.d_out_off;
    gp.out_len = ot.d_out_len;
    gp.status = ot.d_status;
    gp.read_base = g0;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)nr), dim3(256), 0, s, gp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev[4], s));
    HIP_TRY(hipStreamSynchronize(s));   // staging and device blocks are reused by the next group

    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1])); ctx->timing[3] += ms;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2])); ctx->timing[1] += ms;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); ctx->timing[2] += ms;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4])); ctx->timing[2] += ms;
    ctx->timing[6] += (double)tb_tot / tbs * (2 * r + 1);
    ctx->timing[7] += 1;
    return NPORE_OK;
}

int run_core(npore_ctx *ctx, const AlignArgs &a, const OutTarget &ot, hipStream_t s)
{
    if (a.n_reads < 0) return fail(NPORE_E_INVALID, "n_reads < 0");
    if (a.r < 1) return fail(NPORE_E_INVALID, "r must be >= 1");
    if (a.max_b_rows < 2) return fail(NPORE_E_INVALID, "max_b_rows must be >= 2");
    if (a.max_b_rows > 60000)
        return fail(NPORE_E_UNSUPPORTED, "max_b_rows > 60000: run lengths are kept in 16 bits");
    const int ng = pick_shape(a.r, ctx->force_nw, ctx->force_ng);
    if (!ng) return fail(NPORE_E_UNSUPPORTED, "band half-width r > 255");
    std::fill(ctx->timing, ctx->timing + 8, 0.0);
    if (a.n_reads == 0) return NPORE_OK;
    HIP_TRY(hipSetDevice(ctx->device));

    const int threads = ctx->host_threads > 0 ? ctx->host_threads
                                              : (int)std::max(1u, std::thread::hardware_concurrency());
    const double t0 = now_ms();
    std::vector<ReadPlan> plans(a.n_reads);
    parallel_for(a.n_reads, threads, [&](int64_t i) {
        const int64_t S = a.seq_off[i + 1] - a.seq_off[i], R = a.ref_off[i + 1] - a.ref_off[i];
        ReadPlan &pl = plans[i];
        pl.ok = build_path(a.cigs + a.cig_off[i], a.cig_off[i + 1] - a.cig_off[i], S, R, a.max_b_rows, pl.path);
        if (pl.ok) {
            const uint8_t *sq = a.seqs + a.seq_off[i], *rf = a.refs + a.ref_off[i];
            for (int64_t k = 0; k < S && pl.ok; k++) pl.ok = sq[k] <= 4;
            for (int64_t k = 0; k < R && pl.ok; k++) pl.ok = rf[k] <= 4;
        }
        pl.n_chunks = pl.ok ? (int)pl.path.breaks.size() - 1 : 0;
    });
    ctx->timing[5] += now_ms() - t0;

    // groups bounded by the traceback budget
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    int64_t budget = ctx->tb_budget_mb > 0 ? ctx->tb_budget_mb * (int64_t)1048576
                                           : (int64_t)((double)(free_b + ctx->tb.cap) * 0.6);
    const int tbs = tb_stride(a.r);
    int64_t g0 = 0;
    while (g0 < a.n_reads) {
        int64_t g1 = g0, acc = 0;
        while (g1 < a.n_reads) {
            const int64_t B = plans[g1].ok ? (int64_t)plans[g1].path.inss.size() + plans[g1].n_chunks : 0;
            const int64_t need = B * tbs * 4;
            if (g1 > g0 && acc + need > budget) break;
            acc += need;
            g1++;
        }
        if (int rc = run_group(ctx, a, g0, g1, plans, ot, s, ng)) return rc;
        for (int64_t i = g0; i < g1; i++) plans[i] = ReadPlan();   // free host memory early
        g0 = g1;
    }
    return NPORE_OK;
}

}  // namespace

extern "C" {

int npore_abi_version(void) { return NPORE_ABI_VERSION; }
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
{
    if (!sub_scores || !np_scores || max_n < 1 || max_n > MAX_PERIOD || max_l < 2 || max_l > 255) {
        fail(NPORE_E_INVALID, "npore_ctx_create: need tables, 1 <= max_n <= 6, 2 <= max_l <= 255");
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
    ctx->device = device_id;
    ctx->max_n = max_n;
    ctx->max_l = max_l;
    const size_t np_elems = (size_t)max_n * (max_l + 1) * (max_l + 1);
    ctx->h_sub.assign(sub_scores, sub_scores + 25);
    ctx->h_np.assign(np_scores, np_scores + np_elems);
    bool ok = hipSetDevice(device_id) == hipSuccess && hipStreamCreate(&ctx->stream) == hipSuccess &&
              hipMalloc((void **)&ctx->d_sub, 25 * sizeof(float)) == hipSuccess &&
              hipMalloc((void **)&ctx->d_np, np_elems * sizeof(float)) == hipSuccess &&
              hipMemcpy(ctx->d_sub, sub_scores, 25 * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(ctx->d_np, np_scores, np_elems * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    for (auto &e : ctx->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    if (!ok) {
        fail(NPORE_E_HIP, "npore_ctx_create: HIP initialisation failed");
        npore_ctx_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

void npore_ctx_destroy(npore_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (DevBuf *b : {&ctx->refs, &ctx->seqs, &ctx->steps, &ctx->inss, &ctx->seqw, &ctx->refw, &ctx->refl,
                      &ctx->descs, &ctx->sched, &ctx->rfc, &ctx->rstat, &ctx->tb, &ctx->cout_, &ctx->clen,
                      &ctx->cstat, &ctx->out, &ctx->out_off, &ctx->out_len, &ctx->status})
        b->release();
    ctx->h_stage.release();
    if (ctx->d_sub) (void)hipFree(ctx->d_sub);
    if (ctx->d_np) (void)hipFree(ctx->d_np);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int npore_align_batch(npore_ctx *ctx, int64_t n_reads, const uint8_t *refs, const int64_t *ref_off,
                      const uint8_t *seqs, const int64_t *seq_off, const char *cigars, const int64_t *cig_off,
                      float indel_start, float indel_extend, int max_b_rows, int r, char *out,
                      const int64_t *out_off, int64_t *out_len, int32_t *status)
{
    if (!ctx) return fail(NPORE_E_INVALID, "null context");
    if (n_reads > 0 && (!ref_off || !seq_off || !cig_off || !out_off || !out_len || !status))
        return fail(NPORE_E_INVALID, "null argument");
    AlignArgs a{n_reads, refs, ref_off, seqs, seq_off, cigars, cig_off, indel_start, indel_extend, max_b_rows, r};
    if (n_reads == 0) return run_core(ctx, a, OutTarget{}, ctx->stream);
    HIP_TRY(hipSetDevice(ctx->device));
    const int64_t out_bytes = out_off[n_reads] - out_off[0];
    if (out_bytes < 0) return fail(NPORE_E_INVALID, "out_off not ascending");
    if (int rc = ctx->out.ensure(out_bytes + 16)) return rc;
    if (int rc = ctx->out_off.ensure((n_reads + 1) * 8)) return rc;
    if (int rc = ctx->out_len.ensure(n_reads * 8)) return rc;
    if (int rc = ctx->status.ensure(n_reads * 4)) return rc;
    std::vector<int64_t> rebased(n_reads + 1);
    for (int64_t i = 0; i <= n_reads; i++) rebased[i] = out_off[i] - out_off[0];
    HIP_TRY(hipMemcpyAsync(ctx->out_off.p, rebased.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    OutTarget ot{ctx->out.as<uint8_t>(), ctx->out_off.as<int64_t>(), ctx->out_len.as<int64_t>(),
                 ctx->status.as<int32_t>()};
    if (int rc = run_core(ctx, a, ot, ctx->stream)) return rc;
    HIP_TRY(hipEventRecord(ctx->ev[5], ctx->stream));
    HIP_TRY(hipMemcpyAsync(out + out_off[0], ctx->out.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_len, ctx->out_len.p, n_reads * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(status, ctx->status.p, n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev[6], ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[6]));
    ctx->timing[4] += ms;
    return NPORE_OK;
}

int npore_align_batch_device(npore_ctx *ctx, int64_t n_reads, const uint8_t *d_refs, const int64_t *d_ref_off,
                             const uint8_t *d_seqs, const int64_t *d_seq_off, const char *d_cigars,
                             const int64_t *d_cig_off, float indel_start, float indel_extend, int max_b_rows,
                             int r, char *d_out, const int64_t *d_out_off, int64_t *d_out_len,
                             int32_t *d_status, void *stream, int sync)
{
    if (!ctx) return fail(NPORE_E_INVALID, "null context");
    if (n_reads == 0) return NPORE_OK;
    if (!d_ref_off || !d_seq_off || !d_cig_off || !d_out_off || !d_out_len || !d_status)
        return fail(NPORE_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    // Round-1: path conversion / n-polymer annotation still run on the host, so the
    // raw inputs make one round trip.  (Replaced by device prep kernels next.)
    std::vector<int64_t> ro(n_reads + 1), so(n_reads + 1), co(n_reads + 1);
    HIP_TRY(hipMemcpyAsync(ro.data(), d_ref_off, (n_reads + 1) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(so.data(), d_seq_off, (n_reads + 1) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(co.data(), d_cig_off, (n_reads + 1) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<uint8_t> refs(ro[n_reads]), seqs(so[n_reads]);
    std::vector<char> cigs(co[n_reads]);
    HIP_TRY(hipMemcpyAsync(refs.data(), d_refs, ro[n_reads], hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(seqs.data(), d_seqs, so[n_reads], hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(cigs.data(), d_cigars, co[n_reads], hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    AlignArgs a{n_reads, refs.data(), ro.data(), seqs.data(), so.data(), cigs.data(), co.data(),
                indel_start, indel_extend, max_b_rows, r};
    OutTarget ot{reinterpret_cast<uint8_t *>(d_out), d_out_off, d_out_len, d_status};
    if (int rc = run_core(ctx, a, ot, s)) return rc;
    (void)sync;   // run_core synchronises per group
    return NPORE_OK;
}

int npore_get_np_info(npore_ctx *ctx, const uint8_t *seq, int64_t len, int32_t *out)
{
    if (!ctx || (len > 0 && (!seq || !out))) return fail(NPORE_E_INVALID, "null argument");
    if (len <= 0) return NPORE_OK;
    const int mn = ctx->max_n;
    std::vector<int32_t> L((size_t)len * mn), I((size_t)len * mn);
    np_info_compute(seq, len, mn, ctx->max_l, L.data(), I.data());
    for (int64_t p = 0; p < len; p++)
        for (int n = 0; n < mn; n++) {
            out[(p * 2 + 0) * mn + n] = L[p * mn + n];
            out[(p * 2 + 1) * mn + n] = I[p * mn + n];
        }
    return NPORE_OK;
}

int npore_last_timing(npore_ctx *ctx, double *ms, int n)
{
    if (!ctx || !ms) return fail(NPORE_E_INVALID, "null argument");
    for (int i = 0; i < n && i < 8; i++) ms[i] = ctx->timing[i];
    return NPORE_OK;
}

int npore_ctx_set(npore_ctx *ctx, const char *key, int64_t value)
{
    if (!ctx || !key) return fail(NPORE_E_INVALID, "null argument");
    const std::string k(key);
    if (k == "tb_budget_mb") ctx->tb_budget_mb = value;
    else if (k == "force_ng") ctx->force_ng = (int)value;
    else if (k == "force_chunks") ctx->force_chunks = (int)value;
    else if (k == "force_nw") ctx->force_nw = (int)value;
    else if (k == "host_threads") ctx->host_threads = (int)value;
    else return fail(NPORE_E_INVALID, "unknown key " + k);
    return NPORE_OK;
}

// debug / self-test entry (used by tests -m gpu): DPP lane shift directions
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

}  // extern "C"
