// hostio.hpp -- BAM ingest and SAM emit around the batched align(), host side, native.
//
// Counterpart of the reference's compiled I/O layer for this path: pysam's fetch + the per-read
// accessors used by get_read_data (reference src/bam.pyx:18-47), the glue of realign_read
// (src/bam.pyx:51-84: expand_cigar / bases_to_int before align(), standardisation and the SAM
// line after it) -- for a whole batch of reads on all host cores instead of one read per
// Python call.  npore_amd/bam.py keeps a pure-Python restatement of the same logic; the tests
// compare the two record by record.
//
// BAM = BGZF (independent deflate blocks of <= 64 KiB, each announcing its compressed size in a
// gzip extra field and its inflated size in its trailer) around a simple binary record stream,
// so the blocks are located with one pass over the file and inflated in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "crc32.hpp"
#include "inflate.hpp"

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "glue.hpp"

namespace npore {

// default worker count: these loops are memory-bound byte shuffles, more than 64 threads only add start-up cost
// CPUs this process may actually use: the affinity mask, cut down by a cgroup CPU quota if there is one (a container
// on a 256-CPU host with a 16-CPU quota sees 256 CPUs; 64 threads there only thrash and grow 64 malloc arenas)
inline int usable_cpus()
{
    static const int n = [] {
        int cpus = (int)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = std::max(1, CPU_COUNT(&set));
        if (FILE *fh = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                    // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            long long period = 0;
            if (std::fscanf(fh, "%31s %lld", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0)
                cpus = std::min<long long>(cpus, std::max<long long>(1, (std::atoll(q) + period - 1) / period));
            std::fclose(fh);
        } else if (FILE *f1 = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
            long long quota = -1, period = 0;
            if (std::fscanf(f1, "%lld", &quota) != 1) quota = -1;
            std::fclose(f1);
            if (FILE *f2 = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(f2, "%lld", &period) != 1) period = 0;
                std::fclose(f2);
            }
            if (quota > 0 && period > 0) cpus = std::min<long long>(cpus, std::max<long long>(1, (quota + period - 1) / period));
        }
        return cpus;
    }();
    return n;
}
inline int host_threads(int threads)
{
    return threads > 0 ? threads : std::min(64, usable_cpus());
}

// uninitialised byte buffer (a std::vector would zero hundreds of megabytes per batch)
// NPORE_ALLOC_TRACE=1: one line on stderr per device / page-locked allocation (what a cold run pays once)
inline bool alloc_trace_on() { static const bool on = std::getenv("NPORE_ALLOC_TRACE") != nullptr; return on; }
struct AllocTrace {
    const char *what;
    size_t bytes;
    std::chrono::steady_clock::time_point t0;
    AllocTrace(const char *w, size_t n) : what(w), bytes(n) { if (alloc_trace_on()) t0 = std::chrono::steady_clock::now(); }
    ~AllocTrace()
    {
        if (!alloc_trace_on()) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "alloc %-14s %12zu bytes %8.2f ms\n", what, bytes, ms);
    }
};

struct RawBuf {
    char *p = nullptr;
    size_t cap = 0;
    bool pinned = false;       // page-locked (hipHostMalloc): buffers that cross PCIe every batch
    RawBuf() = default;
    explicit RawBuf(bool pin) : pinned(pin) {}
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { release(); }
    void release()
    {
        if (!p) return;
        if (pinned) (void)hipHostFree(p);
        else std::free(p);
        p = nullptr;
        cap = 0;
    }
    bool ensure(size_t n)
    {
        if (n <= cap) return true;
        release();
        n += n / 4;            // head-room: batches of a run differ a little in size
        if (pinned) {
            void *q = nullptr;
            AllocTrace tr("hipHostMalloc", n);
            if (hipHostMalloc(&q, n, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                pinned = false;
                if (std::getenv("NPORE_DEBUG")) std::fprintf(stderr, "npore: hipHostMalloc(%zu) failed, pageable staging buffer\n", n);
            }
            else p = static_cast<char *>(q);
        }
        if (!pinned) {
            p = static_cast<char *>(std::malloc(n));
            // a large buffer is touched for the first time by all threads at once (the FASTA's bases, an inflated window,
            // a batch's SAM text): with 2 MB pages that is one page fault per 2 MB instead of 512 (where the kernel
            // offers transparent huge pages on request; a no-op elsewhere)
            if (p && n >= (size_t)32 << 20 && std::getenv("NPORE_NO_THP") == nullptr) {
                const uintptr_t a = ((uintptr_t)p + ((size_t)2 << 20) - 1) & ~(uintptr_t)(((size_t)2 << 20) - 1);
                const uintptr_t e = ((uintptr_t)p + n) & ~(uintptr_t)(((size_t)2 << 20) - 1);
                if (e > a) (void)::madvise(reinterpret_cast<void *>(a), e - a, MADV_HUGEPAGE);
            }
        }
        cap = p ? n : 0;
        return p != nullptr;
    }
};

// Write the cache lines of [p, p + n) back to memory.  A page-locked buffer that host threads have just filled is
// still spread over the caches of the cores that wrote it, and the DMA engine then reads it at a fraction of the
// PCIe rate (measured: 3-5 GB/s, against 50 GB/s for the same buffer once it has aged out of the caches); with the
// lines written back first the upload runs at the link's rate.  clwb keeps the (now clean) line for the host's own
// later reads; clflushopt / clflush where the CPU has no clwb.
#if defined(__x86_64__)
#include <cpuid.h>
inline int cache_writeback_kind()
{
    static const int kind = [] {
        unsigned a = 0, b = 0, c = 0, d = 0;
        if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return 0;
        return (b & (1u << 24)) ? 2 : (b & (1u << 23)) ? 1 : 0;      // CLWB, CLFLUSHOPT
    }();
    return kind;
}
inline void cache_writeback(const void *p, size_t n)
{
    if (!n) return;
    const int kind = cache_writeback_kind();
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)63, a1 = reinterpret_cast<uintptr_t>(p) + n;
    for (uintptr_t a = a0; a < a1; a += 64) {
        if (kind == 2) asm volatile(".byte 0x66, 0x0f, 0xae, 0x30" ::"a"(a) : "memory");        // clwb (%rax)
        else if (kind == 1) asm volatile(".byte 0x66, 0x0f, 0xae, 0x38" ::"a"(a) : "memory");   // clflushopt (%rax)
        else asm volatile("clflush (%0)" ::"r"(a) : "memory");
    }
    asm volatile("sfence" ::: "memory");
}
#else
inline void cache_writeback(const void *, size_t) {}
#endif

template <class F>
inline void parallel_for(int64_t n, int threads, F &&body)   // body(i) for i in [0, n), dynamic
{
    const int nt = (int)std::min<int64_t>(host_threads(threads), n);
    std::atomic<int64_t> next{0};
    // an exception of a body (an allocation that fails) must not leave its thread: the first one is kept, the other
    // workers stop taking indices, and the caller sees it after the join (the C ABI's entry points catch it there)
    std::exception_ptr failed;
    std::atomic<bool> stop{false};
    std::mutex failed_m;
    auto work = [&] {
        try {
            for (;;) {
                const int64_t i = next.fetch_add(1);
                if (i >= n || stop.load(std::memory_order_relaxed)) break;
                body(i);
            }
        } catch (...) {
            std::lock_guard<std::mutex> lk(failed_m);
            if (!failed) failed = std::current_exception();
            stop.store(true);
        }
    };
    if (nt <= 1) { work(); if (failed) std::rethrow_exception(failed); return; }
    std::vector<std::thread> pool;
    try {
        for (int t = 0; t < nt; t++) pool.emplace_back(work);
    } catch (...) {                                     // (no more threads to be had: the ones started do the work)
        if (pool.empty()) work();
    }
    for (auto &t : pool) t.join();
    if (failed) std::rethrow_exception(failed);
}

// a whole file, read-only: mapped (pages come in as the worker threads touch them, nothing is copied or zeroed)
struct MappedFile {
    const uint8_t *p = nullptr;
    size_t n = 0;
    bool ok = false;
    MappedFile() = default;
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
    ~MappedFile() { if (p && n) ::munmap(const_cast<uint8_t *>(p), n); }
    bool open(const char *path)
    {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); return false; }
        n = (size_t)st.st_size;
        if (n) {
            void *q = ::mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (q == MAP_FAILED) { ::close(fd); n = 0; return false; }
            p = static_cast<const uint8_t *>(q);
        }
        ::close(fd);
        return ok = true;
    }
};

inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline int32_t rdi32(const uint8_t *p) { return (int32_t)rd32(p); }

// Inflate a BGZF file.  Falls back to member-by-member gzip decoding if a member lacks the BC field.
struct ByteSpan {
    const uint8_t *p;
    size_t n;
    size_t size() const { return n; }
    const uint8_t *data() const { return p; }
    const uint8_t &operator[](size_t i) const { return p[i]; }
};

struct BgzfBlock { uint64_t in_off, in_len, out_off, out_len; };    // deflate payload in the file; its place in the stream

// header of the BGZF member at p[0 .. avail): payload offset and member size; false = not the header of a BGZF member
// (need = bytes wanted, when more input could help; 0 when none can)
inline bool bgzf_header(const uint8_t *p, size_t avail, size_t &pay_off, size_t &bsize, size_t &need)
{
    need = 18;
    if (avail < 18) return false;
    if (p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) { need = 0; return false; }
    const size_t xlen = rd16(p + 10), xend = 12 + xlen;
    need = xend;
    if (xend > avail) return false;
    bsize = 0;
    for (size_t q = 12; q + 4 <= xend;) {
        const size_t slen = rd16(p + q + 2);
        if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= xend) bsize = (size_t)rd16(p + q + 4) + 1;
        q += 4 + slen;
    }
    if (!bsize || bsize < xlen + 20) { need = 0; return false; }
    pay_off = xend;
    return true;
}

// the whole member at p[0 .. avail): payload offset / length, inflated length, member size; false = not a complete BGZF
// member (need as above)
inline bool bgzf_member(const uint8_t *p, size_t avail, size_t &pay_off, size_t &pay_len, size_t &isize, size_t &bsize, size_t &need)
{
    if (!bgzf_header(p, avail, pay_off, bsize, need)) return false;
    need = bsize;
    if (bsize > avail) return false;
    pay_len = bsize - pay_off - 8;
    isize = rd32(p + bsize - 4);
    return true;
}

inline bool inflate_block(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, int force = 0)
{
    if (!out_len) return true;
    // force: 1 = this library's decoder only, 2 = zlib only (tests); 0 = the decoder, zlib for what it declines
    if (force != 2 && inflate_raw_fast(in, in_len, out, out_len)) return true;
    if (force == 1) return false;
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef *>(in);
    zs.avail_in = (uInt)in_len;
    zs.next_out = out;
    zs.avail_out = (uInt)out_len;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == out_len;
    inflateEnd(&zs);
    return ok;
}

// Whether the BGZF readers compare every member's CRC-32 with the one in its trailer (htslib does: the reference's reader
// raises on a damaged member whose bytes still inflate).  On unless NPORE_BGZF_CRC=0.
inline bool bgzf_check_crc()
{
    static const bool on = [] { const char *e = std::getenv("NPORE_BGZF_CRC"); return !(e && e[0] == '0'); }();
    return on;
}

// several blocks by one thread, side by side (inflate.hpp: as many dependent chains in one loop); zlib for what the decoder
// declines.  Returns the number of blocks that did not inflate.  crc: every job's payload is followed by its member's
// trailer (CRC-32 of the inflated bytes, little-endian), and a block whose bytes do not have it counts as not inflated.
inline int inflate_blocks(const FastInflate::Job *jobs, int n, int force = 0, bool crc = false)
{
    int bad = 0;
    auto crc_ok = [&](int k) {
        return !crc || crc32_fast(0u, jobs[k].out, jobs[k].out_len) == rd32(jobs[k].in + jobs[k].in_len);
    };
    if (force == 2) {
        for (int k = 0; k < n; k++) bad += !(inflate_block(jobs[k].in, jobs[k].in_len, jobs[k].out, jobs[k].out_len, 2) && crc_ok(k));
        return bad;
    }
    bool ok_small[64];
    std::vector<char> ok_big;
    bool *ok = ok_small;
    if (n > 64) { ok_big.resize((size_t)n); ok = reinterpret_cast<bool *>(ok_big.data()); }
    inflate_raw_fast_many(jobs, n, ok);
    for (int k = 0; k < n; k++) {
        if (!ok[k] && (force == 1 || !inflate_block(jobs[k].in, jobs[k].in_len, jobs[k].out, jobs[k].out_len, 2))) bad++;
        else if (!crc_ok(k)) bad++;
    }
    return bad;
}

inline bool bgzf_inflate(const ByteSpan &raw, int threads, RawBuf &out, size_t &out_size, std::string &err)
{
    typedef BgzfBlock Blk;
    std::vector<Blk> blocks;
    size_t p = 0, total = 0;
    bool is_bgzf = true;
    while (p + 18 <= raw.size()) {
        size_t po, pl, isz, bs, need;
        if (!bgzf_member(raw.data() + p, raw.size() - p, po, pl, isz, bs, need)) { is_bgzf = false; break; }
        blocks.push_back({p + po, pl, total, isz});
        total += isz;
        p += bs;
    }
    if (is_bgzf && p == raw.size()) {
        if (!out.ensure(total + 1)) { err = "out of memory"; return false; }
        out_size = total;
        std::atomic<int> bad{0};
        const int64_t per = 8;                                  // blocks per task
        parallel_for(((int64_t)blocks.size() + per - 1) / per, threads, [&](int64_t t) {
            FastInflate::Job jobs[8];
            int n = 0;
            uint8_t *const o = reinterpret_cast<uint8_t *>(out.p);
            for (size_t i = (size_t)(t * per); i < std::min(blocks.size(), (size_t)((t + 1) * per)); i++)
                jobs[n++] = {raw.data() + blocks[i].in_off, blocks[i].in_len, o + blocks[i].out_off, blocks[i].out_len};
            bad += inflate_blocks(jobs, n, 0, bgzf_check_crc());      // (the trailer follows the payload in the mapping)
        });
        if (bad) { err = "corrupt BGZF block"; return false; }
        return true;
    }
    // plain (multi-member) gzip
    std::vector<uint8_t> acc;
    p = 0;
    std::vector<uint8_t> chunk(1 << 20);
    while (p < raw.size()) {
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 31) != Z_OK) { err = "zlib init failed"; return false; }
        zs.next_in = const_cast<Bytef *>(raw.data() + p);
        zs.avail_in = (uInt)std::min<size_t>(raw.size() - p, 0x7fffffffu);
        int rc;
        do {
            zs.next_out = chunk.data();
            zs.avail_out = (uInt)chunk.size();
            rc = inflate(&zs, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); err = "not a gzip/BGZF stream"; return false; }
            acc.insert(acc.end(), chunk.data(), chunk.data() + (chunk.size() - zs.avail_out));
        } while (rc != Z_STREAM_END);
        p += zs.total_in;
        inflateEnd(&zs);
        if (zs.total_in == 0) break;
    }
    if (!out.ensure(acc.size() + 1)) { err = "out of memory"; return false; }
    std::memcpy(out.p, acc.data(), acc.size());
    out_size = acc.size();
    return true;
}

// A file read with pread(): the streamed BAM mode never maps the file, so that what it has read does not stay in
// this process's resident set.
struct PreadFile {
    int fd = -1;
    uint64_t size = 0;
    PreadFile() = default;
    PreadFile(const PreadFile &) = delete;
    PreadFile &operator=(const PreadFile &) = delete;
    ~PreadFile() { if (fd >= 0) ::close(fd); }
    bool open(const char *path)
    {
        fd = ::open(path, O_RDONLY);
        struct stat st;
        if (fd < 0 || ::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return false;
        size = (uint64_t)st.st_size;
        return true;
    }
    bool read(uint64_t off, void *dst, size_t n) const
    {
        char *d = static_cast<char *>(dst);
        while (n) {
            const ssize_t k = ::pread(fd, d, n, (off_t)off);
            if (k <= 0) return false;
            d += k; off += (uint64_t)k; n -= (size_t)k;
        }
        return true;
    }
};

// block table of a BGZF file, read sequentially in 8 MB pieces; false = not BGZF throughout
inline bool bgzf_scan(const PreadFile &f, std::vector<BgzfBlock> &blocks, uint64_t &total)
{
    // From member header to member header: the last four bytes of one member (its inflated length) and the header of the
    // next lie side by side, so the table costs one read of a few dozen bytes per member -- not one pass over the file
    // (1.45 GB of page cache took 0.18 s of a 0.83 s run that way).
    total = 0;
    const size_t HEAD = 64;
    std::vector<uint8_t> buf(4 + HEAD);
    uint64_t fp = 0;                                            // the member whose header is at buf[4 ..)
    size_t have = (size_t)std::min<uint64_t>(HEAD, f.size);
    if (have && !f.read(0, buf.data() + 4, have)) return false;
    while (fp < f.size) {
        size_t po, bs, need;
        if (!bgzf_header(buf.data() + 4, have, po, bs, need)) {
            if (need <= have || fp + need > f.size) return false;          // not a BGZF member, or cut short
            buf.resize(4 + need);                                           // a longer extra field than we guessed
            if (!f.read(fp + have, buf.data() + 4 + have, need - have)) return false;
            have = need;
            if (!bgzf_header(buf.data() + 4, have, po, bs, need)) return false;
        }
        if (fp + bs > f.size) return false;                     // the file ends inside a member
        const uint64_t nx = fp + bs;
        have = (size_t)std::min<uint64_t>(HEAD, f.size - nx);
        if (buf.size() < 4 + HEAD) buf.resize(4 + HEAD);
        if (!f.read(nx - 4, buf.data(), 4 + have)) return false;
        const size_t isz = rd32(buf.data());
        blocks.push_back({fp + po, bs - po - 8, total, isz});
        total += isz;
        fp = nx;
    }
    return true;
}

// inflate blocks [b0, b1) of the table back to back into dst (their out_off relative to blocks[b0].out_off)
inline bool bgzf_inflate_range(const PreadFile &f, const std::vector<BgzfBlock> &blocks, size_t b0, size_t b1, uint8_t *dst, int threads)
{
    if (b0 >= b1) return true;
    std::atomic<int> bad{0};
    const uint64_t o0 = blocks[b0].out_off;
    const int64_t per = 8;                                      // blocks per task
    // every task reads the compressed bytes of its own blocks (a few hundred KB, into a buffer its thread keeps): the
    // copy out of the page cache runs on all threads like the inflation, and no window-sized buffer is allocated and
    // faulted in per call.  (Rounds 3 - 4 read the whole range with one pread on the calling thread first: 40 MB per
    // 64 MB window of a BAM with real qualities, a serial 10 - 20 ms in front of every window.)
    parallel_for(((int64_t)(b1 - b0) + per - 1) / per, threads, [&](int64_t t) {
        static thread_local std::vector<uint8_t> comp;
        const size_t i0 = b0 + (size_t)t * per, i1 = std::min(b1, b0 + (size_t)(t + 1) * per);
        // (+ 8: the last block's trailer -- CRC-32, inflated length --; the table was made from whole members)
        const uint64_t in0 = blocks[i0].in_off, in1 = blocks[i1 - 1].in_off + blocks[i1 - 1].in_len + 8;
        if (comp.size() < (size_t)(in1 - in0) + 64) comp.resize((size_t)(in1 - in0) + 64);
        if (!f.read(in0, comp.data(), (size_t)(in1 - in0))) { bad += (int)(i1 - i0); return; }
        FastInflate::Job jobs[8];
        int n = 0;
        for (size_t i = i0; i < i1; i++)
            jobs[n++] = {comp.data() + (blocks[i].in_off - in0), blocks[i].in_len, dst + (blocks[i].out_off - o0), blocks[i].out_len};
        bad += inflate_blocks(jobs, n, 0, bgzf_check_crc());
    });
    return bad == 0;
}

// Virtual offsets of a .bai index's LINEAR index (SAM specification 5.2: per reference, for every 16 kb window the
// virtual file offset -- compressed offset of a BGZF block << 16 | offset inside the inflated block -- of the first
// alignment that overlaps the window), all references, ascending, without duplicates and zeros.  Each one is where a
// record starts: the cut points at which a coordinate-sorted BAM can be dealt to several readers without a pass over it.
inline bool bai_linear_offsets(const char *path, std::vector<uint64_t> &out, std::vector<uint8_t> *ref_has_reads = nullptr)
{
    MappedFile mf;
    if (!mf.open(path) || mf.n < 8 || std::memcmp(mf.p, "BAI\1", 4) != 0) return false;
    const uint8_t *p = mf.p;
    const size_t n = mf.n;
    size_t q = 4;
    auto have = [&](size_t k) { return q + k <= n; };
    if (!have(4)) return false;
    const int32_t n_ref = (int32_t)rd32(p + q); q += 4;
    if (n_ref < 0) return false;
    out.clear();
    if (ref_has_reads) ref_has_reads->assign((size_t)n_ref, 0);
    for (int32_t r = 0; r < n_ref; r++) {
        if (!have(4)) return false;
        const int32_t n_bin = (int32_t)rd32(p + q); q += 4;
        if (n_bin < 0) return false;
        if (ref_has_reads && n_bin > 0) (*ref_has_reads)[(size_t)r] = 1;
        for (int32_t b = 0; b < n_bin; b++) {
            if (!have(8)) return false;
            const int32_t n_chunk = (int32_t)rd32(p + q + 4); q += 8;
            if (n_chunk < 0 || !have((size_t)n_chunk * 16)) return false;
            q += (size_t)n_chunk * 16;
        }
        if (!have(4)) return false;
        const int32_t n_intv = (int32_t)rd32(p + q); q += 4;
        if (n_intv < 0 || !have((size_t)n_intv * 8)) return false;
        for (int32_t k = 0; k < n_intv; k++) {
            const uint64_t v = (uint64_t)rd32(p + q) | ((uint64_t)rd32(p + q + 4) << 32);
            q += 8;
            if (v) {
                out.push_back(v);
                if (ref_has_reads) (*ref_has_reads)[(size_t)r] = 1;
            }
        }
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    return true;
}

}  // namespace npore

namespace npore {
// The bytes of a batch of records: pointers into the resident stream, or (streamed handle) into `buf`, which holds
// the runs of consecutive BGZF blocks the batch's records lie in, inflated for this batch (bam_fetch).
struct RecFetch {
    RawBuf buf;
    std::vector<const uint8_t *> ptr;
};
}  // namespace npore

// ---- handles ---------------------------------------------------------------------------------
struct npore_bam {
    std::unique_ptr<npore::MappedFile> raw_map;   // the file itself when it already is an inflated BAM stream
    npore::RawBuf data_buf;               // inflated stream
    const uint8_t *data = nullptr;
    size_t data_size = 0;
    std::string text;                     // header text
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lens;
    std::vector<uint8_t> ref_has_reads;
    // STREAMED mode (large files): the inflated stream is never held as a whole.  The handle keeps the BGZF block
    // table and, per record, what selection needs (below); the bytes of a batch of records are inflated on demand
    // from the blocks that hold them (RecFetch).  data == nullptr then.
    bool streamed = false;
    std::unique_ptr<npore::PreadFile> file;
    std::vector<npore::BgzfBlock> blocks;
    std::vector<int32_t> m_ref, m_pos, m_span;     // per record: reference id, position, reference span
    std::vector<uint16_t> m_flag;
    std::vector<int64_t> rec_off;         // offset of each record's block_size field (in the inflated stream)
    std::vector<std::vector<int64_t>> by_ref;   // record indices per reference id, file order
    std::vector<uint8_t> ref_sorted;      // ... which is ascending in position (regions then need no full scan)
    std::vector<int64_t> ref_max_len;     // longest reference span of a record on that reference
    npore::RecFetch api_fetch;            // records of the last npore_bam_format_sam call (the const entry points keep theirs local)
    npore::RawBuf sam;                    // text of the last formatted batch
    npore::RawBuf w_finals;               // final CIGARs of the last batch (work buffer, reused)
    // ONE-PASS handle, several processes (npore_bam_set_share): the contiguous stretch of the record stream this process
    // walks, as offsets into the inflated stream; share_begin is where a record starts (a virtual offset of the file's
    // .bai linear index), share_end where the next process's stretch begins (UINT64_MAX: the end of the file)
    bool has_share = false;
    uint64_t share_begin = 0, share_end = UINT64_MAX;
    size_t share_block = 0;               // the BGZF block share_begin lies in
    double stage_ms[4] = {0, 0, 0, 0};    // last npore_bam_realign_batch: pack, align, standardise, format
    double file_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // last npore_bam_realign_file (npore_bam_file_timing)
};

struct npore_fasta {
    std::vector<std::string> names;
    npore::RawBuf bases;                  // all contigs back to back, upper-cased
    std::vector<int64_t> off;             // [n + 1] contig k = bases[off[k] .. off[k+1])
    uint64_t serial = 0;                  // unique per opened FASTA of the process (a context's device copy is keyed by it, not by the address)
    int64_t len(size_t k) const { return off[k + 1] - off[k]; }
    const char *seq(size_t k) const { return bases.p + off[k]; }
};

namespace npore {

// fixed part of a BAM alignment record, after block_size
struct RecView {
    const uint8_t *p;          // start of the fixed fields
    int32_t block_size;
    int32_t ref_id() const { return rdi32(p); }
    int32_t pos() const { return rdi32(p + 4); }
    int l_read_name() const { return p[8]; }
    int mapq() const { return p[9]; }
    int n_cigar() const { return rd16(p + 12); }
    int flag() const { return rd16(p + 14); }
    int32_t l_seq() const { return rdi32(p + 16); }
    const char *name() const { return reinterpret_cast<const char *>(p + 32); }
    const uint8_t *cigar() const { return p + 32 + l_read_name(); }
    const uint8_t *seq() const { return cigar() + 4 * (size_t)n_cigar(); }
    const uint8_t *qual() const { return seq() + ((size_t)l_seq() + 1) / 2; }
    const uint8_t *aux() const { return qual() + l_seq(); }
    const uint8_t *end() const { return p + block_size; }
    uint32_t cig(int k) const { return rd32(cigar() + 4 * (size_t)k); }
};

inline RecView rec_view(const uint8_t *q) { return RecView{q + 4, rdi32(q)}; }

inline RecView rec_of(const RecFetch &f, int64_t k) { return rec_view(f.ptr[(size_t)k]); }

inline bool bam_fetch(const npore_bam &b, const int64_t *idx, int64_t n, int threads, RecFetch &f, std::string &err)
{
    f.ptr.resize((size_t)n);
    if (!b.streamed) {
        for (int64_t k = 0; k < n; k++) f.ptr[(size_t)k] = b.data + b.rec_off[(size_t)idx[k]];
        return true;
    }
    // blocks touched by the records (a record may straddle blocks), as sorted runs of consecutive blocks
    auto block_of = [&](uint64_t off) {      // last block with out_off <= off and out_len > 0 covering it
        size_t lo = 0, hi = b.blocks.size();
        while (hi - lo > 1) { const size_t mid = (lo + hi) >> 1; if (b.blocks[mid].out_off <= off) lo = mid; else hi = mid; }
        return lo;
    };
    const int64_t n_rec = (int64_t)b.rec_off.size();
    std::vector<std::pair<size_t, size_t>> need((size_t)n);      // [first, last] block of each record
    for (int64_t k = 0; k < n; k++) {
        const uint64_t a = (uint64_t)b.rec_off[(size_t)idx[k]];
        const uint64_t e = idx[k] + 1 < n_rec ? (uint64_t)b.rec_off[(size_t)idx[k] + 1] : (uint64_t)b.data_size;
        need[(size_t)k] = {block_of(a), block_of(e ? e - 1 : 0)};
    }
    std::vector<std::pair<size_t, size_t>> runs(need);
    std::sort(runs.begin(), runs.end());
    size_t m = 0;
    for (size_t k = 0; k < runs.size(); k++) {
        if (m && runs[k].first <= runs[m - 1].second + 1) runs[m - 1].second = std::max(runs[m - 1].second, runs[k].second);
        else runs[m++] = runs[k];
    }
    runs.resize(m);
    std::vector<uint64_t> base(m + 1, 0);
    for (size_t k = 0; k < m; k++)
        base[k + 1] = base[k] + (b.blocks[runs[k].second].out_off + b.blocks[runs[k].second].out_len - b.blocks[runs[k].first].out_off);
    if (!f.buf.ensure((size_t)base[m] + 8)) { err = "out of memory"; return false; }
    for (size_t k = 0; k < m; k++)
        if (!bgzf_inflate_range(*b.file, b.blocks, runs[k].first, runs[k].second + 1, reinterpret_cast<uint8_t *>(f.buf.p) + base[k], threads)) {
            err = "corrupt BGZF block";
            return false;
        }
    for (int64_t k = 0; k < n; k++) {
        const size_t r = (size_t)(std::upper_bound(runs.begin(), runs.end(), std::make_pair(need[(size_t)k].first, (size_t)-1)) - runs.begin()) - 1;
        f.ptr[(size_t)k] = reinterpret_cast<const uint8_t *>(f.buf.p) + base[r] +
                           ((uint64_t)b.rec_off[(size_t)idx[k]] - b.blocks[runs[r].first].out_off);
    }
    return true;
}

// reference length consumed: M, D, N, =, X  (pysam reference_length)
inline int64_t rec_ref_len(const RecView &r)
{
    int64_t n = 0;
    for (int k = 0; k < r.n_cigar(); k++) {
        const uint32_t c = r.cig(k), op = c & 15u;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) n += c >> 4;
    }
    return n;
}

// soft-clipped bases at either end (a hard clip may precede / follow them), src/bam.pyx:42-44 via
// pysam query_alignment_sequence
inline void rec_clips(const RecView &r, int64_t &lead, int64_t &trail)
{
    const int nc = r.n_cigar();
    lead = trail = 0;
    if (nc >= 1 && (r.cig(0) & 15u) == 4) lead = r.cig(0) >> 4;
    if (nc > 1 && (r.cig(0) & 15u) == 5 && (r.cig(1) & 15u) == 4) lead = r.cig(1) >> 4;
    if (nc > 1 && (r.cig(nc - 1) & 15u) == 4) trail = r.cig(nc - 1) >> 4;
    if (nc > 2 && (r.cig(nc - 1) & 15u) == 5 && (r.cig(nc - 2) & 15u) == 4) trail = r.cig(nc - 2) >> 4;
}

// integer HP tag or 0 (src/bam.pyx:46)
inline int64_t rec_hp(const RecView &r)
{
    const uint8_t *q = r.aux(), *end = r.end();
    while (q + 3 <= end) {
        const bool is_hp = q[0] == 'H' && q[1] == 'P';
        const char typ = (char)q[2];
        q += 3;
        int64_t val = 0;
        size_t w = 0;
        if (q + 4 > end && typ != 'c' && typ != 'C' && typ != 'A' && typ != 'Z' && typ != 'H' && !(q + 2 <= end && (typ == 's' || typ == 'S')))
            return 0;                   // not enough bytes left for a 4-byte value
        if (q >= end) return 0;
        switch (typ) {
            case 'c': val = (int8_t)q[0]; w = 1; break;
            case 'C': val = q[0]; w = 1; break;
            case 's': val = (int16_t)rd16(q); w = 2; break;
            case 'S': val = rd16(q); w = 2; break;
            case 'i': val = rdi32(q); w = 4; break;
            case 'I': val = rd32(q); w = 4; break;
            case 'A': w = 1; break;
            case 'f': w = 4; break;
            case 'Z': case 'H': { const uint8_t *z = q; while (z < end && *z) z++; w = (size_t)(z - q) + 1; break; }
            case 'B': {
                if (q + 5 > end) return 0;
                const char sub = (char)q[0];
                const uint32_t cnt = rd32(q + 1);
                const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                w = 5 + (size_t)cnt * es;
                break;
            }
            default: return 0;
        }
        if (q + w > end) return 0;      // truncated tag
        if (is_hp && (typ == 'c' || typ == 'C' || typ == 's' || typ == 'S' || typ == 'i' || typ == 'I')) return val;
        q += w;
    }
    return 0;
}

static const char SEQ16[] = "=ACMGRSVTWYHKDBN";
static const char CIGOPS[] = "MIDNSHP=XB";

inline uint8_t base_code(char c)   // src/cig.pyx:212-229: 'NACGT-' -> 0..5, anything else 0 (upper case only)
{
    switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 3; case 'T': return 4; case '-': return 5; default: return 0; }
}

// base_code over a run of characters: compares and masks only, so the compiler vectorises it
inline void base_codes(const char *src, uint8_t *dst, int64_t n)
{
    for (int64_t q = 0; q < n; q++) {
        const char c = src[q];
        dst[q] = (uint8_t)((c == 'A' ? 1 : 0) + (c == 'C' ? 2 : 0) + (c == 'G' ? 3 : 0) + (c == 'T' ? 4 : 0) + (c == '-' ? 5 : 0));
    }
}

// base codes of n bases of a BAM record's 4-bit packed sequence from base `first` on: two per table lookup
inline void nibble_codes(const uint8_t *packed, int64_t first, uint8_t *dst, int64_t n)
{
    static const struct Tab {
        uint16_t pair[256];
        Tab() { for (int b = 0; b < 256; b++) pair[b] = (uint16_t)(base_code(SEQ16[b >> 4]) | (base_code(SEQ16[b & 15]) << 8)); }
    } tab;
    int64_t q = 0, t = first;
    if (n > 0 && (t & 1)) { dst[q++] = (uint8_t)(tab.pair[packed[t >> 1]] >> 8); t++; }
    const uint8_t *src = packed + (t >> 1);
    const int64_t pairs = (n - q) >> 1;
    for (int64_t j = 0; j < pairs; j++) {
        const uint16_t v = tab.pair[src[j]];
        dst[q + 2 * j] = (uint8_t)v;
        dst[q + 2 * j + 1] = (uint8_t)(v >> 8);
    }
    q += 2 * pairs;
    if (q < n) dst[q] = (uint8_t)tab.pair[src[pairs]];
}

// FASTA -> contig names + upper-cased bases, on all cores: the file is cut into pieces at line starts; a first
// pass counts the bases every piece contributes to the contig open at its start and to the contigs whose
// headers it contains, a second pass copies the lines to their final place.  Lines are trimmed of blanks, tabs
// and '\r' at both ends; a header's name is the text after '>' up to the first blank; bases before the first
// header are ignored.
inline bool fasta_parse(const ByteSpan &raw, int threads, npore_fasta &f)
{
    struct Piece {
        size_t beg = 0, end = 0;
        int64_t pre = 0;                              // bases before the piece's first header
        std::vector<std::pair<std::string, int64_t>> hdrs;   // (name, bases until the next header / piece end)
        int64_t pre_ctg = -1, pre_at = 0;             // destination of `pre`
        int64_t first_ctg = 0;                        // contig index of hdrs[0]
    };
    const size_t N = raw.n;
    const uint8_t *d = raw.p;
    const int nt = host_threads(threads);
    size_t np = std::max<size_t>(1, std::min<size_t>((size_t)nt * 4, N >> 18));
    std::vector<Piece> pc(np);
    size_t at = 0;
    for (size_t k = 0; k < np; k++) {
        pc[k].beg = at;
        size_t e = (k + 1 == np) ? N : std::max(at, N / np * (k + 1));
        if (e < N) {
            const void *nl = std::memchr(d + e, '\n', N - e);
            e = nl ? (size_t)(static_cast<const uint8_t *>(nl) - d) + 1 : N;
        }
        pc[k].end = at = e;
    }
    auto blank = [](uint8_t c) { return c == ' ' || c == '\t' || c == '\r'; };
    // calls on_header(name_begin, name_end) / on_bases(begin, end) for every line of [beg, end)
    auto walk = [&](size_t beg, size_t end, auto &&on_header, auto &&on_bases) {
        size_t p = beg;
        while (p < end) {
            const void *nl = std::memchr(d + p, '\n', end - p);
            const size_t e = nl ? (size_t)(static_cast<const uint8_t *>(nl) - d) : end;
            size_t le = e, ls = p;
            while (le > p && blank(d[le - 1])) le--;
            while (ls < le && (d[ls] == ' ' || d[ls] == '\t')) ls++;
            if (ls < le && d[ls] == '>') {
                size_t w = ls + 1;
                while (w < le && d[w] != ' ' && d[w] != '\t') w++;
                on_header(ls + 1, w);
            } else if (le > ls) {
                on_bases(ls, le);
            }
            p = e + 1;
        }
    };
    parallel_for((int64_t)np, threads, [&](int64_t k) {
        Piece &q = pc[(size_t)k];
        walk(q.beg, q.end,
             [&](size_t a, size_t b) { q.hdrs.emplace_back(std::string(reinterpret_cast<const char *>(d + a), b - a), 0); },
             [&](size_t a, size_t b) { (q.hdrs.empty() ? q.pre : q.hdrs.back().second) += (int64_t)(b - a); });
    });
    std::vector<int64_t> sizes;
    for (Piece &q : pc) {
        q.pre_ctg = (int64_t)sizes.size() - 1;
        if (q.pre_ctg >= 0) { q.pre_at = sizes.back(); sizes.back() += q.pre; }
        q.first_ctg = (int64_t)sizes.size();
        for (auto &h : q.hdrs) { f.names.push_back(std::move(h.first)); sizes.push_back(h.second); }
    }
    f.off.assign(sizes.size() + 1, 0);
    for (size_t k = 0; k < sizes.size(); k++) f.off[k + 1] = f.off[k] + sizes[k];
    if (!f.bases.ensure((size_t)f.off.back() + 1)) return false;
    parallel_for((int64_t)np, threads, [&](int64_t k) {
        const Piece &q = pc[(size_t)k];
        int64_t ctg = q.pre_ctg, next = q.first_ctg;
        char *w = ctg >= 0 ? f.bases.p + f.off[(size_t)ctg] + q.pre_at : nullptr;
        walk(q.beg, q.end,
             [&](size_t, size_t) { ctg = next++; w = f.bases.p + f.off[(size_t)ctg]; },
             [&](size_t a, size_t b) {
                 if (!w) return;
                 for (size_t i = a; i < b; i++) { const uint8_t c = d[i]; *w++ = (char)((c >= 'a' && c <= 'z') ? c - 32 : c); }
             });
    });
    return true;
}

}  // namespace npore
