// inflate_bench.cpp -- the BGZF readers' DEFLATE decoder (csrc/inflate.hpp) against zlib on the blocks of a BAM, one core.
//   g++ -O3 -std=c++17 -I npore_amd/csrc -o /tmp/inflate_bench scripts/microbench/inflate_bench.cpp -lz && /tmp/inflate_bench file.bam [reps]
// Prints MB/s of inflated bytes for both, the share of literal bytes, and checks that the two outputs are equal.
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "inflate.hpp"

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: inflate_bench file.bam [reps]\n"); return 2; }
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const size_t n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> file(n + 64);
    if (fread(file.data(), 1, n, f) != n) return 1;
    fclose(f);
    struct Blk { size_t in, in_len, out, out_len; };
    std::vector<Blk> blocks;
    size_t p = 0, total = 0;
    while (p + 18 <= n) {
        const size_t bsize = (size_t)(file[p + 16] | (file[p + 17] << 8)) + 1;
        const size_t isize = (size_t)file[p + bsize - 4] | ((size_t)file[p + bsize - 3] << 8) | ((size_t)file[p + bsize - 2] << 16) | ((size_t)file[p + bsize - 1] << 24);
        blocks.push_back({p + 18, bsize - 26, total, isize});
        total += isize;
        p += bsize;
    }
    std::vector<uint8_t> a(total + 64), b(total + 64);
    auto now = [] { return std::chrono::steady_clock::now(); };
    double best_fast = 1e30, best_z = 1e30, best_pair = 1e30;
    std::vector<uint8_t> c(total + 64);
    size_t declined_pair = 0;
    size_t declined = 0;
    for (int r = 0; r < reps; r++) {
        auto t0 = now();
        declined = 0;
        for (auto &k : blocks)
            if (k.out_len && !npore::inflate_raw_fast(file.data() + k.in, k.in_len, a.data() + k.out, k.out_len)) declined++;
        {
            auto tp0 = now();
            declined_pair = 0;
            for (size_t i = 0; i < blocks.size(); i += 8) {         // a task of the readers: eight blocks, NPORE_INFLATE_LANES side by side
                npore::FastInflate::Job jobs[8];
                bool ok[8];
                int m = 0;
                for (size_t j = i; j < blocks.size() && j < i + 8; j++)
                    jobs[m++] = {file.data() + blocks[j].in, blocks[j].in_len, c.data() + blocks[j].out, blocks[j].out_len};
                npore::inflate_raw_fast_many(jobs, m, ok);
                for (int j = 0; j < m; j++) declined_pair += !ok[j];
            }
            const double dp = std::chrono::duration<double>(now() - tp0).count();
            if (dp < best_pair) best_pair = dp;
        }
        auto t1 = now();
        for (auto &k : blocks) {
            if (!k.out_len) continue;
            z_stream z;
            memset(&z, 0, sizeof z);
            inflateInit2(&z, -15);
            z.next_in = file.data() + k.in; z.avail_in = (uInt)k.in_len;
            z.next_out = b.data() + k.out; z.avail_out = (uInt)k.out_len;
            if (inflate(&z, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "zlib failed\n"); return 1; }
            inflateEnd(&z);
        }
        auto t2 = now();
        const double df = std::chrono::duration<double>(t1 - t0).count(), dz = std::chrono::duration<double>(t2 - t1).count();
        if (df < best_fast) best_fast = df;
        if (dz < best_z) best_z = dz;
    }
    const bool same = (declined || memcmp(a.data(), b.data(), total) == 0) && (declined_pair || memcmp(c.data(), b.data(), total) == 0);
    printf("%zu blocks, %.1f MB compressed -> %.1f MB; csrc/inflate.hpp %.0f MB/s (declined %zu), %d blocks side by side %.0f MB/s (declined %zu), zlib %.0f MB/s, equal=%d\n",
           blocks.size(), n / 1e6, total / 1e6, total / 1e6 / best_fast, declined, NPORE_INFLATE_LANES, total / 1e6 / best_pair, declined_pair, total / 1e6 / best_z, (int)same);
    return same ? 0 : 1;
}
