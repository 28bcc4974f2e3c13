// inflate.hpp -- raw DEFLATE (RFC 1951) decoder for BGZF blocks.
//
// BGZF is a series of gzip members of at most 64 KB, each a raw deflate stream with a known output size; the
// one-pass BAM reader inflates every block of a file exactly once, and on a 16-CPU lease that inflation was the
// largest single item of the file pipeline's host time with zlib.  This decoder does what
// fast inflaters do -- a 64-bit bit buffer refilled eight bytes at a time, one table lookup per symbol (11-bit
// primary table for literal / length codes, 8-bit for distances, second-level tables for longer codes), matches
// copied in 8-byte words -- and nothing else: no streaming, no window (the output buffer is the window), no
// dictionary.  Every read and write is bounds-checked (a block's output lies between its neighbours', which other
// threads are writing: a match is copied in whole words only where eight bytes of the block's own room lie behind
// it).  Anything malformed -- or the few legal oddities it does not take, like a literal / length code of a single
// symbol -- returns false, and the caller then hands the block to zlib, which decides.
// SEVERAL BLOCKS AT A TIME (inflate_raw_fast_many): a DEFLATE stream decodes as one dependent chain -- the bits of a symbol are
// known only when the previous one has been looked up: mask, table load, shift, ~8 cycles per literal -- and a BAM with
// real base qualities is mostly literals (a uniform draw per base compresses to 6.6 bits: 82 us of inflation per 10 kb read
// and core).  BGZF blocks are independent streams, so one thread decodes two of them in ONE loop, a symbol of each in
// turn: two chains the core's out-of-order window overlaps.  A stream that leaves the fast region (end of a deflate
// block, the last bytes of its buffers, anything unusual) goes through the careful loop on its own and rejoins.
// Checked against zlib on every block of the test BAMs and on generated streams of all block types, alone and in pairs
// (tests/test_host_logic.py).
#pragma once
#include <cstdint>
#include <cstring>

namespace npore {

class FastInflate {
    // table entry: bits 0-7 number of code bits to consume (primary) / total incl. the first level (secondary),
    //              bits 8-15 kind, bits 16-31 value
    enum : uint32_t { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4, K_DIST = 5 };
    static constexpr int LB = 11, DB = 8;
    uint32_t lit_[(1 << LB) + 2048];       // primary + second-level tables
    uint32_t dist_[(1 << DB) + 1024];
    const uint8_t *in_, *in_end_;
    uint64_t bb_ = 0;
    int bc_ = 0;

    static uint32_t entry(uint32_t kind, uint32_t bits, uint32_t value) { return bits | (kind << 8) | (value << 16); }

    // canonical Huffman table of `n` symbols with code lengths `len` (0 = unused) into tab (primary bits PB);
    // sym_entry(symbol) gives kind / value of a symbol (its `bits` field is filled in here).  false: over-subscribed
    // or incomplete (a single code of one bit is accepted where the caller says so: the distance table, as zlib does)
    template <class SymEntry>
    static bool build(const uint8_t *len, int n, uint32_t *tab, int PB, int tab_cap, SymEntry sym_entry, bool one_code_ok = false)
    {
        int count[16] = {0};
        for (int i = 0; i < n; i++) count[len[i]]++;
        if (count[0] == n) {                      // no codes at all (a block without matches may have no distance code)
            for (int i = 0; i < (1 << PB); i++) tab[i] = entry(K_BAD, 1, 0);
            return true;
        }
        int left = 1;
        for (int l = 1; l < 16; l++) {
            left = (left << 1) - count[l];
            if (left < 0) return false;
        }
        if (left > 0 && !(one_code_ok && count[0] == n - 1 && count[1] == 1)) return false;     // incomplete (one 1-bit DISTANCE code is legal)
        uint16_t next[16];
        uint32_t code = 0;
        count[0] = 0;
        for (int l = 1; l < 16; l++) {
            code = (code + (uint32_t)count[l - 1]) << 1;
            next[l] = (uint16_t)code;
        }
        for (int i = 0; i < (1 << PB); i++) tab[i] = entry(K_BAD, 1, 0);
        int sub_at = 1 << PB;                     // where the next second-level table goes
        // second-level tables are indexed by the primary prefix; sized by the longest code sharing that prefix
        // pass 1: primary entries of short codes; the longest code length per long prefix
        uint8_t sub_bits[1 << 11];
        std::memset(sub_bits, 0, (size_t)1 << PB);
        uint16_t codes[320];
        for (int i = 0; i < n; i++) {
            const int l = len[i];
            if (!l) continue;
            uint32_t c = next[l]++, rev = 0;
            for (int b = 0; b < l; b++) rev |= ((c >> b) & 1u) << (l - 1 - b);     // codes are packed starting from their MSB: reverse
            codes[i] = (uint16_t)rev;
            if (l <= PB) {
                const uint32_t e = (sym_entry(i) & ~0xFFu) | (uint32_t)l;
                for (uint32_t k = rev; k < (1u << PB); k += 1u << l) tab[k] = e;
            } else {
                const uint32_t pre = rev & ((1u << PB) - 1);
                if (l - PB > sub_bits[pre]) sub_bits[pre] = (uint8_t)(l - PB);
            }
        }
        for (int i = 0; i < n; i++) {
            const int l = len[i];
            if (l <= PB) continue;
            const uint32_t rev = codes[i], pre = rev & ((1u << PB) - 1);
            if ((tab[pre] >> 8 & 0xFF) != K_SUB) {
                if (sub_at + (1 << sub_bits[pre]) > tab_cap) return false;
                tab[pre] = entry(K_SUB, sub_bits[pre], (uint32_t)sub_at);
                for (int k = 0; k < (1 << sub_bits[pre]); k++) tab[sub_at + k] = entry(K_BAD, (uint32_t)PB + 1, 0);
                sub_at += 1 << sub_bits[pre];
            }
            const uint32_t base = tab[pre] >> 16, sb = tab[pre] & 0xFF;
            const uint32_t e = (sym_entry(i) & ~0xFFu) | (uint32_t)l;
            for (uint32_t k = rev >> PB; k < (1u << sb); k += 1u << (l - PB)) tab[base + k] = e;
        }
        return true;
    }

    static uint32_t litlen_entry(int s)
    {
        static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        if (s < 256) return entry(K_LIT, 0, (uint32_t)s);
        if (s == 256) return entry(K_EOB, 0, 0);
        if (s > 285) return entry(K_BAD, 0, 0);
        return entry(K_LEN, 0, (uint32_t)base[s - 257] | ((uint32_t)extra[s - 257] << 12));      // value: base (<= 258: 9 bits ... 12) | extra bits << 12
    }
    static uint32_t dist_entry(int s)
    {
        static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        if (s >= 30) return entry(K_BAD, 0, 0);
        // a distance needs base (15 bits) and the extra-bit count (4 bits): the value field has 16, so the count rides in
        // the kind byte's upper half
        return (uint32_t)0 | ((K_DIST | ((uint32_t)extra[s] << 4)) << 8) | ((uint32_t)base[s] << 16);
    }

    // ---- bit reader (bits come LSB first)
    void refill_fast()                           // needs in_ + 8 <= in_end_
    {
        uint64_t w;
        std::memcpy(&w, in_, 8);
        bb_ |= w << bc_;
        in_ += (63 - bc_) >> 3;
        bc_ |= 56;
    }
    void refill()                                // at least 48 bits, or all that is left (zeros beyond the end)
    {
        if (in_ + 8 <= in_end_) { refill_fast(); return; }
        while (bc_ <= 56 && in_ < in_end_) { bb_ |= (uint64_t)*in_++ << bc_; bc_ += 8; }
        if (bc_ <= 56 && in_ == in_end_) over_ += (64 - bc_) >> 3, bc_ = 64 - ((64 - bc_) & 7);      // pretend zero bytes: the stream must end before it needs them
    }
    int over_ = 0;                               // zero bytes pretended beyond the input
    uint32_t peek(int n) const { return (uint32_t)(bb_ & ((1ull << n) - 1)); }
    void drop(int n) { bb_ >>= n; bc_ -= n; }
    uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }

    bool dynamic_tables()
    {
        refill();
        const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
        if (hlit > 286 || hdist > 30) return false;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; i++) { if (bc_ < 3) refill(); cl[order[i]] = (uint8_t)take(3); }
        uint32_t ct[1 << 7];
        if (!build(cl, 19, ct, 7, 1 << 7, [](int s) { return entry(K_LIT, 0, (uint32_t)s); })) return false;
        uint8_t lens[286 + 30];
        for (int i = 0; i < hlit + hdist;) {
            refill();
            const uint32_t e = ct[peek(7)];
            if ((e >> 8 & 0xFF) != K_LIT) return false;
            drop((int)(e & 0xFF));
            const int s = (int)(e >> 16);
            if (s < 16) { lens[i++] = (uint8_t)s; continue; }
            int rep;
            uint8_t v = 0;
            if (s == 16) { if (i == 0) return false; v = lens[i - 1]; rep = 3 + (int)take(2); }
            else if (s == 17) rep = 3 + (int)take(3);
            else rep = 11 + (int)take(7);
            if (i + rep > hlit + hdist) return false;
            while (rep--) lens[i++] = v;
        }
        if (lens[256] == 0) return false;        // no end-of-block code
        return build(lens, hlit, lit_, LB, (int)(sizeof lit_ / 4), litlen_entry) &&
               build(lens + hlit, hdist, dist_, DB, (int)(sizeof dist_ / 4), dist_entry, true);
    }
    bool fixed_tables()
    {
        uint8_t lens[288 + 32];
        for (int i = 0; i < 144; i++) lens[i] = 8;
        for (int i = 144; i < 256; i++) lens[i] = 9;
        for (int i = 256; i < 280; i++) lens[i] = 7;
        for (int i = 280; i < 288; i++) lens[i] = 8;
        for (int i = 0; i < 32; i++) lens[288 + i] = 5;
        return build(lens, 288, lit_, LB, (int)(sizeof lit_ / 4), litlen_entry) &&
               build(lens + 288, 32, dist_, DB, (int)(sizeof dist_ / 4), dist_entry);
    }

    // ---- the output side of the stream being decoded (the whole output so far is the window)
    uint8_t *out0_ = nullptr, *o_ = nullptr, *out_end_ = nullptr;
    bool last_ = false;

    // The fast loop: while 16 input bytes and 320 output bytes of the block's own room remain, nothing in it needs a
    // bounds check -- one refill feeds up to three literals or a whole length + distance pair (at most 48 bits), a
    // match is copied in words of eight bytes (at most 258 + 7 bytes written).  The table entry of the NEXT symbol is
    // looked up as soon as the bits of this one are consumed -- for a match: before its bytes are copied -- so the
    // lookup's latency lies beside the copy.  A symbol it cannot finish unchecked (a match reaching in front of the
    // output, a bad code, the end of the block) leaves the bit buffer untouched and falls to the careful loop, which decides.
    // One step = a run of up to three literals, or one match, of stream X (its state in the locals o##X, in##X, bb##X,
    // bc##X, e##X, the tables lit##X / dst##X, the limits in_stop##X / out_stop##X); leaves by `goto LEAVE` with the
    // stream's bits as they were in front of the symbol it could not take
#define NPORE_INFL_REFILL(X) do { uint64_t w_; std::memcpy(&w_, in##X, 8); bb##X |= w_ << bc##X; in##X += (63 - bc##X) >> 3; bc##X |= 56; } while (0)
#define NPORE_INFL_STEP(X, LEAVE)                                                                                        \
    do {                                                                                                                 \
        if ((e##X & 0xFF00u) == 0) {                   /* K_LIT: up to three from the bits at hand (<= 33 of >= 56) */   \
            bb##X >>= (e##X & 0xFF); bc##X -= (int)(e##X & 0xFF);                                                        \
            *o##X++ = (uint8_t)(e##X >> 16);                                                                             \
            e##X = lit##X[bb##X & ((1u << LB) - 1)];                                                                     \
            if ((e##X & 0xFF00u) == 0) {                                                                                 \
                bb##X >>= (e##X & 0xFF); bc##X -= (int)(e##X & 0xFF);                                                    \
                *o##X++ = (uint8_t)(e##X >> 16);                                                                         \
                e##X = lit##X[bb##X & ((1u << LB) - 1)];                                                                 \
                if ((e##X & 0xFF00u) == 0) {                                                                             \
                    bb##X >>= (e##X & 0xFF); bc##X -= (int)(e##X & 0xFF);                                                \
                    *o##X++ = (uint8_t)(e##X >> 16);                                                                     \
                    e##X = lit##X[bb##X & ((1u << LB) - 1)];                                                             \
                }                                                                                                        \
            }                                                                                                            \
            if (!(in##X <= in_stop##X && o##X <= out_stop##X)) goto LEAVE;                                               \
            NPORE_INFL_REFILL(X);                      /* (the low bits e was looked up with stay where they are) */     \
            break;                                                                                                       \
        }                                                                                                                \
        {                                                                                                                \
            const uint64_t bb0 = bb##X;                                                                                  \
            const int bc0 = bc##X;                                                                                       \
            if ((e##X >> 8 & 0xFF) == K_SUB) e##X = lit##X[(e##X >> 16) + ((bb##X >> LB) & ((1u << (e##X & 0xFF)) - 1))]; \
            const uint32_t kind = e##X >> 8 & 0xFF;                                                                      \
            bb##X >>= (e##X & 0xFF); bc##X -= (int)(e##X & 0xFF);                                                        \
            if (kind == K_LIT) {                                                                                         \
                *o##X++ = (uint8_t)(e##X >> 16);                                                                         \
                if (!(in##X <= in_stop##X && o##X <= out_stop##X)) goto LEAVE;                                           \
                NPORE_INFL_REFILL(X);                                                                                    \
                e##X = lit##X[bb##X & ((1u << LB) - 1)];                                                                 \
                break;                                                                                                   \
            }                                                                                                            \
            if (kind != K_LEN) { bb##X = bb0; bc##X = bc0; goto LEAVE; }       /* end of block, or a bad code */          \
            const uint32_t v = e##X >> 16, xl = v >> 12;                                                                 \
            const int len = (int)(v & 0xFFF) + (int)(bb##X & ((1u << xl) - 1));                                          \
            bb##X >>= xl; bc##X -= (int)xl;                                                                              \
            uint32_t d = dst##X[bb##X & ((1u << DB) - 1)];                                                               \
            if ((d >> 8 & 0xF) == K_SUB) d = dst##X[(d >> 16) + ((bb##X >> DB) & ((1u << (d & 0xFF)) - 1))];             \
            if ((d >> 8 & 0xF) != K_DIST) { bb##X = bb0; bc##X = bc0; goto LEAVE; }                                      \
            bb##X >>= (d & 0xFF); bc##X -= (int)(d & 0xFF);                                                              \
            const uint32_t xd = d >> 12 & 0xF;                                                                           \
            const size_t dist = (size_t)(d >> 16) + (size_t)(bb##X & ((1u << xd) - 1));                                  \
            bb##X >>= xd; bc##X -= (int)xd;                                                                              \
            if (dist > (size_t)(o##X - out0##X)) { bb##X = bb0; bc##X = bc0; goto LEAVE; }                               \
            /* the next symbol's entry before the copy (16 input bytes are there: the loop's condition held) */          \
            NPORE_INFL_REFILL(X);                                                                                        \
            e##X = lit##X[bb##X & ((1u << LB) - 1)];                                                                     \
            const uint8_t *src = o##X - dist;                                                                            \
            if (dist >= 8) {                             /* words of eight, one after the other: each reads bytes already written */ \
                uint64_t w;                                                                                              \
                std::memcpy(&w, src, 8); std::memcpy(o##X, &w, 8);                                                       \
                if (len > 8) {                                                                                           \
                    std::memcpy(&w, src + 8, 8); std::memcpy(o##X + 8, &w, 8);                                           \
                    for (int k = 16; k < len; k += 8) { std::memcpy(&w, src + k, 8); std::memcpy(o##X + k, &w, 8); }     \
                }                                                                                                        \
            } else if (dist == 1) {                                                                                      \
                const uint64_t w = 0x0101010101010101ull * (uint64_t)*src;                                               \
                for (int k = 0; k < len; k += 8) std::memcpy(o##X + k, &w, 8);                                           \
            } else {                                                                                                     \
                for (int k = 0; k < len; k++) o##X[k] = src[k];                                                          \
            }                                                                                                            \
            o##X += len;                                                                                                 \
            if (!(in##X <= in_stop##X && o##X <= out_stop##X)) goto LEAVE;                                               \
        }                                                                                                                \
    } while (0)
    // (the readers' state in named locals -- registers: a byte store may alias any member, and the compiler would reload
    // them all behind every literal; with arrays indexed by the lane the compiler kept the state on the stack)
#define NPORE_INFL_LOAD(X, S)                                                                                            \
    uint8_t *o##X = (S).o_, *const out0##X = (S).out0_;                                                                  \
    const uint8_t *in##X = (S).in_;                                                                                      \
    const uint8_t *const in_stop##X = (S).in_end_ - ((S).in_fast_region() ? 16 : 0);      /* (never formed outside the buffer) */ \
    uint8_t *const out_stop##X = (S).out_end_ - ((S).in_fast_region() ? 320 : 0);                                        \
    uint64_t bb##X = (S).bb_;                                                                                            \
    int bc##X = (S).bc_;                                                                                                 \
    const uint32_t *const lit##X = (S).lit_, *const dst##X = (S).dist_;                                                  \
    uint32_t e##X = 0
#define NPORE_INFL_STORE(X, S) do { (S).o_ = o##X; (S).in_ = in##X; (S).bb_ = bb##X; (S).bc_ = bc##X; } while (0)

    // The current compressed blocks' symbols of one stream, or of two with a step of each in turn, while the fast region
    // lasts for all of them: returns the index of the stream that has to leave it, or -1 if one of them was outside the
    // region to begin with.  Compiled twice on x86-64: for the baseline instruction set and with BMI / BMI2 (shifts by a
    // register count without the detour through CL, three-operand and-not: 380 -> 404 ... 417 MB/s on literal-heavy blocks
    // on the build container's core), chosen once per process by what the CPU has.
#define NPORE_INFL_DEFINE_FAST(SUFFIX, ATTR)                                                                             \
    ATTR static int codes_fast_1##SUFFIX(FastInflate &sa)                                                                \
    {                                                                                                                    \
        int who = -1;                                                                                                    \
        NPORE_INFL_LOAD(A, sa);                                                                                          \
        if (sa.in_fast_region()) {                                                                                       \
            NPORE_INFL_REFILL(A);                                                                                        \
            eA = litA[bbA & ((1u << LB) - 1)];       /* the next symbol's entry is always looked up ahead */              \
            who = 0;                                                                                                     \
            for (;;) NPORE_INFL_STEP(A, leave1);                                                                         \
        }                                                                                                                \
    leave1:                                                                                                              \
        NPORE_INFL_STORE(A, sa);                                                                                         \
        return who;                                                                                                      \
    }                                                                                                                    \
    ATTR static int codes_fast_2##SUFFIX(FastInflate &sa, FastInflate &sb)                                               \
    {                                                                                                                    \
        int who = -1;                                                                                                    \
        NPORE_INFL_LOAD(A, sa);                                                                                          \
        NPORE_INFL_LOAD(B, sb);                                                                                          \
        if (sa.in_fast_region() && sb.in_fast_region()) {                                                                \
            NPORE_INFL_REFILL(A);                                                                                        \
            eA = litA[bbA & ((1u << LB) - 1)];                                                                           \
            NPORE_INFL_REFILL(B);                                                                                        \
            eB = litB[bbB & ((1u << LB) - 1)];                                                                           \
            for (;;) {                                                                                                   \
                NPORE_INFL_STEP(A, leave_a);                                                                             \
                NPORE_INFL_STEP(B, leave_b);                                                                             \
            }                                                                                                            \
        leave_a:                                                                                                         \
            who = 0;                                                                                                     \
            goto leave2;                                                                                                 \
        leave_b:                                                                                                         \
            who = 1;                                                                                                     \
        }                                                                                                                \
    leave2:                                                                                                              \
        NPORE_INFL_STORE(A, sa);                                                                                         \
        NPORE_INFL_STORE(B, sb);                                                                                         \
        return who;                                                                                                      \
    }
    NPORE_INFL_DEFINE_FAST(, )
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    NPORE_INFL_DEFINE_FAST(_bmi2, __attribute__((target("bmi,bmi2"))))
    static bool cpu_has_bmi2()
    {
        static const bool has = __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
        return has;
    }
#else
    static bool cpu_has_bmi2() { return false; }
    static int codes_fast_1_bmi2(FastInflate &sa) { return codes_fast_1(sa); }
    static int codes_fast_2_bmi2(FastInflate &sa, FastInflate &sb) { return codes_fast_2(sa, sb); }
#endif
#undef NPORE_INFL_DEFINE_FAST
    template <int N>
    static int codes_fast_n(FastInflate *const *s)
    {
        static_assert(N == 1 || N == 2, "lanes");
        if constexpr (N == 1) return cpu_has_bmi2() ? codes_fast_1_bmi2(*s[0]) : codes_fast_1(*s[0]);
        else return cpu_has_bmi2() ? codes_fast_2_bmi2(*s[0], *s[1]) : codes_fast_2(*s[0], *s[1]);
    }
    void codes_fast()
    {
        FastInflate *one[1] = {this};
        codes_fast_n<1>(one);
    }
#undef NPORE_INFL_LOAD
#undef NPORE_INFL_STORE
#undef NPORE_INFL_STEP
#undef NPORE_INFL_REFILL

    // the rest of the current compressed block, every access checked: true at its end-of-block code
    bool codes_careful()
    {
        uint8_t *const out0 = out0_, *const out_end = out_end_;
        uint8_t *&out = o_;
        for (;;) {
            refill();
            uint32_t e = lit_[peek(LB)];
            if ((e >> 8 & 0xFF) == K_SUB) e = lit_[(e >> 16) + ((bb_ >> LB) & ((1u << (e & 0xFF)) - 1))];
            const uint32_t kind = e >> 8 & 0xFF;
            drop((int)(e & 0xFF));
            if (kind == K_LIT) {
                if (out == out_end) return false;
                *out++ = (uint8_t)(e >> 16);
                // a second and a third literal from the bits at hand (a code has at most 15 bits, 48 were there)
                uint32_t e2 = lit_[peek(LB)];
                if ((e2 >> 8 & 0xFF) == K_LIT && out < out_end) {
                    drop((int)(e2 & 0xFF));
                    *out++ = (uint8_t)(e2 >> 16);
                    e2 = lit_[peek(LB)];
                    if ((e2 >> 8 & 0xFF) == K_LIT && out < out_end) { drop((int)(e2 & 0xFF)); *out++ = (uint8_t)(e2 >> 16); }
                }
                continue;
            }
            if (kind == K_EOB) return bc_ >= 0;
            if (kind != K_LEN) return false;
            const uint32_t v = e >> 16;
            int len = (int)(v & 0xFFF) + (int)take((int)(v >> 12));
            if (bc_ < 32) refill();
            uint32_t d = dist_[peek(DB)];
            if ((d >> 8 & 0xF) == K_SUB) d = dist_[(d >> 16) + ((bb_ >> DB) & ((1u << (d & 0xFF)) - 1))];
            if ((d >> 8 & 0xF) != K_DIST) return false;
            drop((int)(d & 0xFF));
            const size_t dist = (size_t)(d >> 16) + take((int)(d >> 12 & 0xF));
            if (dist > (size_t)(out - out0) || (size_t)len > (size_t)(out_end - out)) return false;
            const uint8_t *src = out - dist;
            if (dist >= 8 && out_end - out >= len + 8) {              // whole words (may write up to 7 bytes beyond the match: inside the block)
                uint8_t *o = out;
                for (int k = 0; k < len; k += 8) { uint64_t w; std::memcpy(&w, src + k, 8); std::memcpy(o + k, &w, 8); }
            } else if (dist == 1) {
                std::memset(out, *src, (size_t)len);
            } else {
                for (int k = 0; k < len; k++) out[k] = src[k];
            }
            out += len;
        }
    }

    // header of the next block; stored blocks are copied here.  1: a compressed block's tables are ready, 2: the stream
    // has ended and everything checks, 0: malformed / declined
    int advance()
    {
        for (;;) {
            if (last_) return (o_ == out_end_ && over_ * 8 <= bc_) ? 2 : 0;      // every output byte, and no bit consumed beyond the input
            refill();
            last_ = take(1) != 0;
            const uint32_t type = take(2);
            if (type == 0) {                      // stored
                drop(bc_ & 7);
                refill();
                const uint32_t n = take(16), nn = take(16);
                if ((n ^ nn) != 0xFFFFu) return 0;
                // give back the whole bytes still in the bit buffer
                const int back = bc_ >> 3;
                if (over_ > back) return 0;
                in_ -= back - over_; over_ = 0; bb_ = 0; bc_ = 0;
                if ((size_t)(in_end_ - in_) < n || (size_t)(out_end_ - o_) < n) return 0;
                std::memcpy(o_, in_, n);
                o_ += n; in_ += n;
            } else if (type == 1 || type == 2) {
                return (type == 1 ? fixed_tables() : dynamic_tables()) ? 1 : 0;
            } else return 0;
        }
    }
    void begin(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
    {
        in_ = in; in_end_ = in + in_len; bb_ = 0; bc_ = 0; over_ = 0;
        out0_ = o_ = out; out_end_ = out + out_len; last_ = false;
    }
    // from a compressed block whose tables are ready (advance() == 1) to the end of the stream
    bool finish_alone()
    {
        for (;;) {
            codes_fast();
            if (!codes_careful()) return false;
            const int st = advance();
            if (st != 1) return st == 2;
        }
    }

public:
    // the raw deflate stream in[0, in_len) must inflate to exactly out_len bytes at out
    bool run(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
    {
        begin(in, in_len, out, out_len);
        const int st = advance();
        return st == 1 ? finish_alone() : st == 2;
    }
    // `n` independent streams, N of them side by side at any time (see the header comment): ok[k] as run() would return
    // for stream k.  lanes: N decoders of the calling thread.  A lane whose stream has ended takes the next one.
    struct Job { const uint8_t *in; size_t in_len; uint8_t *out; size_t out_len; };
    template <int N>
    static void run_many(FastInflate *lanes, const Job *jobs, int n, bool *ok)
    {
        FastInflate *ln[N];
        int job_of[N];
        int next = 0, active = 0;
        // the next stream with a compressed block into lane x (streams that end before one -- empty, stored only,
        // malformed at once -- are settled here); false when no stream is left
        auto load = [&](int x) -> bool {
            while (next < n) {
                const int k = next++;
                if (!jobs[k].out_len) { ok[k] = true; continue; }
                lanes[x].begin(jobs[k].in, jobs[k].in_len, jobs[k].out, jobs[k].out_len);
                const int st = lanes[x].advance();
                if (st == 1) { job_of[x] = k; return true; }
                ok[k] = st == 2;
            }
            return false;
        };
        for (int x = 0; x < N; x++) {
            ln[x] = &lanes[x];
            job_of[x] = -1;
            if (load(x)) active++;
        }
        while (active == N) {
            int who = codes_fast_n<N>(ln);
            if (who < 0) {
                for (who = 0; who < N - 1 && ln[who]->in_fast_region(); who++) {}
            }
            // the stream that cannot go on in the fast region (the end of a deflate block, the last bytes of its buffers,
            // a symbol the fast loop does not take) finishes its block on its own and takes the next one; the others are
            // simply where the loop left them and rejoin as they are
            FastInflate &x = *ln[who];
            const int st = x.codes_careful() ? x.advance() : 0;
            if (st != 1) {
                ok[job_of[who]] = st == 2;
                job_of[who] = -1;
                if (!load(who)) active--;
            }
        }
        for (int x = 0; x < N; x++)
            if (job_of[x] >= 0) ok[job_of[x]] = ln[x]->finish_alone();
    }

private:
    bool in_fast_region() const { return (size_t)(in_end_ - in_) >= 16 && (size_t)(out_end_ - o_) >= 320; }
};

inline bool inflate_raw_fast(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
    static thread_local FastInflate fi;
    return fi.run(in, in_len, out, out_len);
}

// n independent raw deflate streams by one thread, NPORE_INFLATE_LANES of them side by side at any time
#if !defined(NPORE_INFLATE_LANES)
#define NPORE_INFLATE_LANES 2
#endif
inline void inflate_raw_fast_many(const FastInflate::Job *jobs, int n, bool *ok)
{
    static thread_local FastInflate lanes[NPORE_INFLATE_LANES];
    FastInflate::run_many<NPORE_INFLATE_LANES>(lanes, jobs, n, ok);
}

}  // namespace npore
