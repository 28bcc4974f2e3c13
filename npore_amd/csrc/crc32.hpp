// crc32.hpp -- the CRC-32 of a BGZF member's inflated bytes (RFC 1952; what htslib / pysam verify when the reference
// reads a BAM: src/bam.pyx:18-47 through pysam.AlignmentFile), fast enough to stay on by default: folding by carry-less
// multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ", Intel 2009: four 128-bit
// accumulators folded across 64 bytes per step, then to one, then Barrett reduction) where the CPU has PCLMULQDQ, zlib's
// table-driven crc32 otherwise and for the bytes around the 16-byte grid.  24 KB per 10 kb read: ~2 us on one core against
// ~35 us for inflating them.  tests/test_host_logic.py: equal to zlib's on random lengths, offsets and running values.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace npore {

#if defined(__x86_64__)
#define NPORE_CRC_TARGET __attribute__((target("pclmul,sse4.1")))
// (helpers of the folding loop as functions of their own: a lambda does not inherit its function's target attribute)
NPORE_CRC_TARGET inline __m128i crc32_ld(const uint8_t *q) { return _mm_loadu_si128(reinterpret_cast<const __m128i *>(q)); }
NPORE_CRC_TARGET inline __m128i crc32_fold(__m128i acc, __m128i k, __m128i next)
{
    const __m128i lo = _mm_clmulepi64_si128(acc, k, 0x00), hi = _mm_clmulepi64_si128(acc, k, 0x11);
    return _mm_xor_si128(_mm_xor_si128(lo, hi), next);
}
// `state` is the register value (the running CRC inverted); n >= 64 and n % 16 == 0
NPORE_CRC_TARGET inline uint32_t crc32_fold_pclmul(const uint8_t *p, size_t n, uint32_t state)
{
    // x^(k) mod P for the bit-reflected gzip polynomial, as the paper derives them: 512+64 / 512 bits apart (four
    // accumulators), 128+64 / 128 (one), 96; then P' and mu for the reduction
    const __m128i k_512 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);
    const __m128i k_128 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
    const __m128i k_96 = _mm_set_epi64x(0, 0x0163cd6124);
    const __m128i k_red = _mm_set_epi64x(0x01f7011641, 0x01db710641);
    const __m128i low32s = _mm_setr_epi32(~0, 0, ~0, 0);
    const auto ld = crc32_ld;
    const auto fold = crc32_fold;
    __m128i a = _mm_xor_si128(ld(p), _mm_cvtsi32_si128((int)state)), b = ld(p + 16), c = ld(p + 32), d = ld(p + 48);
    p += 64;
    n -= 64;
    for (; n >= 64; p += 64, n -= 64) {
        a = fold(a, k_512, ld(p));
        b = fold(b, k_512, ld(p + 16));
        c = fold(c, k_512, ld(p + 32));
        d = fold(d, k_512, ld(p + 48));
    }
    a = fold(a, k_128, b);
    a = fold(a, k_128, c);
    a = fold(a, k_128, d);
    for (; n >= 16; p += 16, n -= 16) a = fold(a, k_128, ld(p));
    // 128 -> 64 bits
    __m128i t = _mm_clmulepi64_si128(a, k_128, 0x10);
    a = _mm_xor_si128(_mm_srli_si128(a, 8), t);
    t = _mm_srli_si128(a, 4);
    a = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(a, low32s), k_96, 0x00), t);
    // Barrett
    t = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(a, low32s), k_red, 0x10), low32s);
    a = _mm_xor_si128(a, _mm_clmulepi64_si128(t, k_red, 0x00));
    return (uint32_t)_mm_extract_epi32(a, 1);
}
inline bool cpu_has_pclmul()
{
    static const bool yes = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return yes;
}
#endif

// crc32(crc, p, n) of zlib: `crc` the value so far (0 at the start)
inline uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t n)
{
#if defined(__x86_64__)
    if (n >= 64 && cpu_has_pclmul()) {
        const size_t body = n & ~(size_t)15;
        crc = ~crc32_fold_pclmul(p, body, ~crc);
        p += body;
        n -= body;
    }
#endif
    while (n) {                                      // (zlib takes 32-bit lengths)
        const size_t k = n < ((size_t)1 << 30) ? n : ((size_t)1 << 30);
        crc = (uint32_t)::crc32(crc, p, (uInt)k);
        p += k;
        n -= k;
    }
    return crc;
}

}  // namespace npore
