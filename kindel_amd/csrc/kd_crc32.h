// kd_crc32.h -- CRC-32 (the gzip polynomial, reflected 0xEDB88320) of a buffer, for the BGZF block trailers (host code).
// samtools / htslib -- what the reference's parse_bam reads through (kindel.py:131-153) -- refuse a block whose CRC does not
// match; a reader that skips the check piles up a corrupted block silently.  zlib's table-driven crc32 would cost
// more than the inflater it guards (~1.3 GB/s per core against the ~1 GB/s per core the own DEFLATE decoder delivers), so the
// bulk is folded 64 bytes at a time with carry-less multiplies (Gopal et al., "Fast CRC Computation for Generic Polynomials
// Using PCLMULQDQ Instruction", Intel 2009; constants for the reflected CRC-32 as published there), ~15-20 GB/s per core; the
// last < 16 bytes -- and everything on a CPU without PCLMULQDQ -- go through zlib.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace kdz {

#if defined(__x86_64__)
// raw (non-inverted) CRC state in, raw state out; len >= 64 and a multiple of 16
__attribute__((target("pclmul,sse4.1"))) static inline uint32_t crc32_fold(const uint8_t *buf, size_t len, uint32_t crc) {
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596LL, 0x0154442bd4LL);   // x^(4*128+32), x^(4*128-32) mod P (reflected)
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009eLL, 0x01751997d0LL);   // x^(128+32), x^(128-32) mod P
    const __m128i k5k0 = _mm_set_epi64x(0x0000000000LL, 0x0163cd6124LL);   // x^64 mod P
    const __m128i poly = _mm_set_epi64x(0x01f7011641LL, 0x01db710641LL);   // mu, P
    __m128i x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00)), x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20)), x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    buf += 64; len -= 64;
    while (len >= 64) {      // four independent 128-bit lanes folded over 64 bytes each
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11); x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i *)(buf + 0x00)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i *)(buf + 0x10)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i *)(buf + 0x20)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i *)(buf + 0x30)));
        buf += 64; len -= 64;
    }
    // the four lanes into one
    __m128i a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x2), a);
    a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x3), a);
    a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x4), a);
    while (len >= 16) {
        a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), _mm_loadu_si128((const __m128i *)buf)), a);
        buf += 16; len -= 16;
    }
    // 128 -> 64 bits
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    x2 = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5k0, 0x00), x2);
    // Barrett reduction to 32 bits
    x2 = _mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10);
    x2 = _mm_clmulepi64_si128(_mm_and_si128(x2, mask32), poly, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
static inline bool crc32_have_clmul() {
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return have;
}
#endif

// zlib's convention: crc32_buf(p, n) == crc32(crc32(0, NULL, 0), p, n)
static inline uint32_t crc32_buf(const uint8_t *p, size_t n) {
    uint32_t crc = 0;
#if defined(__x86_64__)
    if (n >= 64 && crc32_have_clmul()) {
        const size_t bulk = n & ~(size_t)15;
        crc = ~crc32_fold(p, bulk, ~crc);
        p += bulk; n -= bulk;
    }
#endif
    while (n) {    // (zlib takes a 32-bit length)
        const size_t step = n < ((size_t)1 << 30) ? n : ((size_t)1 << 30);
        crc = (uint32_t)crc32(crc, p, (uInt)step);
        p += step; n -= step;
    }
    return crc;
}

}  // namespace kdz
