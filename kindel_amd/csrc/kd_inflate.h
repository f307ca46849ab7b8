// kd_inflate.h -- raw DEFLATE (RFC 1951) decoder for BGZF blocks: the whole input and the exact output size are known up
// front (a BGZF block is at most 64 KiB each way), so there is no streaming state, no window copy and no per-call
// allocation.  Host side of the ingest row: zlib's inflate was 79 % of the decoder's CPU time.
//   * 64-bit bit buffer refilled with one unaligned 8-byte load (branch-free),
//   * one table look-up per symbol: 11-bit primary table for literals / lengths, 8-bit for distances, second-level
//     tables for longer codes; an entry carries the literal / base value, the number of extra bits and the code length,
//   * up to three literals per refill, the table look-up issued in front of the refill where the buffer still holds an index'
//     worth of bits (a refill only adds bits above the valid ones: the load's latency runs beside it), matches copied 8
//     bytes at a time -- the first 16 unconditionally: 87 % of a BAM block's matches are 3 - 4 bytes long -- (byte pattern
//     replicated for distances < 8); measured on BAM records with Phred-like qualities, one 2.1 GHz core: 285 -> 320 MB/s inflated,
//   * a fast loop while both buffers have slack, the same decode step with exact bounds for the tail.
// Every malformed stream (over-subscribed or incomplete code used, distance before the start of the output, output size
// other than announced, input overrun) returns false; nothing is ever written outside [out, out + out_len).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace kdz {

// One 32-bit word per table entry:
//   bits 0..3   code bits to consume at this level (1..15; second-level entries: code length - primary bits)
//   bits 4..7   extra bits of a length / distance symbol (0..13); K_SUB: index bits of the second-level table
//   bits 8..12  kind
//   bits 16..31 literal value / length base / distance base / K_SUB: offset of the second-level table
enum : uint32_t { K_LIT = 0x0100u, K_LEN = 0x0200u, K_EOB = 0x0400u, K_SUB = 0x0800u, K_BAD = 0x1000u };
static inline uint32_t mk(uint32_t kind, uint32_t nbits, uint32_t extra, uint32_t value) {
    return (value << 16) | kind | (extra << 4) | nbits;
}

constexpr int LIT_BITS = 11, DIST_BITS = 10;
constexpr int LIT_TABLE = (1 << LIT_BITS) + 288 * 16, DIST_TABLE = (1 << DIST_BITS) + 32 * 32;

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// the low `len` (<= 15) bits of code, reversed
static inline uint32_t rev_bits(uint32_t code, int len) {
    uint32_t x = code;
    x = ((x & 0x5555u) << 1) | ((x >> 1) & 0x5555u);
    x = ((x & 0x3333u) << 2) | ((x >> 2) & 0x3333u);
    x = ((x & 0x0f0fu) << 4) | ((x >> 4) & 0x0f0fu);
    x = ((x & 0x00ffu) << 8) | ((x >> 8) & 0x00ffu);
    return x >> (16 - len);
}

// Canonical Huffman code of `n` symbols with lengths lens[] (0 = unused) -> look-up table indexed by the next bits of the
// stream (LSB first).  entry_of(sym) gives kind / extra / value; the code length is filled in here.  Returns false for an
// over-subscribed code and -- zlib's rule (inflate_table: "incomplete set"), so that this reader refuses what htslib refuses --
// for an INCOMPLETE one, unless it has no symbol at all or is a literal / length or distance code of one single 1-bit word (then
// the unused half stays K_BAD: an error only if the stream uses it); the code-length code (`cl_code`) must be complete.
template <class EntryOf>
static bool build_table(const uint8_t *lens, int n, int primary_bits, uint32_t *table, int table_cap, EntryOf entry_of, bool cl_code = false) {
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return false; }
    {
        int mx = 15;
        while (mx > 0 && !count[mx]) mx--;
        if (left > 0 && mx != 0 && (cl_code || mx != 1)) return false;
    }
    uint32_t next_code[16];
    { uint32_t code = 0; for (int l = 1; l <= 15; l++) { code = (code + (uint32_t)count[l - 1]) << 1; next_code[l] = code; } }
    const int psize = 1 << primary_bits;
    int maxlen = 15;
    while (maxlen > 0 && !count[maxlen]) maxlen--;
    // a complete code whose longest word fits the primary index overwrites every primary entry: nothing to prepare
    const bool simple = left == 0 && maxlen <= primary_bits;
    uint32_t codes[288];
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (l) codes[s] = rev_bits(next_code[l]++, l);
    }
    uint8_t sub_bits[1 << LIT_BITS];
    if (!simple) {
        for (int i = 0; i < psize; i++) table[i] = K_BAD | 1u;
        if (maxlen > primary_bits) {
            // second-level tables: the longest code under each primary prefix decides the table's width
            memset(sub_bits, 0, (size_t)psize);
            for (int s = 0; s < n; s++) {
                const int l = lens[s];
                if (l > primary_bits) {
                    const uint32_t p = codes[s] & (uint32_t)(psize - 1);
                    if (l - primary_bits > sub_bits[p]) sub_bits[p] = (uint8_t)(l - primary_bits);
                }
            }
            int used = psize;
            for (int p = 0; p < psize; p++) {
                if (!sub_bits[p]) continue;
                const int sz = 1 << sub_bits[p];
                if (used + sz > table_cap) return false;
                table[p] = mk(K_SUB, (uint32_t)primary_bits, sub_bits[p], (uint32_t)used);
                for (int i = 0; i < sz; i++) table[used + i] = K_BAD | 1u;
                used += sz;
            }
        }
    }
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = codes[s];
        if (l <= primary_bits) {
            const uint32_t e = entry_of(s) | (uint32_t)l;
            for (uint32_t i = r; i < (uint32_t)psize; i += 1u << l) table[i] = e;
        } else {
            const uint32_t p = r & (uint32_t)(psize - 1);
            const uint32_t base = table[p] >> 16, sb = sub_bits[p];
            const uint32_t e = entry_of(s) | (uint32_t)(l - primary_bits);
            for (uint32_t i = r >> primary_bits; i < (1u << sb); i += 1u << (l - primary_bits)) table[base + i] = e;
        }
    }
    return true;
}

static inline uint32_t litlen_entry(int s) {
    if (s < 256) return mk(K_LIT, 0, 0, (uint32_t)s);
    if (s == 256) return mk(K_EOB, 0, 0, 0);
    if (s > 285) return K_BAD;
    return mk(K_LEN, 0, LEN_EXTRA[s - 257], LEN_BASE[s - 257]);
}
static inline uint32_t dist_entry(int s) {
    if (s > 29) return K_BAD;
    return mk(K_LEN, 0, DIST_EXTRA[s], DIST_BASE[s]);
}

struct Decoder {
    uint32_t lit[LIT_TABLE], dist[DIST_TABLE];
    const uint8_t *in, *in_end;
    uint8_t *out, *out0, *out_end;
    uint64_t bb;     // bit buffer, LSB = next bit
    int bc;          // valid bits in bb
    size_t over;     // bytes of zero padding fed past the end of the input (the stream is bad if any of them was consumed)

    // at least 56 valid bits afterwards (while input lasts); branch-free when 8 bytes can be read
    inline void refill() {
        if (in_end - in >= 8) {
            uint64_t w;
            memcpy(&w, in, 8);
            bb |= w << bc;
            const int adv = (63 - bc) >> 3;
            in += adv;
            bc += adv * 8;
        } else {
            while (bc <= 56) {
                if (in < in_end) bb |= (uint64_t)*in++ << bc; else over++;
                bc += 8;
            }
        }
    }
    inline uint32_t take(int n) { const uint32_t v = (uint32_t)(bb & ((1ull << n) - 1)); bb >>= n; bc -= n; return v; }
    // true if more bits were consumed than the input holds
    inline bool overrun() const { return over * 8 > (size_t)bc; }

    bool stored() {
        const int drop = bc & 7;      // to the byte boundary
        bb >>= drop; bc -= drop;
        refill();
        const uint32_t len = take(16), nlen = take(16);
        if ((len ^ nlen) != 0xffffu || overrun()) return false;
        if ((size_t)(out_end - out) < len) return false;
        uint32_t left = len;
        while (left && bc >= 8) { *out++ = (uint8_t)take(8); left--; }   // bytes already in the bit buffer
        if (left) {
            if (bc != 0) return false;
            if ((size_t)(in_end - in) < left) return false;
            memcpy(out, in, left);
            in += left; out += left;
            bb = 0;     // (the bit buffer may hold look-ahead bits of the bytes just copied)
        }
        return true;
    }

    bool dynamic_tables() {
        refill();
        const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
        if (hlit > 286 || hdist > 30) return false;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; i++) { if (bc < 3) refill(); cl[order[i]] = (uint8_t)take(3); }
        uint32_t clt[128 + 19 * 2];
        if (!build_table(cl, 19, 7, clt, 128 + 19 * 2, [](int s) { return mk(K_LIT, 0, 0, (uint32_t)s); }, true)) return false;
        uint8_t lens[286 + 30 + 138];
        int n = 0;
        while (n < hlit + hdist) {
            refill();
            const uint32_t e = clt[bb & 127u];
            if (e & K_BAD) return false;
            take((int)(e & 15u));
            const uint32_t s = e >> 16;
            if (s < 16) { lens[n++] = (uint8_t)s; continue; }
            int rep; uint8_t v = 0;
            if (s == 16) { if (!n) return false; v = lens[n - 1]; rep = 3 + (int)take(2); }
            else if (s == 17) rep = 3 + (int)take(3);
            else rep = 11 + (int)take(7);
            if (n + rep > hlit + hdist) return false;
            memset(lens + n, v, (size_t)rep);
            n += rep;
        }
        if (overrun() || lens[256] == 0) return false;
        if (!build_table(lens, hlit, LIT_BITS, lit, LIT_TABLE, litlen_entry)) return false;
        return build_table(lens + hlit, hdist, DIST_BITS, dist, DIST_TABLE, dist_entry);
    }

    void fixed_tables() {
        uint8_t lens[288 + 32];
        for (int i = 0; i < 144; i++) lens[i] = 8;
        for (int i = 144; i < 256; i++) lens[i] = 9;
        for (int i = 256; i < 280; i++) lens[i] = 7;
        for (int i = 280; i < 288; i++) lens[i] = 8;
        for (int i = 0; i < 32; i++) lens[288 + i] = 5;
        build_table(lens, 288, LIT_BITS, lit, LIT_TABLE, litlen_entry);
        build_table(lens + 288, 32, DIST_BITS, dist, DIST_TABLE, dist_entry);
    }

    inline uint32_t lookup_lit() {
        uint32_t e = lit[bb & ((1u << LIT_BITS) - 1)];
        if (e & K_SUB) {
            bb >>= LIT_BITS; bc -= LIT_BITS;
            e = lit[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 4) & 15u)) - 1))];
        }
        return e;
    }
    inline uint32_t lookup_dist() {
        uint32_t e = dist[bb & ((1u << DIST_BITS) - 1)];
        if (e & K_SUB) {
            bb >>= DIST_BITS; bc -= DIST_BITS;
            e = dist[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 4) & 15u)) - 1))];
        }
        return e;
    }

    // one compressed block's symbols; FAST: both buffers have slack for a whole step (checked by the caller's loop)
    bool block() {
        for (;;) {
            // fast steps: >= 16 input bytes for the refills, >= 3 literals + 258 + 16 bytes of output slack (a match's first 16 bytes
            // are copied whatever its length)
            while (in_end - in >= 16 && out_end - out >= 3 + 258 + 16) {
                // the look-up first when the buffer still holds an index' worth of bits (a refill only adds bits ABOVE the valid ones):
                // the load's latency runs beside the refill instead of behind it
                uint32_t e;
                if (bc >= LIT_BITS) { e = lit[bb & ((1u << LIT_BITS) - 1)]; refill(); }
                else { refill(); e = lit[bb & ((1u << LIT_BITS) - 1)]; }
                bool fresh = true;
                if (e & K_LIT) {
                    fresh = false;
                    bb >>= (e & 15u); bc -= (int)(e & 15u);
                    *out++ = (uint8_t)(e >> 16);
                    e = lit[bb & ((1u << LIT_BITS) - 1)];
                    if (e & K_LIT) {
                        bb >>= (e & 15u); bc -= (int)(e & 15u);
                        *out++ = (uint8_t)(e >> 16);
                        e = lit[bb & ((1u << LIT_BITS) - 1)];
                        if (e & K_LIT) {
                            bb >>= (e & 15u); bc -= (int)(e & 15u);
                            *out++ = (uint8_t)(e >> 16);
                            continue;
                        }
                    }
                }
                if (e & K_SUB) {
                    bb >>= LIT_BITS; bc -= LIT_BITS;
                    e = lit[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 4) & 15u)) - 1))];
                    if (e & K_LIT) {
                        bb >>= (e & 15u); bc -= (int)(e & 15u);
                        *out++ = (uint8_t)(e >> 16);
                        continue;
                    }
                    fresh = false;     // (a long length code: be generous with the bits)
                }
                bb >>= (e & 15u); bc -= (int)(e & 15u);
                if (!(e & K_LEN)) {
                    if (e & K_EOB) return true;
                    return false;
                }
                const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << ((e >> 4) & 15u)) - 1));
                { const int x = (int)((e >> 4) & 15u); bb >>= x; bc -= x; }
                if (!fresh && bc < 28) refill();      // (distance code 15 + 13 extra bits)
                uint32_t d = lookup_dist();
                bb >>= (d & 15u); bc -= (int)(d & 15u);
                if (!(d & K_LEN)) return false;
                const uint32_t dd = (d >> 16) + (uint32_t)(bb & ((1u << ((d >> 4) & 15u)) - 1));
                { const int x = (int)((d >> 4) & 15u); bb >>= x; bc -= x; }
                if (dd > (size_t)(out - out0)) return false;
                const uint8_t *src = out - dd;
                uint8_t *dst = out;
                out += len;
                // (the copies may write up to 7 bytes past out: inside the slack the loop condition guarantees)
                if (dd >= 8) {
                    { uint64_t w; memcpy(&w, src, 8); memcpy(dst, &w, 8); }
                    { uint64_t w; memcpy(&w, src + 8, 8); memcpy(dst + 8, &w, 8); }
                    if (len > 16) { src += 16; dst += 16; do { uint64_t w; memcpy(&w, src, 8); memcpy(dst, &w, 8); src += 8; dst += 8; } while (dst < out); }
                } else if (dd == 1) {
                    const uint64_t w = 0x0101010101010101ull * (uint64_t)*src;
                    do { memcpy(dst, &w, 8); dst += 8; } while (dst < out);
                } else {
                    do { *dst++ = *src++; } while (dst < out);
                }
            }
            // one exact step (the tail of the buffers)
            refill();
            uint32_t e = lookup_lit();
            bb >>= (e & 15u); bc -= (int)(e & 15u);
            if (overrun()) return false;
            if (e & K_LIT) {
                if (out >= out_end) return false;
                *out++ = (uint8_t)(e >> 16);
                continue;
            }
            if (!(e & K_LEN)) return (e & K_EOB) != 0;
            uint32_t len = (e >> 16) + take((int)((e >> 4) & 15u));
            if (bc < 32) refill();
            uint32_t d = lookup_dist();
            bb >>= (d & 15u); bc -= (int)(d & 15u);
            if (!(d & K_LEN)) return false;
            const uint32_t dd = (d >> 16) + take((int)((d >> 4) & 15u));
            if (overrun() || dd > (size_t)(out - out0) || len > (size_t)(out_end - out)) return false;
            const uint8_t *src = out - dd;
            for (uint32_t k = 0; k < len; k++) out[k] = src[k];
            out += len;
        }
    }

    bool run() {
        bb = 0; bc = 0; over = 0;
        for (;;) {
            refill();
            const uint32_t final_block = take(1), type = take(2);
            if (type == 0) { if (!stored()) return false; }
            else if (type == 1) { fixed_tables(); if (!block()) return false; }
            else if (type == 2) { if (!dynamic_tables() || !block()) return false; }
            else return false;
            if (overrun()) return false;
            if (final_block) break;
        }
        return out == out_end;
    }
};

// the whole raw DEFLATE stream in[0, in_len) must decode to exactly out_len bytes
static inline bool inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
    Decoder d;
    d.in = in; d.in_end = in + in_len;
    d.out = out; d.out0 = out; d.out_end = out + out_len;
    return d.run();
}

}  // namespace kdz
