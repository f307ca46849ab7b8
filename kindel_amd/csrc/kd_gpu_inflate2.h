// kd_gpu_inflate2.h -- round 6: raw DEFLATE (RFC 1951) of BGZF blocks on the GPU in TWO passes, 64 blocks per wavefront.
// (the row behind parse_bam's record iteration, kindel.py:131-153.)
//
// kd_gpu_inflate.h (rounds 3 - 5) gives a block a whole wavefront whose 64 lanes all walk the SAME symbol stream on the scalar unit:
// ~100 instructions per symbol, one symbol at a time per wavefront -- 85 % of the device-side ingest's 0.29 s on the C3-sized file.  The
// Huffman walk is serial per block but the blocks are independent, and what makes a lane-per-block decoder hard on a GPU is only the
// LZ77 window (a match reads bytes the same lane stored a moment ago: a memory round trip per match).  So the work is cut in two:
//
//   pass 1  k_inflate_tokens   ONE LANE PER BLOCK.  Bit reader, Huffman decode and block headers per lane; literals are stored at
//           their FINAL place in the output (a lane knows its output cursor), matches are only RECORDED -- a 32-bit token (literals
//           since the last token, length, distance) in the block's token list -- and the cursor moves on.  No lane ever reads what it
//           wrote.  The canonical code is decoded WITHOUT a per-lane look-up table (64 lanes x 2 KB would be the CU's whole LDS for
//           one wavefront): the 15 left-justified code limits of a code live in registers, the code length is the number of limits
//           the next 15 bits reach (a fixed chain of compare + conditional add: the same instructions for every lane, whatever its
//           code), and the symbol comes from ONE read of the lane's sorted-symbol array in LDS (316 entries x u16 per lane = 40 KB per
//           wavefront: four wavefronts per CU, one per SIMD).  Code lengths while a header is read share those entries (high nibble).
//   pass 2  k_inflate_resolve  ONE WAVEFRONT PER BLOCK.  64 tokens at a time: a prefix sum gives every match its place; a match is
//           READY when no unfinished earlier match of the batch writes into its source (two binary searches over the batch's sorted
//           output ranges + the pending mask); the ready matches' bytes are copied by all 64 lanes, byte-balanced (binary search over
//           the ready lengths' prefix sums); stores are drained, the next round looks again.  The first pending match is always ready.
//
// Same contract as k_gpu_inflate: status[b] = GI_OK or what went wrong (zlib's strictness: over-subscribed / incomplete codes refused),
// nothing is written outside [out_off, out_off + out_len) and outside the block's token region.  Checked against zlib on the CPU
// emulator (tests/test_gpu_inflate_proto.py) with the same cases as the one-pass kernel.
#pragma once
#include <stdint.h>
#ifndef KD_EMU
#include "kd_common.h"      // the wavefront helpers (kd_ballot, kd_wave_scan_add, kd_readlane); the emulator's come from tests/emu/hip_emu.h
#endif
#include "kd_gpu_inflate.h"

// between two rounds of pass 2: the round's stores must have reached the L2 before the next round's loads (which bypass the L1) ask for them
#ifndef KD_EMU
#define GI2_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define GI2_NO_IF_CONVERT() asm volatile("" ::: "memory")      // an asm statement cannot be executed speculatively: the block around it stays a branch
#else
#define GI2_DRAIN()
#define GI2_NO_IF_CONVERT()
#endif

struct __attribute__((packed, aligned(1))) KdChunk16 { uint32_t x, y, z, w; };      // 16 bytes at any address (one global_load_dwordx4)
#define GI2_LIT 286u       // sorted-symbol entries of the literal / length code
#define GI2_SLOTS 316u     // + 30 of the distance code
// Token (u32): bits 0-8 literals since the last token (0 .. 510) | bits 9-16 match length - 3 | bits 17-31 distance - 1;
// bits 0-8 == 511: no match, bits 9-31 = that many literals to skip (a literal run of more than 510 bytes).
#define GI2_SKIP 511u
// the token region of block b starts at  out_off / 3 + out_off / 256 + 9 b  (tokens): consecutive regions are at least
// out_len / 3 + out_len / 256 + 9 apart (floor(x + y) >= floor(x) + floor(y)), and a block emits at most one token per 3 bytes of matches
// plus one per 511 bytes of literals
__host__ __device__ __forceinline__ unsigned long long gi2_tok_off(unsigned long long out_off, unsigned long long b) {
    return out_off / 3ull + out_off / 256ull + 9ull * b;
}

#ifndef KD_EMU
__device__ __forceinline__ uint32_t gi2_brev(uint32_t v) { return __brev(v); }
#else
static inline uint32_t gi2_brev(uint32_t v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    return (v >> 16) | (v << 16);
}
#endif

// A canonical Huffman code of at most MAXL bits, for a decoder that holds it in registers.  With the next MAXL bits of the stream
// left-justified in w (first bit read = most significant):  lim[l-1] = left-justified end of the codes of length <= l  (non-
// decreasing in l), so  length = 1 + #{l < MAXL : w >= lim[l-1]},  and with  pk[l-1] = count[l] << 20 | 1 << 16 | lim[l-1] - lim[l-2]
// the sum of the pk of those l holds the length, the number of shorter codes (= the symbol's rank offset) and lim[length-2] at once.
template <int MAXL>
struct Gi2Code {
    uint32_t lim[MAXL], pk[MAXL];
};
// -> the code's length and the rank of its symbol among the coded symbols ordered by (length, value); false: no such code word
template <int MAXL>
__device__ __forceinline__ bool gi2_decode(const Gi2Code<MAXL> &c, uint32_t w, uint32_t &len, uint32_t &rank) {
    uint32_t acc = 0;
#pragma unroll
    for (int l = 0; l < MAXL - 1; l++) acc += (w >= c.lim[l]) ? c.pk[l] : 0u;
    len = 1u + ((acc >> 16) & 15u);
    rank = (acc >> 20) + ((w - (acc & 0xffffu)) >> ((uint32_t)MAXL - len));
    return w < c.lim[MAXL - 1];
}
// lim / pk from the counts per length (cnt[l-1] = codes of length l).  false: over-subscribed, or incomplete where zlib refuses that
// (inflate_table: an incomplete set is only accepted for a literal / length or distance code with a single one-bit code)
template <int MAXL>
__device__ __forceinline__ bool gi2_make_code(const uint32_t (&cnt)[MAXL], bool cl_code, Gi2Code<MAXL> &c) {
    int left = 1, mx = 0;
    bool ok = true;
    uint32_t code = 0, prev_lim = 0;
#pragma unroll
    for (int l = 1; l <= MAXL; l++) {
        left = (left << 1) - (int)cnt[l - 1];
        if (left < 0) ok = false;
        if (cnt[l - 1]) mx = l;
        const uint32_t end = code + cnt[l - 1];
        const uint32_t lim = end << (MAXL - l);
        c.lim[l - 1] = lim;
        c.pk[l - 1] = (cnt[l - 1] << 20) | (1u << 16) | ((lim - prev_lim) & 0xffffu);
        prev_lim = lim;
        code = end << 1;
    }
    if (left > 0 && mx != 0 && (cl_code || mx != 1)) ok = false;
    return ok;
}

// counters of the lengths 1 .. 15 (or running offsets), 9 bits each, in registers, dynamically indexed
struct Gi2Cnt15 {
    unsigned long long a = 0, b = 0;   // lengths 1-7, 8-14
    uint32_t c = 0;                    // length 15
    __device__ __forceinline__ void add(uint32_t l, uint32_t x) {
        if (l - 1u < 7u) a += (unsigned long long)x << (9u * (l - 1u));
        else if (l - 8u < 7u) b += (unsigned long long)x << (9u * (l - 8u));
        else if (l == 15u) c += x;
    }
    __device__ __forceinline__ uint32_t get(uint32_t l) const {
        if (l - 1u < 7u) return (uint32_t)(a >> (9u * (l - 1u))) & 511u;
        if (l - 8u < 7u) return (uint32_t)(b >> (9u * (l - 8u))) & 511u;
        return l == 15u ? c : 0u;
    }
};

// ---------------------------------------------------------------------------------------------------------------------------------
// pass 1: one LANE per block.  comp: the file (readable 32 bytes past every block's input); out: the inflated bytes of all blocks (the
// literals are written here); tokens: the match tokens, block b's from gi2_tok_off(out_off, b0 + b); n_tok[b] tokens; status[b].
// blocks / n_tok / status point at the launch's first block, b0 = that block's index in the file (the token regions are laid out by it).
// work: a zeroed counter.  The launch holds what the chip can keep resident (four wavefronts per CU); a lane that has finished its block
// takes the next one from the counter -- a file of 68 568 blocks is 1 072 wavefronts' worth, 48 more than the 1 024 that fit: without
// this the launch took two rounds for 5 % more work (measured: 54 ms against 26 ms for the same file at a quarter of the depth).
//
// The bit reader (round 6, second version): a 64-bit window (lo, hi) of the stream and a bit offset, read with ONE v_alignbit_b32 per
// code word (the code and its extra bits -- at most 28 bits -- come out of the same 32); the input arrives 16 bytes at a time into
// registers, the NEXT 16 requested one buffer ahead.  (The first version kept a 64-bit shift register and asked for the next dword inside
// the refill that consumed the last one: the compiler had to wait for the load on the spot -- a memory round trip every fourth
// symbol -- and 64-bit shifts are quarter rate.)
// ---------------------------------------------------------------------------------------------------------------------------------
#ifndef KD_EMU
__device__ __forceinline__ uint32_t gi2_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
#else
static inline uint32_t gi2_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31u)); }
#endif
struct Gi2In {      // one lane's view of its block's compressed bytes
    const uint8_t *in;
    uint32_t in_cap;                // the last byte offset a 16-byte load may start at (the block's input + 16)
    uint32_t q0, q1, q2, q3;        // the 16 bytes the window is fed from
    uint32_t n0, n1, n2, n3;        // the 16 bytes behind them (requested when q was filled)
    uint32_t qi, qat;               // next dword of q; byte offset of n in the input
    uint32_t lo, hi, bo, wpos;      // the window: stream bits from byte wpos on, bo of them consumed
    // 16 bytes of the input from byte `at` on -- ONE load whatever `at` is (a guarded second path made the compiler wait for the
    // load where it is issued): past the block's end the stream reads what lies behind it (the next block's header; the file buffer
    // is readable 32 bytes past every block), and a runaway cursor is held at the block's end + 16 -- what is decoded from there on
    // is wrong either way and ends in GI_E_INPUT / GI_E_SIZE
    __device__ __forceinline__ void ld16(uint32_t at, uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d) const {
        const KdChunk16 v = *reinterpret_cast<const KdChunk16 *>(in + (at < in_cap ? at : in_cap));
        a = v.x; b = v.y; c = v.z; d = v.w;
    }
    __device__ __forceinline__ void start(uint32_t at) {      // the stream from byte `at` of the input on
        ld16(at, q0, q1, q2, q3);
        ld16(at + 16u, n0, n1, n2, n3);
        qat = at + 16u;                                        // (n is requested: q's first dword is taken here, not by next_dword)
        lo = q0; hi = q1; qi = 2u; bo = 0u; wpos = at;
    }
    __device__ __forceinline__ uint32_t next_dword() {
        const uint32_t d = qi == 0u ? q0 : qi == 1u ? q1 : qi == 2u ? q2 : q3;
        ++qi;
        // The 16 bytes behind q are REQUESTED when q's first dword is taken and MOVED into q when its last one is: two different
        // moments, so that the request writes n in place (nothing reads n there) and the wait for it sits three dwords -- a dozen
        // symbols -- later.  (Requested in the same branch that moves n into q, the compiler loads into temporaries, waits on the
        // spot and copies: a memory round trip per 16 bytes, and with four wavefronts' 256 input streams the L1 holds none of them.)
        if (qi == 1u) { qat += 16u; ld16(qat, n0, n1, n2, n3); }
        if (qi == 4u) {
            GI2_NO_IF_CONVERT();      // (a real branch: as selects, the moves read n -- and wait for its load -- on every call)
            q0 = n0; q1 = n1; q2 = n2; q3 = n3; qi = 0u;
        }
        return d;
    }
    __device__ __forceinline__ void norm() {                   // bo < 32 afterwards (at most 32 bits are consumed between two calls)
        if (bo >= 32u) { lo = hi; hi = next_dword(); bo -= 32u; wpos += 4u; }
    }
    __device__ __forceinline__ uint32_t peek() const { return gi2_alignbit(hi, lo, bo); }      // the next 32 bits (after norm())
    __device__ __forceinline__ unsigned long long bits_used() const { return 8ull * wpos + bo; }
};

// A workgroup is FOUR wavefronts -- one per SIMD of the CU, placed there by construction -- and takes the CU's whole LDS (4 x 40 448 bytes,
// dynamic).  (As four one-wavefront workgroups the same launch was 1.6 x faster with two and 2.2 x with four wavefronts per CU than with
// one: single-wavefront workgroups do not reliably land on different SIMDs; profiles/r06_inflate_pass1_experiments.txt.)
#define GI2_WG 256u
#define GI2_LDS_BYTES (GI2_SLOTS * GI2_WG * 2u)
__global__ void __launch_bounds__(GI2_WG) k_inflate_tokens(const uint8_t *comp, const GiBlock *blocks, uint32_t n_blocks, uint8_t *out,
                                                           uint32_t *tokens, uint32_t *n_tok, uint32_t *status, uint32_t b0, uint32_t *work) {
    KD_DYN_SHARED(uint16_t, slots_all);
    uint16_t *slots = slots_all + (threadIdx.x >> 6) * (GI2_SLOTS * 64u);      // this wavefront's part
    // entry i of lane l: bits 0-8 i-th coded symbol in (length, value) order | bits 12-15 code length of symbol i
    const uint32_t lane = threadIdx.x & 63u;
#define GI2_SLOT(i) slots[(uint32_t)(i) * 64u + lane]
    // (no wavefront-wide operation, no barrier anywhere in this kernel: every lane is on its own)
    for (uint32_t b = blockIdx.x * GI2_WG + threadIdx.x; b < n_blocks; b = gridDim.x * GI2_WG + atomicAdd(work, 1u)) {
    const GiBlock B = blocks[b];
    uint8_t *dst = out + B.out_off;
    uint32_t *tk = tokens + gi2_tok_off(B.out_off, (unsigned long long)b0 + b);
    const uint32_t tok_cap = (uint32_t)(gi2_tok_off(B.out_off + B.out_len, (unsigned long long)b0 + b + 1ull) - gi2_tok_off(B.out_off, (unsigned long long)b0 + b));
    Gi2In s;
    s.in = comp + B.in_off; s.in_cap = B.in_len + 16u;
    s.start(0u);
    uint32_t pos = 0, run = 0, ntok = 0, err = GI_OK;
    Gi2Code<15> c_lit, c_dist;
    // canonical code over the lengths in the high nibbles of entries [s0, s0 + n): c, and the symbols' ranks into the low bits
    auto build15 = [&](uint32_t s0, uint32_t n, Gi2Code<15> &c) -> bool {
        Gi2Cnt15 k;
        for (uint32_t x = 0; x < n; x++) k.add((uint32_t)GI2_SLOT(s0 + x) >> 12, 1u);
        uint32_t cnt[15];
#pragma unroll
        for (int l = 1; l <= 15; l++) cnt[l - 1] = k.get((uint32_t)l);
        if (!gi2_make_code<15>(cnt, false, c)) return false;
        Gi2Cnt15 o;                                  // first rank of every length
        uint32_t sum = 0;
#pragma unroll
        for (int l = 1; l <= 15; l++) { o.add((uint32_t)l, sum); sum += cnt[l - 1]; }
        for (uint32_t x = 0; x < n; x++) {
            const uint32_t l = (uint32_t)GI2_SLOT(s0 + x) >> 12;
            if (!l) continue;
            const uint32_t r = o.get(l);
            o.add(l, 1u);
            GI2_SLOT(s0 + r) = (uint16_t)((GI2_SLOT(s0 + r) & 0xf000u) | x);
        }
        return true;
    };
    for (bool last = false; !last && err == GI_OK;) {
        s.norm();
        uint32_t p = s.peek();
        last = (p & 1u) != 0;
        const uint32_t type = (p >> 1) & 3u;
        s.bo += 3u;
        if (type == 0) {
            // stored: to the byte boundary, LEN / NLEN, then LEN bytes straight from the input
            s.bo = (s.bo + 7u) & ~7u;
            s.norm();
            p = s.peek();
            s.bo += 32u;
            const uint32_t len = p & 0xffffu, nlen = p >> 16;
            if ((len ^ nlen) != 0xffffu) { err = GI_E_STORED; break; }
            if (pos + len > B.out_len) { err = GI_E_SIZE; break; }
            const uint32_t at = s.wpos + s.bo / 8u;  // the next unread input byte
            if (at > B.in_len || len > B.in_len - at) { err = GI_E_INPUT; break; }
            for (uint32_t i = 0; i < len; i++) dst[pos + i] = s.in[at + i];
            pos += len; run += len;
            s.start(at + len);
            continue;
        }
        if (type == 3) { err = GI_E_BTYPE; break; }
        bool fixed = false;
        if (type == 1) {
            // the fixed code: 288 literal / length symbols (two more than a dynamic code may have: the table borrows the first two entries
            // of the distance part, and the fixed distance code needs none -- 30 five-bit codes in symbol order)
            fixed = true;
            uint32_t cl[15], cd[15];
#pragma unroll
            for (int l = 0; l < 15; l++) { cl[l] = 0; cd[l] = 0; }
            cl[6] = 24; cl[7] = 152; cl[8] = 112; cd[4] = 32;
            (void)gi2_make_code<15>(cl, false, c_lit);
            (void)gi2_make_code<15>(cd, false, c_dist);
            for (uint32_t r = 0; r < 288u; r++)      // ranks: 256 .. 279 (7 bits), 0 .. 143 and 280 .. 287 (8), 144 .. 255 (9)
                GI2_SLOT(r) = (uint16_t)(r < 24u ? 256u + r : r < 168u ? r - 24u : r < 176u ? 280u + (r - 168u) : 144u + (r - 176u));
        } else {
            const uint32_t hlit = ((p >> 3) & 31u) + 257u, hdist = ((p >> 8) & 31u) + 1u, hclen = ((p >> 13) & 15u) + 4u;
            s.bo += 14u;
            if (hlit > 286u || hdist > 30u) { err = GI_E_CODES; break; }
            // the code-length code: 19 symbols of at most 7 bits, entirely in registers
            unsigned long long cll = 0;              // 3 bits per symbol
            {
                const uint32_t ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
#pragma unroll
                for (int i = 0; i < 19; i++) {
                    if ((uint32_t)i < hclen) { s.norm(); cll |= (unsigned long long)(s.peek() & 7u) << (3u * ord[i]); s.bo += 3u; }
                }
            }
            Gi2Code<7> c_cl;
            unsigned long long srt0 = 0, srt1 = 0;   // the coded symbols in (length, value) order, 5 bits each (12 + 7)
            {
                unsigned long long k7 = 0;           // 5 bits per length 0 .. 7
#pragma unroll
                for (int x = 0; x < 19; x++) k7 += 1ull << (5u * ((uint32_t)(cll >> (3u * x)) & 7u));
                uint32_t cnt[7];
#pragma unroll
                for (int l = 1; l <= 7; l++) cnt[l - 1] = (uint32_t)(k7 >> (5u * l)) & 31u;
                if (!gi2_make_code<7>(cnt, true, c_cl)) { err = GI_E_CODES; break; }
                unsigned long long o7 = 0;
                uint32_t sum = 0;
#pragma unroll
                for (int l = 1; l <= 7; l++) { o7 += (unsigned long long)sum << (5u * l); sum += cnt[l - 1]; }
#pragma unroll
                for (int x = 0; x < 19; x++) {
                    const uint32_t l = (uint32_t)(cll >> (3u * x)) & 7u;
                    if (l) {
                        const uint32_t r = (uint32_t)(o7 >> (5u * l)) & 31u;
                        o7 += 1ull << (5u * l);
                        if (r < 12u) srt0 |= (unsigned long long)x << (5u * r); else srt1 |= (unsigned long long)x << (5u * (r - 12u));
                    }
                }
            }
            for (uint32_t i = 0; i < GI2_SLOTS; i++) GI2_SLOT(i) = 0;
            uint32_t n = 0, prev = 0;
            bool bad = false;
            while (n < hlit + hdist) {
                s.norm();
                p = s.peek();
                uint32_t len, r;
                if (!gi2_decode<7>(c_cl, gi2_brev(p) >> 25, len, r) || r >= 19u) { bad = true; break; }
                const uint32_t sy = (uint32_t)((r < 12u ? srt0 >> (5u * r) : srt1 >> (5u * (r - 12u))) & 31ull);
                const uint32_t x = p >> len;         // the bits behind the code word
                uint32_t rep = 1, v = sy;
                if (sy == 16u) { if (!n) { bad = true; break; } v = prev; rep = 3u + (x & 3u); len += 2u; }
                else if (sy == 17u) { v = 0; rep = 3u + (x & 7u); len += 3u; }
                else if (sy == 18u) { v = 0; rep = 11u + (x & 127u); len += 7u; }
                else if (sy > 18u) { bad = true; break; }
                s.bo += len;
                if (n + rep > hlit + hdist) { bad = true; break; }
                if (v) for (uint32_t k = 0; k < rep; k++) {
                    const uint32_t y = n + k;
                    GI2_SLOT(y < hlit ? y : GI2_LIT + (y - hlit)) = (uint16_t)(v << 12);
                }
                n += rep; prev = v;
            }
            if (bad) { err = GI_E_CODES; break; }
            if (((uint32_t)GI2_SLOT(256) >> 12) == 0) { err = GI_E_CODES; break; }
            if (!build15(0, GI2_LIT, c_lit) || !build15(GI2_LIT, 30u, c_dist)) { err = GI_E_CODES; break; }
        }
        // ---- the symbols of this block ----
        for (;;) {
            s.norm();
            p = s.peek();
            uint32_t len, r;
            if (!gi2_decode<15>(c_lit, gi2_brev(p) >> 17, len, r) || r >= (fixed ? 288u : GI2_LIT)) { err = GI_E_SYMBOL; break; }
            const uint32_t sy = (uint32_t)GI2_SLOT(r) & 0x1ffu;
            if (sy < 256u) {
                s.bo += len;
                if (pos >= B.out_len) { err = GI_E_SIZE; break; }
#ifndef GI2_EXP_NOSTORE      // (measurement builds only: scripts/gpu_inflate_proto.hip -DGI2_EXP_NOSTORE -- what the literal stores cost)
                dst[pos] = (uint8_t)sy;
#endif
                pos++;
                run++;
                continue;
            }
            if (sy == 256u) { s.bo += len; break; }
            if (sy > 285u) { err = GI_E_SYMBOL; break; }
            const uint32_t k = sy - 257u;
            uint32_t mlen;
            if (k < 8u) mlen = 3u + k;
            else if (k == 28u) mlen = 258u;
            else { const uint32_t e = (k - 4u) >> 2; mlen = 3u + ((4u + (k & 3u)) << e) + ((p >> len) & ((1u << e) - 1u)); len += e; }
            s.bo += len;
            s.norm();
            p = s.peek();
            if (!gi2_decode<15>(c_dist, gi2_brev(p) >> 17, len, r) || r >= 30u) { err = GI_E_SYMBOL; break; }
            const uint32_t dc = fixed ? r : ((uint32_t)GI2_SLOT(GI2_LIT + r) & 0x1ffu);
            if (dc > 29u) { err = GI_E_SYMBOL; break; }
            uint32_t dist;
            if (dc < 4u) dist = dc + 1u;
            else { const uint32_t e = (dc >> 1) - 1u; dist = 1u + ((2u + (dc & 1u)) << e) + ((p >> len) & ((1u << e) - 1u)); len += e; }
            s.bo += len;
            if (dist > pos) { err = GI_E_DIST; break; }
            if (pos + mlen > B.out_len) { err = GI_E_SIZE; break; }
            if (ntok + 2u > tok_cap) { err = GI_E_SIZE; break; }      // (cannot happen for a stream that fits its output: one token per >= 3 bytes)
            if (run >= GI2_SKIP) { tk[ntok++] = (run << 9) | GI2_SKIP; run = 0; }
            tk[ntok++] = run | ((mlen - 3u) << 9) | ((dist - 1u) << 17);
            run = 0;
            pos += mlen;
        }
        // consumed more than the block holds?
        if (err == GI_OK && s.bits_used() > 8ull * B.in_len) err = GI_E_INPUT;
    }
    if (err == GI_OK && pos != B.out_len) err = GI_E_SIZE;
    n_tok[b] = ntok;
    status[b] = err;
    }
#undef GI2_SLOT
}

// ---------------------------------------------------------------------------------------------------------------------------------
// pass 2: one WAVEFRONT per block: the matches of the block's token list, 64 tokens at a time (see the header of this file).
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_inflate_resolve(const GiBlock *blocks, uint32_t n_blocks, uint8_t *out, const uint32_t *tokens,
                                                        const uint32_t *n_tok, const uint32_t *status, uint32_t b0) {
    __shared__ uint32_t s_dst[64], s_end[64], s_dist[64], s_pre[65];
    const uint32_t lane = threadIdx.x, b = blockIdx.x;
    if (b >= n_blocks || status[b] != GI_OK) return;           // (uniform over the wavefront)
    const GiBlock B = blocks[b];
    uint8_t *dst = out + B.out_off;
    const uint32_t *tk = tokens + gi2_tok_off(B.out_off, (unsigned long long)b0 + b);
    const uint32_t nt = n_tok[b];
    uint32_t base = 0;
    for (uint32_t t0 = 0; t0 < nt; t0 += 64u) {                 // (uniform)
        const uint32_t tok = t0 + lane < nt ? tk[t0 + lane] : GI2_SKIP;      // (behind the list: "skip 0 literals")
        const bool skip = (tok & 511u) == GI2_SKIP;
        const uint32_t lit = skip ? tok >> 9 : tok & 511u;
        const uint32_t mlen = skip ? 0u : ((tok >> 9) & 255u) + 3u, dist = skip ? 1u : (tok >> 17) + 1u;
        const uint32_t incl = kd_wave_scan_add(lit + mlen);
        const uint32_t d = base + (incl - (lit + mlen)) + lit;  // where this token's match starts
        const uint32_t src = d - dist, span = mlen < dist ? mlen : dist;     // its source: [src, src + span)  (dist <= d: pass 1 checked)
        GI_WAVE_SYNC();
        s_dst[lane] = d; s_end[lane] = d + mlen; s_dist[lane] = dist;
        GI_WAVE_SYNC();
        // the earlier tokens of the batch whose OUTPUT [s_dst, s_end) meets this lane's source: indices [ia, ib) -- both arrays ascend
        uint32_t ia = 0, ib = 0;
        {
            // (first index with a property that, once true, stays true: lo <= answer <= hi, 65 candidates -> 7 halvings; the probe
            // index is < 64 while lo < hi)
            uint32_t lo = 0, hi = 64;                            // first i with s_end[i] > src
#pragma unroll
            for (int it = 0; it < 7; it++) { const uint32_t m = (lo + hi) >> 1; if (lo < hi) { if (s_end[m] > src) hi = m; else lo = m + 1u; } }
            ia = lo;
            lo = 0; hi = 64;                                     // first i with s_dst[i] >= src + span
#pragma unroll
            for (int it = 0; it < 7; it++) { const uint32_t m = (lo + hi) >> 1; if (lo < hi) { if (s_dst[m] >= src + span) hi = m; else lo = m + 1u; } }
            ib = lo < lane ? lo : lane;
        }
        const unsigned long long dep = ib > ia ? (((ib < 64u ? 1ull << ib : 0ull) - 1ull) & ~((1ull << ia) - 1ull)) : 0ull;
        unsigned long long pending = kd_ballot(mlen != 0u);
        while (pending) {                                       // (uniform; the first pending match is always ready)
            const bool ready = ((pending >> lane) & 1ull) && (pending & dep) == 0ull;
            const unsigned long long rmask = kd_ballot(ready);
            const uint32_t inc = kd_wave_scan_add(ready ? mlen : 0u);
            GI_WAVE_SYNC();
            s_pre[lane + 1u] = inc;
            if (lane == 0) s_pre[0] = 0;
            GI_WAVE_SYNC();
            const uint32_t tot = kd_readlane(inc, 63);
            for (uint32_t k = lane; k < tot; k += 64u) {        // byte k of the ready matches' bytes, laid end to end
                uint32_t lo = 0, hi = 63;                       // first m with s_pre[m + 1] > k (m = 63 has it: k < tot): 64 candidates, 6 halvings
#pragma unroll
                for (int it = 0; it < 6; it++) { const uint32_t m = (lo + hi) >> 1; if (lo < hi) { if (s_pre[m + 1u] > k) hi = m; else lo = m + 1u; } }
                const uint32_t m = lo, o = k - s_pre[m], md = s_dist[m], mp = s_dst[m];
                // an overlapping match (distance < length) repeats its first `distance` bytes
                uint32_t j = o;
                if (o >= md) { j = o - (uint32_t)((float)o * (1.0f / (float)md)) * md; if (j >= md) j -= md; }
                dst[mp + o] = GI_LOAD_FAR(dst + (mp - md + j));
            }
            pending &= ~rmask;
            GI2_DRAIN();
        }
        base += kd_readlane(incl, 63);
    }
}
