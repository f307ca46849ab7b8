// kd_gpu_inflate.h -- raw DEFLATE (RFC 1951) of BGZF blocks ON THE GPU, one wavefront per block: the first stage of the device-side
// ingest (kd_ingest.h; SURVEY 8f rank 2: parse_bam's record iteration, kindel.py:131-153, without the host decoder).  Began as round 3's
// prototype (profiles/r03_gpu_inflate_prototype.json: 13 - 34 GB/s against the host decoder's 3.5 - 6 GB/s on all host threads); also
// built stand-alone by scripts/gpu_inflate_proto.hip (measurement) and tests/emu/gpu_inflate_emu.cpp (checked against zlib).
// The design is the simplest correct one: every lane of the wavefront decodes the SAME symbol stream (the Huffman walk is serial by
// nature) on the scalar unit, the 64 lanes share what is parallel (table construction, match copies, the output flush).  The host
// counterpart is kd_inflate.h.
//
//   * input: the lanes hold 256 bytes of the compressed stream in one register (lane l = dword l), the next 256 in another;
//     the bit buffer is refilled with v_readlane -- no memory latency on the symbol path;
//   * tables in LDS: 10-bit primary table for literals / lengths, 8-bit for distances (16-bit entries: symbol << 4 | code length); a
//     longer code (rare: they belong to rare symbols) falls back to the canonical bit-by-bit walk over count[] / sorted[];
//   * output: a 2 KiB ring in LDS (small on purpose: 6 KB of LDS per wavefront = 26 wavefronts per CU, and the kernel lives on
//     wavefronts in flight); every completed 256 bytes are stored to HBM by all lanes at once (one dword each); a match whose
//     source is still in the ring is copied LDS -> LDS by all lanes, an older one is read back from HBM with loads that bypass
//     the L1, behind a wait for all but the newest stores (its bytes left the ring many flushes ago);
//   * per block: a status word (0 = ok); nothing is ever written outside [out_off, out_off + out_len);
//   * bound (measured, profiles/r03_gpu_inflate_prototype.json): ~100 instructions per symbol, 70 of them on the scalar unit, issued
//     one per ~4 clocks and SIMD -- with 26 wavefronts per CU the scalar issue slots are ~75 % taken.  What moved it: the far-match
//     wait (a release fence per far match was 44 % of a quality-less file's time), no integer modulo per match, the small ring
//     (13 -> 26 wavefronts per CU): 34.1 -> 23.4 ms (Phred qualities), 12.9 -> 4.9 ms (none) for 448 MB.  What did not: a run loop for
//     literals with the next look-up in flight.  Beyond: several wavefronts per block from speculative bit positions.
#pragma once
#include <stdint.h>

#ifndef KD_EMU
#define GI_WAVE_SYNC()                                          \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#define GI_LOAD_FAR(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// Before a far match reads flushed output back: its newest source byte lies at least GI_NEAR - 257 bytes behind `pos` and `flushed` is
// at most 255 bytes behind `pos`, so at least (GI_NEAR - 513) / 256 flushes (one store instruction each) were issued AFTER the one
// that carried the byte; memory operations of a wavefront complete in the order issued, so "at most that many vector memory
// operations outstanding" means the byte has reached the L2 (which the sc1 loads read).  The count follows from the ring's
// geometry (GI_FAR_VMCNT below: 3 for the 2 KiB ring -- round 3 shipped vmcnt(8), derived for the 8 KiB ring it had first, with
// which the newest source byte's store could still be in flight: a timing-dependent stale read).  A release fence here waits for
// the NEWEST flush as well and writes the L2 back: 8 us per far match measured, 44 % of a quality-less file's inflate time.
#define GI_DRAIN_STORES() asm volatile(GI_FAR_WAIT ::: "memory")
__device__ __forceinline__ uint32_t gi_readlane(uint32_t v, uint32_t lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)lane));
}
#else
#define GI_WAVE_SYNC() KD_WAVE_SYNC()
#define GI_LOAD_FAR(p) (*(p))
#define GI_DRAIN_STORES()
static inline uint32_t gi_readlane(uint32_t v, uint32_t lane) { return kd_shfl(v, lane); }
#endif

// a value every lane holds alike, moved to a scalar register: the whole symbol walk (bit buffer, cursors, table entries) is
// wave-uniform, and on the scalar unit it costs one issue cycle per step instead of a vector instruction's latency
#ifndef KD_EMU
__device__ __forceinline__ uint32_t gi_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
#else
static inline uint32_t gi_uni(uint32_t v) { return v; }
#endif

#ifndef GI_RING
#define GI_RING 2048u      // (bytes, a power of two >= 1024; measured 16 / 8 / 4 / 2 KiB: 34.2 / 29.8 / 26.5 / 23.4 ms -- wavefronts per CU decide)
#endif
#define GI_RING_MASK (GI_RING - 1u)
#define GI_NEAR (GI_RING - 512u)      // a match at most this far back is copied inside the ring
// vector memory operations that may still be outstanding when a far match reads its source back (GI_DRAIN_STORES)
#define GI_FAR_VMCNT ((GI_NEAR - 513u) / 256u)
#if GI_RING < 1024u
#error "GI_RING: a power of two >= 1024 (a far match's source must have left the ring at least one flush ago)"
#elif (GI_RING - 512u - 513u) / 256u >= 8u
#define GI_FAR_WAIT "s_waitcnt vmcnt(8)"
#elif (GI_RING - 512u - 513u) / 256u >= 3u
#define GI_FAR_WAIT "s_waitcnt vmcnt(3)"
#elif (GI_RING - 512u - 513u) / 256u >= 1u
#define GI_FAR_WAIT "s_waitcnt vmcnt(1)"
#else
#define GI_FAR_WAIT "s_waitcnt vmcnt(0)"
#endif
static_assert((GI_RING & (GI_RING - 1u)) == 0u && GI_NEAR + 258u <= GI_RING, "k_gpu_inflate: a near match (source and the bytes it writes) must fit the ring");
#define GI_LIT_BITS 10
#define GI_DIST_BITS 8
#define GI_CL_BITS 7

enum { GI_OK = 0, GI_E_BTYPE = 1, GI_E_STORED = 2, GI_E_CODES = 3, GI_E_SYMBOL = 4, GI_E_DIST = 5, GI_E_SIZE = 6, GI_E_INPUT = 7 };

struct __attribute__((packed, aligned(1))) GiU32 { uint32_t v; };

struct GiBlock { unsigned long long in_off, out_off; uint32_t in_len, out_len; };

// one Huffman code: primary table + what the canonical walk needs
struct GiCode { uint16_t *table; uint16_t *cnt; uint16_t *sorted; uint32_t bits; };

// Serial part of a table's construction (lane 0; `flag` = 0 if the lengths do not form a code zlib would accept):
// counts per length, the symbols sorted by (length, value).
__device__ __forceinline__ void gi_build_serial(const uint8_t *lens, uint32_t n, const GiCode &c, bool cl_code, uint32_t *flag) {
    uint16_t offs[16];
    for (int l = 0; l < 16; l++) c.cnt[l] = 0;
    for (uint32_t s = 0; s < n; s++) c.cnt[lens[s]]++;
    c.cnt[0] = 0;
    int left = 1, mx = 0;
    bool ok = true;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - (int)c.cnt[l]; if (left < 0) ok = false; if (c.cnt[l]) mx = l; }
    if (left > 0 && mx != 0 && (cl_code || mx != 1)) ok = false;        // zlib: incomplete set
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + c.cnt[l]);
    for (uint32_t s = 0; s < n; s++) if (lens[s]) c.sorted[offs[lens[s]]++] = (uint16_t)s;
    *flag = ok ? 1u : 0u;
}

// the entry of primary index i: symbol << 4 | code length, or 0 = no code of <= bits length starts with these bits
__device__ __forceinline__ uint32_t gi_entry(const GiCode &c, uint32_t i) {
    uint32_t code = 0, first = 0, index = 0;
    for (uint32_t l = 1; l <= c.bits; l++) {
        code |= (i >> (l - 1)) & 1u;
        const uint32_t cn = c.cnt[l];
        if (code < first + cn) return ((uint32_t)c.sorted[index + (code - first)] << 4) | l;
        index += cn; first += cn; first <<= 1; code <<= 1;
    }
    return 0;
}

struct GiStream {
    const uint8_t *in;      // the block's compressed bytes
    uint32_t n_dw;          // dwords that may be loaded (the rest reads as 0)
    uint32_t idx;           // next dword to enter the bit buffer
    uint32_t cur, nxt;      // lane l: dword (window base + l) / (window base + 64 + l)
    unsigned long long bb;
    uint32_t bc;
};

__device__ __forceinline__ uint32_t gi_load_dw(const GiStream &s, uint32_t d) {
    return d < s.n_dw ? reinterpret_cast<const GiU32 *>(s.in + 4ull * d)->v : 0u;
}
// (one dword is enough: no consumer takes more than 32 bits between two refills -- a literal / length code and its extra bits 20, a
// distance code and its extra bits 28, LEN + NLEN of a stored block 32 -- and a buffer of >= 0 bits + 32 holds them)
__device__ __forceinline__ void gi_refill(GiStream &s, uint32_t lane) {
    if (s.bc <= 32) {
        const uint32_t w = gi_readlane(s.cur, s.idx & 63u);
        s.bb |= (unsigned long long)w << s.bc;
        s.bc += 32;
        s.idx++;
        if ((s.idx & 63u) == 0) { s.cur = s.nxt; s.nxt = gi_load_dw(s, s.idx + 64u + lane); }
    }
}
__device__ __forceinline__ uint32_t gi_take(GiStream &s, uint32_t n) {
    const uint32_t v = (uint32_t)(s.bb & ((1ull << n) - 1ull));
    s.bb >>= n; s.bc -= n;
    return v;
}
// one symbol of code c (the bit buffer holds > 32 bits); 0xffffffff = no such code word
__device__ __forceinline__ uint32_t gi_symbol(GiStream &s, const GiCode &c) {
    const uint32_t e = gi_uni(c.table[(uint32_t)s.bb & ((1u << c.bits) - 1u)]);
    if (e & 15u) { s.bb >>= (e & 15u); s.bc -= (e & 15u); return e >> 4; }
    uint32_t code = 0, first = 0, index = 0;      // a code longer than the primary index: the canonical walk, bit by bit
#pragma unroll 1
    for (uint32_t l = 1; l <= 15; l++) {
        code |= (uint32_t)(s.bb & 1ull);
        s.bb >>= 1; s.bc -= 1;
        const uint32_t cn = gi_uni(c.cnt[l]);
        if (code < first + cn) return gi_uni(c.sorted[index + (code - first)]);
        index += cn; first += cn; first <<= 1; code <<= 1;
    }
    return 0xffffffffu;
}

// One wavefront inflates one block.  comp: the file (or any buffer the blocks' in_off index), readable 8 bytes past every
// block's input; out: the inflated bytes of all blocks; status[b] = GI_OK or what went wrong.
__global__ void __launch_bounds__(64) k_gpu_inflate(const uint8_t *comp, const GiBlock *blocks, uint32_t n_blocks, uint8_t *out,
                                                    uint32_t *status) {
    __shared__ __attribute__((aligned(16))) uint8_t ring[GI_RING];
    __shared__ uint16_t t_lit[1u << GI_LIT_BITS], t_dist[1u << GI_DIST_BITS], t_cl[1u << GI_CL_BITS];
    __shared__ uint16_t cnt_lit[16], cnt_dist[16], cnt_cl[16], srt_lit[288], srt_dist[32], srt_cl[19];
    __shared__ uint8_t lens[288 + 32 + 4];
    __shared__ uint32_t s_flag;
    const uint32_t lane = threadIdx.x, b = blockIdx.x;
    if (b >= n_blocks) return;
    const GiBlock B = blocks[b];
    uint8_t *dst = out + B.out_off;
    GiStream s;
    s.in = comp + B.in_off;
    s.n_dw = (B.in_len + 3u) / 4u + 1u;
    s.idx = 0; s.bb = 0; s.bc = 0;
    s.cur = gi_load_dw(s, lane); s.nxt = gi_load_dw(s, 64u + lane);
    const GiCode c_lit{t_lit, cnt_lit, srt_lit, GI_LIT_BITS}, c_dist{t_dist, cnt_dist, srt_dist, GI_DIST_BITS}, c_cl{t_cl, cnt_cl, srt_cl, GI_CL_BITS};
    uint32_t pos = 0, flushed = 0, err = GI_OK;
    uint32_t in_rem = B.in_len;   // bytes of the block's input from s.in on (a stored block restarts the window behind its bytes)
    bool dirty = false;      // flushed bytes whose stores may still be on their way
#ifdef GI_CLOCKS             // profiling build (scripts/gpu_inflate_proto.py --clocks): where the wavefront's clocks go
    long long gc_mark = clock64(), gc_build = 0, gc_sym = 0, gc_near = 0, gc_far = 0, gc_flush = 0, gc_other = 0;
    unsigned long long gn_lit = 0, gn_near = 0, gn_far = 0, gn_flush = 0, gn_slow = 0;
#define GI_MARK(acc) { const long long n_ = clock64(); acc += n_ - gc_mark; gc_mark = n_; }
#define GI_COUNT(x) x++
#else
#define GI_MARK(acc)
#define GI_COUNT(x)
#endif
    // all completed 256-byte pieces of the ring -> HBM, one dword per lane and piece
    auto flush = [&]() {
        GI_MARK(gc_sym) GI_COUNT(gn_flush);
        GI_WAVE_SYNC();
        while (pos - flushed >= 256u) {
            const uint32_t at = flushed + 4u * lane;
            reinterpret_cast<GiU32 *>(dst + at)->v = *reinterpret_cast<const uint32_t *>(ring + (at & GI_RING_MASK));
            flushed += 256u;
        }
        dirty = true;
        GI_WAVE_SYNC();
        GI_MARK(gc_flush)
    };
    auto build = [&](const uint8_t *ln, uint32_t n, const GiCode &c, bool cl_code) -> bool {
        GI_WAVE_SYNC();
        if (lane == 0) gi_build_serial(ln, n, c, cl_code, &s_flag);
        GI_WAVE_SYNC();
        for (uint32_t i = lane; i < (1u << c.bits); i += 64u) c.table[i] = (uint16_t)gi_entry(c, i);
        GI_WAVE_SYNC();
        return gi_uni(s_flag) != 0;
    };
    for (bool last = false; !last && err == GI_OK;) {
        gi_refill(s, lane);
        last = gi_take(s, 1) != 0;
        const uint32_t type = gi_take(s, 2);
        if (type == 0) {
            // stored: skip to the byte boundary, LEN / NLEN, then LEN bytes straight from the input
            gi_take(s, s.bc & 7u);
            gi_refill(s, lane);
            const uint32_t len = gi_take(s, 16), nlen = gi_take(s, 16);
            if ((len ^ nlen) != 0xffffu) { err = GI_E_STORED; break; }
            if (pos + len > B.out_len) { err = GI_E_SIZE; break; }
            // the bit buffer holds whole bytes now: the next unread input byte is
            const unsigned long long at = 4ull * s.idx - s.bc / 8u;
            if (at + len > in_rem) { err = GI_E_INPUT; break; }    // (against what is LEFT of the block: a second stored block must not read on)
            for (uint32_t done = 0; done < len;) {      // through the ring like everything else, 256 bytes at a time
                const uint32_t step = len - done < 256u ? len - done : 256u;
                GI_WAVE_SYNC();
                for (uint32_t i = lane; i < step; i += 64u) ring[(pos + i) & GI_RING_MASK] = s.in[at + done + i];
                GI_WAVE_SYNC();
                pos += step; done += step;
                if (pos - flushed >= 256u) flush();
            }
            // restart the input window behind the stored bytes
            const unsigned long long next = at + len;
            in_rem -= (uint32_t)next;
            s.in += next; s.n_dw = (in_rem + 3u) / 4u + 1u;
            s.idx = 0; s.bb = 0; s.bc = 0;
            s.cur = gi_load_dw(s, lane); s.nxt = gi_load_dw(s, 64u + lane);
            continue;
        }
        if (type == 3) { err = GI_E_BTYPE; break; }
        uint32_t hlit = 288, hdist = 32;
        if (type == 1) {
            GI_WAVE_SYNC();
            for (uint32_t i = lane; i < 288u + 32u; i += 64u) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
        } else {
            gi_refill(s, lane);
            hlit = gi_take(s, 5) + 257u; hdist = gi_take(s, 5) + 1u;
            const uint32_t hclen = gi_take(s, 4) + 4u;
            if (hlit > 286u || hdist > 30u) { err = GI_E_CODES; break; }
            GI_WAVE_SYNC();
            if (lane < 19) lens[lane] = 0;
            GI_WAVE_SYNC();
            for (uint32_t i = 0; i < hclen; i++) {
                gi_refill(s, lane);
                const uint32_t v = gi_take(s, 3);
                const uint8_t ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                if (lane == 0) lens[ord[i]] = (uint8_t)v;
            }
            if (!build(lens, 19, c_cl, true)) { err = GI_E_CODES; break; }
            // the hlit + hdist code lengths (kept in registers-free form: every lane walks the same stream, lane 0 stores)
            uint32_t n = 0, prev = 0;
            bool bad = false;
            GI_WAVE_SYNC();
            while (n < hlit + hdist) {
                gi_refill(s, lane);
                const uint32_t sy = gi_symbol(s, c_cl);
                if (sy > 18u) { bad = true; break; }
                uint32_t rep = 1, v = sy;
                if (sy == 16) { if (!n) { bad = true; break; } v = prev; rep = 3 + gi_take(s, 2); }
                else if (sy == 17) { v = 0; rep = 3 + gi_take(s, 3); }
                else if (sy == 18) { v = 0; rep = 11 + gi_take(s, 7); }
                if (n + rep > hlit + hdist) { bad = true; break; }
                if (lane == 0) for (uint32_t k = 0; k < rep; k++) lens[n + k] = (uint8_t)v;
                n += rep; prev = v;
            }
            if (bad) { err = GI_E_CODES; break; }
            GI_WAVE_SYNC();
            if (gi_uni(lens[256]) == 0) { err = GI_E_CODES; break; }
        }
        // literal / length code over lens[0, hlit), distance code over lens[hlit, hlit + hdist)
        if (!build(lens, hlit, c_lit, false) || !build(lens + hlit, hdist, c_dist, false)) { err = GI_E_CODES; break; }
        GI_MARK(gc_build)
        for (;;) {
            gi_refill(s, lane);
            const uint32_t sy = gi_symbol(s, c_lit);
            if (sy < 256u) {
                if (pos >= B.out_len) { err = GI_E_SIZE; break; }
                if (lane == 0) ring[pos & GI_RING_MASK] = (uint8_t)sy;
                pos++;
                GI_COUNT(gn_lit);
            } else if (sy == 256u) {
                break;
            } else {
                if (sy > 285u) { err = GI_E_SYMBOL; break; }
                const uint32_t k = sy - 257u;
                uint32_t len;
                if (k < 8u) len = 3u + k;
                else if (k == 28u) len = 258u;
                else { const uint32_t e = (k - 4u) >> 2; len = 3u + ((4u + (k & 3u)) << e) + gi_take(s, e); }
                gi_refill(s, lane);
                const uint32_t dc = gi_symbol(s, c_dist);
                if (dc > 29u) { err = GI_E_SYMBOL; break; }
                uint32_t dist;
                if (dc < 4u) dist = dc + 1u;
                else { const uint32_t e = (dc >> 1) - 1u; dist = 1u + ((2u + (dc & 1u)) << e) + gi_take(s, e); }
                if (dist > pos) { err = GI_E_DIST; break; }
                if (pos + len > B.out_len) { err = GI_E_SIZE; break; }
                GI_MARK(gc_sym)
                GI_WAVE_SYNC();
                if (dist <= GI_NEAR) {
                    // inside the ring.  (The branch is wave-uniform; `i % dist` for every match was ~35 vector instructions each.)
                    if (dist >= len) {
                        for (uint32_t i = lane; i < len; i += 64u) ring[(pos + i) & GI_RING_MASK] = ring[(pos - dist + i) & GI_RING_MASK];
                    } else {
                        // an overlapping match repeats its first `dist` bytes: i mod dist through a float reciprocal (i < 322, dist < 258:
                        // the product is off by far less than 1 / dist, so the floor is the quotient or one less)
                        const float rd = 1.0f / (float)dist;
                        for (uint32_t i = lane; i < len; i += 64u) {
                            uint32_t j = i - (uint32_t)((float)i * rd) * dist;
                            if (j >= dist) j -= dist;
                            ring[(pos + i) & GI_RING_MASK] = ring[(pos - dist + j) & GI_RING_MASK];
                        }
                    }
                } else {
                    if (dirty) { GI_DRAIN_STORES(); dirty = false; }
                    for (uint32_t i = lane; i < len; i += 64u) ring[(pos + i) & GI_RING_MASK] = GI_LOAD_FAR(dst + (pos - dist + i));
                }
                GI_WAVE_SYNC();
                pos += len;
#ifdef GI_CLOCKS
                if (dist <= GI_NEAR) { GI_MARK(gc_near) gn_near++; } else { GI_MARK(gc_far) gn_far++; }
#endif
            }
            if (pos - flushed >= 256u) flush();
        }
        // consumed more than the block holds?
        if (err == GI_OK && 32ull * s.idx - s.bc > 8ull * in_rem) err = GI_E_INPUT;
    }
    if (err == GI_OK && pos != B.out_len) err = GI_E_SIZE;
    // the tail of the ring, byte by byte
    GI_WAVE_SYNC();
    if (err == GI_OK) for (uint32_t i = flushed + lane; i < pos; i += 64u) dst[i] = ring[i & GI_RING_MASK];
    if (lane == 0) status[b] = err;
#ifdef GI_CLOCKS
    GI_MARK(gc_sym)
    if (lane == 0) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(status + ((n_blocks + 1u) & ~1u));     // 16 words behind the status array
        const unsigned long long v[11] = {(unsigned long long)gc_build, (unsigned long long)gc_sym, (unsigned long long)gc_near, (unsigned long long)gc_far,
                                          (unsigned long long)gc_flush, (unsigned long long)gc_other, gn_lit, gn_near, gn_far, gn_flush, gn_slow};
        for (int k = 0; k < 11; k++) atomicAdd(&dbg[k], v[k]);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// k_bgzf_crc: the CRC-32 of every inflated block against the block's trailer, as htslib's bgzf_read (and the host reader,
// kd_crc32.h) check it.  A wavefront takes blocks in turn; inside a block lane l owns dword column l of the rows of 256 bytes, so
// every load is one coalesced 256-byte row, and CRC's linearity puts the 64 columns together:
//     state after A || B  =  Z_|B|(state after A)  xor  raw(B)         Z_k = "k zero bytes" = multiplication by x^(8k) mod P
// per row  acc = Z_256(acc) xor raw(own dword)  (Z_256 and raw() by byte-indexed tables in LDS, built once per wavefront), at the end
// every lane shifts its column to the block's end (x^(8k) by square-and-multiply over x^(2^j), zlib's multmodp / x2nmodp), the
// lanes are xor-ed together, and the initial 0xffffffff travels |block| bytes.  ~10 us per 64 KiB block and wavefront.
// A pure-Python model of exactly this decomposition is pinned against zlib.crc32 in tests/test_gpu_inflate_proto.py.
// ---------------------------------------------------------------------------------------------------------------------
#define GI_CRC_POLY 0xEDB88320u
// a(x) * b(x) mod P, reflected bit order (bit 31 = x^0); a != 0 (zlib: multmodp)
__device__ __forceinline__ uint32_t gi_multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (int it = 0; it < 32; it++) {
        if (a & m) { p ^= b; if ((a & (m - 1u)) == 0) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ GI_CRC_POLY : b >> 1;
    }
    return p;
}
// x^(8k) mod P from the table x2n[j] = x^(2^j)
__device__ __forceinline__ uint32_t gi_xpow8(const uint32_t *x2n, uint32_t k) {
    uint32_t p = 1u << 31;
    uint32_t n = 8u * k;
    for (uint32_t j = 0; n; n >>= 1, j++) if (n & 1u) p = gi_multmodp(x2n[j], p);
    return p;
}
__global__ void __launch_bounds__(64) k_bgzf_crc(const uint8_t *comp, const GiBlock *blocks, uint32_t n_blocks, const uint8_t *out,
                                                  uint32_t *n_bad) {
    __shared__ uint32_t t_byte[256], t_z256[4][256], x2n[24];
    const uint32_t lane = threadIdx.x;
    if (lane == 0) {
        x2n[0] = 1u << 30;                              // x^1
        for (int j = 1; j < 24; j++) x2n[j] = gi_multmodp(x2n[j - 1], x2n[j - 1]);
    }
    for (uint32_t i = lane; i < 256u; i += 64u) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ GI_CRC_POLY : c >> 1;
        t_byte[i] = c;
    }
    GI_WAVE_SYNC();
    const uint32_t c256 = gi_xpow8(x2n, 256u);
    for (uint32_t i = lane; i < 1024u; i += 64u) t_z256[i >> 8][i & 255u] = gi_multmodp(c256, (i & 255u) << (8u * (i >> 8)));
    GI_WAVE_SYNC();
    auto raw = [&](uint32_t s, uint32_t byte) -> uint32_t { return t_byte[(s ^ byte) & 0xffu] ^ (s >> 8); };
    for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const GiBlock B = blocks[b];
        const uint8_t *d = out + B.out_off;
        const uint32_t n = B.out_len, rows = n >> 8;
        uint32_t acc = 0, end = 0;                      // end: the byte position acc has been shifted to (0 = no byte yet)
        for (uint32_t j = 0; j < rows; j++) {
            const uint32_t w = reinterpret_cast<const GiU32 *>(d + 256u * j + 4u * lane)->v;
            uint32_t r = raw(0u, w & 0xffu);
            r = raw(r, (w >> 8) & 0xffu); r = raw(r, (w >> 16) & 0xffu); r = raw(r, w >> 24);
            acc = t_z256[0][acc & 0xffu] ^ t_z256[1][(acc >> 8) & 0xffu] ^ t_z256[2][(acc >> 16) & 0xffu] ^ t_z256[3][acc >> 24] ^ r;
        }
        if (rows) end = 256u * (rows - 1u) + 4u * lane + 4u;
        const uint32_t s0 = 256u * rows + 4u * lane;    // the partial last row
        if (s0 < n) {
            const uint32_t k = n - s0 < 4u ? n - s0 : 4u;
            uint32_t r = 0;
            for (uint32_t t = 0; t < k; t++) r = raw(r, d[s0 + t]);
            acc = (end ? gi_multmodp(gi_xpow8(x2n, s0 + k - end), acc) : 0u) ^ r;
            end = s0 + k;
        }
        uint32_t v = (end && end < n) ? gi_multmodp(gi_xpow8(x2n, n - end), acc) : (end ? acc : 0u);
        if (lane == 0) v ^= gi_multmodp(gi_xpow8(x2n, n), 0xffffffffu);          // the initial state, n bytes on
#ifndef KD_EMU
        for (uint32_t m = 1; m < 64u; m <<= 1) v ^= (uint32_t)__shfl_xor((int)v, (int)m, 64);
#else
        for (uint32_t m = 1; m < 64u; m <<= 1) v ^= kd_shfl_xor(v, m);
#endif
        const uint32_t crc = v ^ 0xffffffffu;
        const uint32_t want = reinterpret_cast<const GiU32 *>(comp + B.in_off + B.in_len)->v;
        if (lane == 0 && crc != want) atomicAdd(n_bad, 1u);
    }
}
