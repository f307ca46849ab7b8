// kd_window.h -- k_window: the LDS-histogram pileup (dominant kernel), its walkers.
// Part of the device code of kd_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
#include "kd_common.h"

// k_window: persistent workgroups pull (window, slice) work items (k_plan_*).
//
// LDS (dynamic): u16 counters, site-pair major: hist[(W + 2 * KD_HALO) / 2 pairs][19 channels] dwords, two sites per dword:
// three groups of {A,T,G,C,N, bad} -- weights (0-5), clip_start_weights (7-12), clip_end_weights (13-18) -- plus deletions (6).
// "bad" collects bases outside A,C,G,T,N (KeyError in the reference; checked at flush).  38 B per site: W = 640 is
// 25 KB + 4 KB of read lists, five workgroups (20 wavefronts) per CU.  Soft clips are tallied here too because 4 x 10^7
// scattered device-scope atomics cost as much as the whole LDS pass (measured: 1.4 ms vs 1.5 ms).
//
// One LANE per read.  Per 1024-read tile the threads sort the window's reads into a PLAIN list (one M/=/X run spanning
// the read: nothing to decode) and a COMPLEX list; wavefronts then take rows of 64 list entries, lane l the entry
// l * rows + r, so that the lanes of a wavefront sit `rows` reads apart in the coordinate-sorted batch and rarely hit the
// same counter in the same instruction.
//   kd_walk_plain   16-byte chunks of packed bases at any byte address, three chunks of prefetch; a dword = 8 bases: their
//                   channel offsets by three v_perm_b32 look-ups, then per base ONE byte add (pointer + offset) and one
//                   ds_add_u32 with the site pair in the immediate offset (masked variant at the ends of the run / window)
//   kd_walk_short   reads with clips / indels and <= 16 ops: CIGAR -> at most three segments cut to the window, then one
//                   software-pipelined loop over (segment, chunk) steps of branch-free masked adds
//   kd_walk_ops     the general op-by-op walk: short reads with more than three segments, and -- in the second launch --
//                   the SEGMENTS of long reads (k_prep_long), one lane per segment
// Only regular reads are handled here; their clip_starts / clip_ends counters and insertion events are done by
// k_cold_lane / k_long_expand, irregular reads by k_pileup_wave.
#define KD_HCH 19
#ifndef KD_EXP_DELETE
#define KD_EXP_DELETE 0    // TIMING EXPERIMENTS only (wrong tables; scripts/exp/kwindow_deletions.sh): 1 no adds of the plain reads inside the window, 2 no
#endif                     // base loads there, 4 no complex walk, 8 no flush, 16 no zeroing, 32 no classification, 64 no fetch of the tiles' keys,
                           // 128 the plain reads' chunk loads wave-uniform (one line per load instruction instead of 64)
#define KD_HCH_DEL 6u
#define KD_HCH_CSW 7u
#define KD_HCH_CEW 13u

// The LDS histogram is SITE-PAIR major: hist[pair P][channel] with KD_HPITCH = 19 dwords per pair of sites, the dword of
// (P, ch) holding the u16 counters of sites 2P (low half) and 2P + 1 (high half): a work item tallies at most `slice`
// <= 32768 reads and a read adds at most 1 to a counter, so a half cannot overflow into its neighbour.  The 19 channels
// of a pair: weights A,T,G,C,N,bad (0-5), deletions (6), clip_start_weights (7-12), clip_end_weights (13-18).
// With the channel as the FAST index the per-base address is  pointer + 4 * channel  -- a byte add of the looked-up code,
// no multiply -- and the 8 bases of a packed dword get their codes from three v_perm_b32 table look-ups (kd_codes8)
// instead of one 64-bit LUT shift per base.  19 is odd: both the walk (lanes on different pairs) and the flush (lane =
// pair, fixed channel, stride 19 dwords) spread over the 32 banks.
// KD_HALO extra sites on both sides of the window: window clipping is done at DWORD granularity of the packed bases (a
// dword that straddles the window edge is added whole, its outside bases land in the halo and are never flushed), so the
// masked path below is only needed at the ends of a run.
// Counter of (channel ch, window-relative site s in [-KD_HALO, W + KD_HALO)) = half (s & 1) of hist0[(s >> 1) * 19 + ch],
// hist0 = hist + (KD_HALO / 2) * 19; Wh = (W + 2 * KD_HALO) / 2 pairs.
#define KD_SEQ_AT(rd, i) ((rd).seq4 + KD_SOFF(rd, i))
#define KD_HALO 8
#define KD_HPITCH 19
#define KD_HPITCHB (4 * KD_HPITCH)

// 8 packed bases -> byte offsets of their channels inside a pair (4 * {A,T,G,C,N = 0..4; anything else = 5, the group's
// bad slot}): rh = bases 0,2,4,6, rl = bases 1,3,5,7 (byte k = base 2k / 2k + 1).  Two 8-entry tables (bit 3 of the nibble
// clear / set) looked up with v_perm_b32, a third v_perm selects per byte.
// ROWS: v holds eight ROW SYMBOLS of an expanded long read (kd_long.h) -- the symbol IS the channel: two instructions.
template <bool ROWS = false>
__device__ __forceinline__ void kd_codes8(uint32_t v, uint32_t &rh, uint32_t &rl) {
    if (ROWS) { rl = (v & 0x0f0f0f0fu) << 2; rh = (v >> 2) & 0x3c3c3c3cu; return; }
    const uint32_t TL_LO = 0x140c0014u, TL_HI = 0x14141408u;   // nibbles 0-7:  '=',A,C,M,G,R,S,V
    const uint32_t TH_LO = 0x14141404u, TH_HI = 0x10141414u;   // nibbles 8-15: T,W,Y,H,K,D,B,N
    const uint32_t tl = v & 0x07070707u, th = (v >> 4) & 0x07070707u;
    const uint32_t sl = ((v >> 1) & 0x04040404u) | 0x03020100u, sh = ((v >> 5) & 0x04040404u) | 0x03020100u;
    rl = kd_perm(kd_perm(TH_HI, TH_LO, tl), kd_perm(TL_HI, TL_LO, tl), sl);
    rh = kd_perm(kd_perm(TH_HI, TH_LO, th), kd_perm(TL_HI, TL_LO, th), sh);
}

__device__ __forceinline__ void kd_hadd(uint32_t *hist0, uint32_t ch, int32_t s) {
    atomicAdd(&hist0[KD_MUL24S(s >> 1, KD_HPITCH) + (int32_t)ch], 1u << (16 * (s & 1)));
}
// all 8 bases of dword v are added; s0 = window-relative site of its first base, gb = byte offset of the channel group
// (0 weights, 28 clip_start_weights, 52 clip_end_weights).  Even bases go through pointer h with add value vp, odd bases
// through hq = h + (s0 & 1) pairs with vq: no per-base parity arithmetic; the pair of base b is an immediate offset.
template <bool ROWS = false>
__device__ __forceinline__ void kd_add8_full(uint32_t *hist0, uint32_t v, int32_t s0, uint32_t gb) {
    const int32_t p = s0 & 1;
    unsigned char *h = reinterpret_cast<unsigned char *>(hist0) + KD_MUL24S(s0 >> 1, KD_HPITCHB) + gb;
    unsigned char *hq = h + KD_HPITCHB * p;
    const uint32_t vp = 1u << (16 * p), vq = 0x10000u >> (16 * p);
    uint32_t rh, rl;
    kd_codes8<ROWS>(v, rh, rl);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const uint32_t code = (((b & 1) ? rl : rh) >> (8 * (b >> 1))) & 0xffu;
        unsigned char *a = ((b & 1) ? hq : h) + code + KD_HPITCHB * (b >> 1);
        atomicAdd(reinterpret_cast<uint32_t *>(a), (b & 1) ? vq : vp);
    }
}
// only bases [blo, bhi) belong to the run
template <bool ROWS = false>
__device__ __forceinline__ void kd_add8_part(uint32_t *hist0, uint32_t v, int32_t s0, int32_t blo, int32_t bhi, uint32_t gb) {
    uint32_t rh, rl;
    kd_codes8<ROWS>(v, rh, rl);
#pragma unroll
    for (int b = 0; b < 8; b++)
        if (b >= blo && b < bhi) kd_hadd(hist0, (gb >> 2) + (((((b & 1) ? rl : rh) >> (8 * (b >> 1))) & 0xffu) >> 2), s0 + b);
}
// One memory dword of a run.  xs = query index of the dword's first base; [xa, xb) = the run's query bases that
// fall inside the window (decides whether the dword is touched at all); [ra, rb) = the run's own query bases
// (decides which of its 8 bases exist); site of base x is sx + x.
template <bool ROWS = false>
__device__ __forceinline__ void kd_add_dword(uint32_t *hist0, uint32_t v, int32_t xs, int32_t xa, int32_t xb,
                                             int32_t ra, int32_t rb, int32_t sx, uint32_t gb) {
    if (xs + 8 <= xa || xs >= xb) return;
    if (xs >= ra && xs + 8 <= rb) kd_add8_full<ROWS>(hist0, v, sx + xs, gb);
    else kd_add8_part<ROWS>(hist0, v, sx + xs, ra - xs, rb - xs, gb);
}

// Bases of dword v whose bit is set in m (bit b = base b) are added; the others add 0 to a counter at most
// 7 sites away from a live one, i.e. inside the halo.  No branches: lanes whose dword is cut by a run end, a clip end or
// the window edge stay in step with lanes whose dword is whole.
__device__ __forceinline__ void kd_add8_masked(uint32_t *hist0, uint32_t v, int32_t s0, uint32_t m, uint32_t gb) {
    const int32_t p = s0 & 1;
    unsigned char *h = reinterpret_cast<unsigned char *>(hist0) + KD_MUL24S(s0 >> 1, KD_HPITCHB) + gb;
    unsigned char *hq = h + KD_HPITCHB * p;
    const uint32_t vp = 1u << (16 * p), vq = 0x10000u >> (16 * p);
    const uint32_t me = m << (16 * p), mo = m << (16 - 16 * p);   // bit b of m moved onto the add value's bit
    uint32_t rh, rl;
    kd_codes8(v, rh, rl);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const uint32_t code = (((b & 1) ? rl : rh) >> (8 * (b >> 1))) & 0xffu;
        unsigned char *a = ((b & 1) ? hq : h) + code + KD_HPITCHB * (b >> 1);
        atomicAdd(reinterpret_cast<uint32_t *>(a), (b & 1) ? ((mo >> b) & vq) : ((me >> b) & vp));
    }
}
// one 16-byte chunk (query bases xs .. xs+31) against the live query range [lo, hi) of a segment: the live bases as ONE
// 32-bit mask, one pointer pair for the chunk (its dwords' counters sit at compile-time offsets: 4 site pairs per dword,
// and the parity of the first site is the same for all four), dwords without a live base skipped
__device__ __forceinline__ void kd_add_chunk_masked(uint32_t *hist0, const KdChunk &cur, int32_t xs, int32_t lo,
                                                    int32_t hi, int32_t sx, uint32_t gb) {
    int32_t l = lo - xs, h = hi - xs;
    l = l < 0 ? 0 : l;
    h = h > 32 ? 32 : h;
    if (h <= l) return;
    const uint32_t live = (0xffffffffu >> (32 - h)) & (0xffffffffu << l);   // 0 <= l < h <= 32
    const int32_t s0 = sx + xs;
    const int32_t p = s0 & 1;
    unsigned char *hb = reinterpret_cast<unsigned char *>(hist0) + KD_MUL24S(s0 >> 1, KD_HPITCHB) + gb;
    unsigned char *hq = hb + KD_HPITCHB * p;
    const uint32_t vp = 1u << (16 * p), vq = 0x10000u >> (16 * p);
    const uint32_t dw[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t m = (live >> (8 * d)) & 0xffu;
        if (!m) continue;
        const uint32_t me = m << (16 * p), mo = m << (16 - 16 * p);   // bit b of m moved onto the add value's bit
        uint32_t rh, rl;
        kd_codes8(dw[d], rh, rl);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint32_t code = (((b & 1) ? rl : rh) >> (8 * (b >> 1))) & 0xffu;
            unsigned char *a = ((b & 1) ? hq : hb) + code + KD_HPITCHB * (4 * d + (b >> 1));
            atomicAdd(reinterpret_cast<uint32_t *>(a), (b & 1) ? ((mo >> b) & vq) : ((me >> b) & vp));
        }
    }
}

// General per-lane walk of one regular read against the window (reads with clips, indels, long CIGARs).
// A state machine over WORK UNITS (one 16-byte chunk = up to 32 live bases of the current run), not over
// CIGAR ops, so that lanes keep adding bases together whatever their op structure.  A soft clip is a
// run of its own on the clip_start / clip_end channel group.
// Ops [k, k_end) of read i, entered with the reference cursor at window-relative site `grel` and the query cursor
// at q: the whole CIGAR of a short read with many segments (k = 0, k_end = n_cig), or ONE SEGMENT of a long read
// (k_prep_long's checkpoint).  `lead` / `foot_end`: reach of the leading clip / window-relative end of the footprint
// (used by the clip ops, which sit in the first / last segment).
__device__ __forceinline__ void kd_walk_ops(const KdReads &rd, kd_u64 i, uint32_t nc, uint32_t k, uint32_t k_end, int32_t grel, int32_t q,
                                            int32_t lead, int32_t foot_end, int32_t Wi, int32_t Wh, uint32_t *hist0) {
    const uint32_t *cg = rd.cigar + KD_COFF(rd, i);
    const KdChunk *src = reinterpret_cast<const KdChunk *>(KD_SEQ_AT(rd, i));
    uint32_t sg = 0;   // byte offset of the current run's channel group inside a pair
    // Per-lane state machine over WORK UNITS (one 16-byte chunk = up to 32 live bases of the
    // current M run), not over CIGAR ops: lanes whose reads have clips or indels still add
    // bases in the same wavefront instructions as their single-run neighbours.
    int32_t xa = 0, xb = 0, sx = 0, c = 1, cb = 0;   // live query range [xa, xb) of the current run; c > cb: none
    // CIGAR words four at a time (one unaligned 16-byte load, the next four already in flight), and the last
    // 16-byte chunk of bases kept: a long read's runs are a few bases each, so consecutive runs share a chunk
    // and one load per op would make the walk a chain of dependent HBM round trips.
    uint32_t kw = k & ~3u;                     // cw_cur holds words kw .. kw + 3
    KdChunk cw_cur = kd_load_cigar4(cg, kw, nc), cw_nxt = cw_cur;
    if (kw + 4 < nc) cw_nxt = kd_load_cigar4(cg, kw + 4, nc);
    int32_t c_have = -1;
    KdChunk cur = cw_cur;
    for (;;) {
        while (c > cb && k < k_end) {   // advance to the next run with live bases
            if (k >= kw + 4) {
                kw += 4; cw_cur = cw_nxt;
                if (kw + 4 < nc) cw_nxt = kd_load_cigar4(cg, kw + 4, nc);
            }
            const uint32_t kk = k & 3u;
            const uint32_t cw = kk == 0 ? cw_cur.x : kk == 1 ? cw_cur.y : kk == 2 ? cw_cur.z : cw_cur.w;
            const int32_t len = (int32_t)(cw >> 4);
            const uint32_t op = cw & 15u;
            k++;
            if (op == 0 || op == 7 || op == 8) {
                // live query range: inside the run and inside the window
                xa = grel < 0 ? q - grel : q;
                xb = Wi - grel < len ? q + (Wi - grel) : q + len;
                sx = grel - q;                      // site of query base x is sx + x
                sg = 0;
                if (xb > xa) { c = xa >> 5; cb = (xb - 1) >> 5; }
                q += len; grel += len;
                if (grel >= Wi) k = k_end;
            } else if (op == 2) {
                for (int32_t j = grel < 0 ? -grel : 0; j < len && grel + j < Wi; j++)
                    kd_hadd(hist0, KD_HCH_DEL, grel + j);
                grel += len;
                if (grel >= Wi) k = k_end;
            } else if (op == 1) {
                q += len;
            } else if (op == 4) {
                if (k == 1) {
                    // leading clip, kindel.py:64-73: base x -> site r - len + x, kept if >= contig start
                    // (`lead` of the len bases); a run on the clip_end_weights channels
                    const int32_t s_first = grel - len;           // site of base 0
                    xa = -s_first > len - lead ? -s_first : len - lead;
                    xb = Wi - s_first < len ? Wi - s_first : len;
                    sx = s_first; sg = 4u * KD_HCH_CEW;
                    if (xb > xa) { c = xa >> 5; cb = (xb - 1) >> 5; }
                    q += len;
                } else {
                    // non-first clip, kindel.py:74-81: bases q.. -> sites r.. while r < L; for a regular
                    // read it is the last op that moves r, so its reach is the end of the footprint
                    const int32_t n_adv = foot_end - grel;
                    xa = grel < 0 ? q - grel : q;
                    xb = Wi - grel < n_adv ? q + (Wi - grel) : q + n_adv;
                    sx = grel - q; sg = 4u * KD_HCH_CSW;
                    if (xb > xa) { c = xa >> 5; cb = (xb - 1) >> 5; }
                    k = k_end;
                }
            }
        }
        if (c > cb) break;
        if (c != c_have) { cur = src[c]; c_have = c; }
        const int32_t xs = 32 * c;
        kd_add_chunk_masked(hist0, cur, xs, xa, xb, sx, sg);   // [xa, xb) lies inside the run: branch-free masked adds
        c++;
    }
}

// SHORT regular reads with clips / indels (at most KD_PREP_MAX_OPS ops, the bulk of the non-plain reads of a
// short-read batch).  Two phases: (1) decode the CIGAR into at most three SEGMENTS -- runs of query bases that
// land on consecutive sites of one channel group: an M/=/X run, the leading clip (clip_end_weights), the non-first
// clip (clip_start_weights) -- each already cut to the window; deletions are tallied on the way; (2) one flat,
// software-pipelined loop over (segment, 16-byte chunk) steps, every step a branch-free masked add, so that the
// lanes of a wavefront stay in step whatever their op structure.  Returns false (nothing added) when the read
// has more than three segments: the caller then takes the general walk.
__device__ __forceinline__ bool kd_walk_short(const KdReads &rd, kd_u64 i, const KdRInfo ri, kd_u64 wlo, int32_t Wi, int32_t Wh,
                                              uint32_t *hist0) {
    const uint32_t nc = ri.pad >> 24;          // (k_prep: a regular short-CIGAR read's CIGAR length rides in its entry)
    const uint32_t *cg = rd.cigar + KD_COFF(rd, i);
    // the first four CIGAR words, all in flight together (a short read rarely has more)
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    if (nc > 0) w0 = cg[0];
    if (nc > 1) w1 = cg[1];
    if (nc > 2) w2 = cg[2];
    if (nc > 3) w3 = cg[3];
    const KdChunk *src = reinterpret_cast<const KdChunk *>(KD_SEQ_AT(rd, i));
    // (round 6) the read's FIRST chunk of bases is requested here, next to the CIGAR words, not behind their decoding: where it lies
    // does not depend on them, and it is the first chunk the loop below wants for every read the window's left edge does not cut --
    // one memory round trip less in a chain of seven (scripts/exp/kwindow_deletions.sh: the 23 % of C3's reads with a clip or an
    // indel cost k_window 0.28 of its 1.02 ms, 2.8 x a plain read each, and what they wait for is this chain, not their adds)
    const KdChunk first = src[0];
    uint32_t n_seg_ops = 0;
    for (uint32_t k = 0; k < nc; k++) {
        const uint32_t cw = k == 0 ? w0 : k == 1 ? w1 : k == 2 ? w2 : k == 3 ? w3 : cg[k];
        const uint32_t op = cw & 15u;
        n_seg_ops += (op == 0 || op == 7 || op == 8 || op == 4) ? 1u : 0u;
    }
    if (n_seg_ops > 3) return false;
    const kd_u64 gs = ri.gstart, span = ri.span_cls >> KD_SPAN_SHIFT;
    const int32_t lead = (int32_t)ri.lead;
    const int32_t foot_end = (int32_t)((uint32_t)(gs + span) - (uint32_t)wlo);  // window-relative end of the footprint
    int32_t grel = (int32_t)((uint32_t)gs - (uint32_t)wlo);                        // window-relative site, may be negative
    int32_t q = 0;
    // segment slots: live query range [a, b), site offset x (site of query base j is x + j), channel-group byte offset g
    int32_t a0 = 0, b0 = 0, x0 = 0, a1 = 0, b1 = 0, x1 = 0, a2 = 0, b2 = 0, x2 = 0;
    uint32_t g0 = 0, g1 = 0, g2 = 0;
    uint32_t ns = 0;
    for (uint32_t k = 0; k < nc; k++) {
        const uint32_t cw = k == 0 ? w0 : k == 1 ? w1 : k == 2 ? w2 : k == 3 ? w3 : cg[k];
        const int32_t len = (int32_t)(cw >> 4);
        const uint32_t op = cw & 15u;
        int32_t xa = 0, xb = 0, sx = 0;
        uint32_t sg = 0;
        if (op == 0 || op == 7 || op == 8) {
            xa = grel < 0 ? q - grel : q;
            xb = Wi - grel < len ? q + (Wi - grel) : q + len;
            sx = grel - q;
            q += len; grel += len;
        } else if (op == 2) {
            for (int32_t j = grel < 0 ? -grel : 0; j < len && grel + j < Wi; j++)
                kd_hadd(hist0, KD_HCH_DEL, grel + j);
            grel += len;
        } else if (op == 1) {
            q += len;
        } else if (op == 4) {
            if (k == 0) {   // leading clip, kindel.py:64-73: base j -> site r - len + j, the last `lead` bases are kept
                const int32_t s_first = grel - len;
                xa = -s_first > len - lead ? -s_first : len - lead;
                xb = Wi - s_first < len ? Wi - s_first : len;
                sx = s_first; sg = 4u * KD_HCH_CEW;
                q += len;
            } else {        // non-first clip, kindel.py:74-81: it is the last op that moves r (regular read)
                const int32_t n_adv = foot_end - grel;
                xa = grel < 0 ? q - grel : q;
                xb = Wi - grel < n_adv ? q + (Wi - grel) : q + n_adv;
                sx = grel - q; sg = 4u * KD_HCH_CSW;
                k = nc;
            }
        }
        if (xb > xa) {
            if (ns == 0) { a0 = xa; b0 = xb; x0 = sx; g0 = sg; }
            else if (ns == 1) { a1 = xa; b1 = xb; x1 = sx; g1 = sg; }
            else { a2 = xa; b2 = xb; x2 = sx; g2 = sg; }
            ns++;
        }
        if (grel >= Wi) break;   // everything further right is outside the window
    }
    if (ns == 0) return true;
    int32_t c = a0 >> 5, cb = (b0 - 1) >> 5;
    KdChunk cur = first;
    if (c != 0) cur = src[c];
    for (;;) {
        // the step after this one: next chunk of the segment, or the first chunk of the next segment
        const bool adv = c + 1 > cb;
        const bool more = !adv || ns > 1;
        const int32_t cn = adv ? (a1 >> 5) : c + 1;
        KdChunk nxt = cur;
        if (more && cn != c) nxt = src[cn];      // (a segment that starts in the chunk its predecessor ended in: the chunk is here)
        kd_add_chunk_masked(hist0, cur, 32 * c, a0, b0, x0, g0);
        if (!more) break;
        if (adv) {
            a0 = a1; b0 = b1; x0 = x1; g0 = g1; a1 = a2; b1 = b2; x1 = x2; g1 = g2;
            ns--;
            cb = (b0 - 1) >> 5;
        }
        c = cn;
        cur = nxt;
    }
    return true;
}

// A PLAIN read: one M/=/X run covering the whole read, no clips (k_prep: KD_INFO_PLAIN).  Nothing to decode:
// query base x lands on site grel + x, for x in [0, span).  ROWS: the ROW of a long read (kd_long.h), symbol x on site grel + x.
template <bool ROWS = false>
__device__ __forceinline__ void kd_walk_plain(const KdReads &rd, kd_u64 i, const KdRInfo ri, kd_u64 wlo, int32_t Wi,
                                              int32_t Wh, uint32_t *hist0) {
    const int32_t grel = (int32_t)(ri.gstart - (uint32_t)wlo);
    const int32_t len = (int32_t)(ri.span_cls >> KD_SPAN_SHIFT);
    const int32_t xa = grel < 0 ? -grel : 0;
    const int32_t xb = Wi - grel < len ? Wi - grel : len;
    if (xb <= xa) return;
    const KdChunk *src = reinterpret_cast<const KdChunk *>(KD_SEQ_AT(rd, i));
    const int32_t ca = xa >> 5, cb = (xb - 1) >> 5;
    // three chunks of prefetch: a 150-base read is 5 chunks, so its loads are (almost) all in flight at once
    KdChunk cur = src[ca], n1 = cur, n2 = cur;
    if (ca + 1 <= cb) n1 = src[ca + 1];
    if (ca + 2 <= cb) n2 = src[ca + 2];
    for (int32_t c = ca; c <= cb; c++) {
        KdChunk n3 = n2;
        if (c + 3 <= cb) n3 = src[c + 3];
        const int32_t xs = 32 * c;
        kd_add_dword<ROWS>(hist0, cur.x, xs, xa, xb, 0, len, grel, 0u);
        kd_add_dword<ROWS>(hist0, cur.y, xs + 8, xa, xb, 0, len, grel, 0u);
        kd_add_dword<ROWS>(hist0, cur.z, xs + 16, xa, xb, 0, len, grel, 0u);
        kd_add_dword<ROWS>(hist0, cur.w, xs + 24, xa, xb, 0, len, grel, 0u);
        cur = n1; n1 = n2; n2 = n3;
    }
}

// The ROW of a long read (kd_long.h) against the window: symbol x on site grel + x, one plain run thousands of sites long.
// Every row of the wavefront covers the whole window, so lanes walking in step would all add to the same few counters in every
// instruction (same-address LDS atomics serialise: 53 ns per instruction with 64 lanes on one address,
// profiles/valu_issue_calibration.json).  Lane `rot` therefore starts at chunk (rot mod chunks) of the window's part of its
// row, at dword (rot / chunks) mod 4 of every chunk, and wraps around: the lanes of a wavefront sit on different sites.
__device__ __forceinline__ void kd_walk_row(const KdReads &rd, kd_u64 i, const KdRInfo ri, kd_u64 wlo, int32_t Wi, uint32_t rot,
                                            uint32_t *hist0) {
    const int32_t grel = (int32_t)(ri.gstart - (uint32_t)wlo);
    const int32_t len = (int32_t)(ri.span_cls >> KD_SPAN_SHIFT);
    const int32_t xa = grel < 0 ? -grel : 0;
    const int32_t xb = Wi - grel < len ? Wi - grel : len;
    if (xb <= xa) return;
    const KdChunk *src = reinterpret_cast<const KdChunk *>(KD_SEQ_AT(rd, i));
    const int32_t ca = xa >> 5, n = ((xb - 1) >> 5) - ca + 1;
    int32_t cpos = (int32_t)(rot % (uint32_t)n);
    const uint32_t o = (rot / (uint32_t)n) & 3u;
    // (one chunk ahead.  Round 6 measured THREE, as kd_walk_plain has them -- a window's part of a row is 14 chunks, i.e. 14 dependent
    // round trips per window with three wavefronts' worth of rows to hide them: k_window_rows 0.141 -> 0.147 ms on C5, dropped.)
    KdChunk cur = src[ca + cpos];
    for (int32_t t = 0; t < n; t++) {
        const int32_t c = ca + cpos;
        cpos = cpos + 1 < n ? cpos + 1 : 0;
        KdChunk nxt = cur;
        if (t + 1 < n) nxt = src[ca + cpos];
        // the chunk's four dwords, starting at dword o
        const uint32_t d0 = o == 0 ? cur.x : o == 1 ? cur.y : o == 2 ? cur.z : cur.w;
        const uint32_t d1 = o == 0 ? cur.y : o == 1 ? cur.z : o == 2 ? cur.w : cur.x;
        const uint32_t d2 = o == 0 ? cur.z : o == 1 ? cur.w : o == 2 ? cur.x : cur.y;
        const uint32_t d3 = o == 0 ? cur.w : o == 1 ? cur.x : o == 2 ? cur.y : cur.z;
        const int32_t xs = 32 * c;
        kd_add_dword<true>(hist0, d0, xs + 8 * (int32_t)(o & 3u), xa, xb, 0, len, grel, 0u);
        kd_add_dword<true>(hist0, d1, xs + 8 * (int32_t)((o + 1u) & 3u), xa, xb, 0, len, grel, 0u);
        kd_add_dword<true>(hist0, d2, xs + 8 * (int32_t)((o + 2u) & 3u), xa, xb, 0, len, grel, 0u);
        kd_add_dword<true>(hist0, d3, xs + 8 * (int32_t)((o + 3u) & 3u), xa, xb, 0, len, grel, 0u);
        cur = nxt;
    }
}

// The 8 bases of dword v through precomputed pointers (h: pair of the dword's first base + channel group, hq = h + one pair if
// that base sits on an odd site) and add values: what kd_add8_full does after its address arithmetic.
__device__ __forceinline__ void kd_add8_ptr(unsigned char *h, unsigned char *hq, uint32_t v, uint32_t vp, uint32_t vq) {
    uint32_t rh, rl;
    kd_codes8(v, rh, rl);
#pragma unroll
    for (int b = 0; b < 8; b++) {
#ifdef KD_EXP_SKIP_BASES   // (TIMING EXPERIMENT only, wrong tables: the last n bases of every dword are not added -- what removing n / 8 of the hot path's
        if (b >= 8 - KD_EXP_SKIP_BASES) continue;   //  ds_add instructions could buy at most, scripts/exp/kwindow_fewer_adds.sh)
#endif
        const uint32_t code = (((b & 1) ? rl : rh) >> (8 * (b >> 1))) & 0xffu;
        unsigned char *a = ((b & 1) ? hq : h) + code + KD_HPITCHB * (b >> 1);
        atomicAdd(reinterpret_cast<uint32_t *>(a), (b & 1) ? vq : vp);
    }
}
// A plain read that lies INSIDE the window (k_window's first list): no window clipping, and for a read of up to 160 bases no
// loop either -- its (up to) five chunks are requested together, every chunk before the last is four whole dwords whose
// counters sit at compile-time offsets from one pointer computed per read.
// ROT (a DEEP tile: dozens of reads start on every site, so the lanes of a wavefront -- `rows` reads apart -- still sit on the same
// handful of sites and in step would add to the same counters in every instruction; same-address LDS atomics serialise): a read
// of four whole chunks (129 - 160 bases: every short-read library) is walked from chunk `rot & 3` round, the partial fifth chunk
// last, so that lanes of different `rot` are 32 sites apart.  Costs a pointer computation per chunk: only where it pays.
template <bool ROT>
__device__ __forceinline__ void kd_walk_inner(const KdReads &rd, kd_u64 i, const KdRInfo ri, kd_u64 wlo, int32_t Wi,
                                              int32_t Wh, uint32_t *hist0, uint32_t rot) {
    const int32_t len = (int32_t)(ri.span_cls >> KD_SPAN_SHIFT);
    const int32_t cb = (len - 1) >> 5;
    if (cb > 4) { kd_walk_plain(rd, i, ri, wlo, Wi, Wh, hist0); return; }
    const int32_t grel = (int32_t)(ri.gstart - (uint32_t)wlo);   // 0 <= grel, grel + len <= W
#if KD_EXP_DELETE & 128    // (TIMING EXPERIMENT: every lane of the wavefront fetches the FIRST lane's read -- the same five requests, one line each instead of 64)
    const kd_u64 so_ = KD_SOFF(rd, i);
    const KdChunk *src = reinterpret_cast<const KdChunk *>(rd.seq4 + (((kd_u64)kd_readfirstlane((uint32_t)(so_ >> 32)) << 32) | kd_readfirstlane((uint32_t)so_)));
#else
    const KdChunk *src = reinterpret_cast<const KdChunk *>(KD_SEQ_AT(rd, i));
#endif
    const uint32_t r4 = (ROT && cb == 4) ? rot & 3u : 0u;        // slot c holds chunk (c + r4) & 3 (slots 0-3 of a read with a fifth chunk)
    const uint32_t c0 = r4, c1 = ROT ? (1u + r4) & 3u : 1u, c2 = ROT ? (2u + r4) & 3u : 2u, c3 = ROT ? (3u + r4) & 3u : 3u;
#if KD_EXP_DELETE & 2      // (TIMING EXPERIMENT: no base loads on this path -- lane-varying constants instead)
    KdChunk k0; { const uint32_t z = (uint32_t)i * 0x9E3779B1u; k0.x = z; k0.y = z * 3u; k0.z = z * 5u; k0.w = z * 7u; (void)src; }
    KdChunk k1 = k0, k2 = k0, k3 = k0, k4 = k0;
#else
    KdChunk k0 = src[c0], k1 = k0, k2 = k0, k3 = k0, k4 = k0;
    if (cb >= 1) k1 = src[c1];
    if (cb >= 2) k2 = src[c2];
    if (cb >= 3) k3 = src[c3];
    if (cb >= 4) k4 = src[4];
#endif
#if KD_EXP_DELETE & 1      // (TIMING EXPERIMENT: the loads stay, nothing is added)
    asm volatile("" :: "v"(k0.x), "v"(k0.w), "v"(k1.x), "v"(k1.w), "v"(k2.x), "v"(k2.w), "v"(k3.x), "v"(k3.w), "v"(k4.x), "v"(k4.w));
    (void)Wh; (void)hist0; (void)c0; (void)c1; (void)c2; (void)c3;
    return;
#endif
    const int32_t p = grel & 1;
    unsigned char *h = reinterpret_cast<unsigned char *>(hist0) + KD_MUL24S(grel >> 1, KD_HPITCHB);
    unsigned char *hq = h + KD_HPITCHB * p;
    const uint32_t vp = 1u << (16 * p), vq = 0x10000u >> (16 * p);
#define KD_INNER_STAGE(kc, c, cc)                                                                        \
    if ((c) < cb) {                                                                                      \
        unsigned char *hc = ROT ? h + KD_MUL24((cc), 16 * KD_HPITCHB) : h + 16 * (c) * KD_HPITCHB;       \
        unsigned char *hqc = ROT ? hq + KD_MUL24((cc), 16 * KD_HPITCHB) : hq + 16 * (c) * KD_HPITCHB;    \
        kd_add8_ptr(hc, hqc, kc.x, vp, vq);                                                              \
        kd_add8_ptr(hc + 4 * KD_HPITCHB, hqc + 4 * KD_HPITCHB, kc.y, vp, vq);                            \
        kd_add8_ptr(hc + 8 * KD_HPITCHB, hqc + 8 * KD_HPITCHB, kc.z, vp, vq);                            \
        kd_add8_ptr(hc + 12 * KD_HPITCHB, hqc + 12 * KD_HPITCHB, kc.w, vp, vq);                          \
    } else if ((c) == cb) {                                                                              \
        kd_add_dword(hist0, kc.x, 32 * (c), 0, len, 0, len, grel, 0u);                                   \
        kd_add_dword(hist0, kc.y, 32 * (c) + 8, 0, len, 0, len, grel, 0u);                               \
        kd_add_dword(hist0, kc.z, 32 * (c) + 16, 0, len, 0, len, grel, 0u);                              \
        kd_add_dword(hist0, kc.w, 32 * (c) + 24, 0, len, 0, len, grel, 0u);                              \
    }
    KD_INNER_STAGE(k0, 0, c0)
    KD_INNER_STAGE(k1, 1, c1)
    KD_INNER_STAGE(k2, 2, c2)
    KD_INNER_STAGE(k3, 3, c3)
    KD_INNER_STAGE(k4, 4, 4u)
#undef KD_INNER_STAGE
}

#ifndef KD_DEEP_READS_PER_SITE
#define KD_DEEP_READS_PER_SITE 12u   // a tile is DEEP when it holds this many candidates per start site
#endif
#ifndef KD_TILE
#define KD_TILE 1024   // reads classified together (a multiple of KD_BLOCK)
#endif
#define KD_TILE_PER_THREAD (KD_TILE / KD_BLOCK)
static_assert(KD_TILE % KD_BLOCK == 0 && KD_TILE <= 1024, "a list entry holds the tile-relative index of a read in 10 bits (the complex list: 14, two flag bits above)");
#ifndef KD_LIST_CARRY
#define KD_LIST_CARRY 1   // list 0 carries every entry's window-relative start and length (no second fetch of its footprint record)
#endif
#define KD_CMETA 352u    // complex entries whose footprint the list carries too (what the LDS of five workgroups per CU has room for)
#define KD_WINDOW_LDS_BYTES(Wh) ((size_t)KD_HCH * (Wh) * 4 + (size_t)KD_TILE * 4 + (size_t)KD_TILE * 2 + (size_t)KD_CMETA * 4)   // Wh = site pairs, halos included; lists: u32 + u16 per tile slot + the complex entries' footprints

#define KD_WINDOW_OCC 5   // workgroups per CU the register budget is set for (= what the LDS footprint allows; 6 / 7 measured: slower, scratch)
// k_window's WORK QUEUE (round 4: the kernel plans for itself).  Rounds 1 - 3 planned in two kernels of their own -- a binary
// search per window for its candidate range (k_plan_ranges, 24 us on C3: 24 dependent loads), a one-workgroup scan of the
// per-window item counts that also wrote an item -> window table (k_plan_scan, 29 us) -- a twentieth of the step on C3 and
// a tenth on a small input (C2, an eighth of C3 on one of eight GPUs).  Now:
//   * the candidate range of a window comes from a BOUNDARY TABLE: first entry whose start lies at or behind site j * gran.
//     For a sorted batch k_prep writes it as it classifies the reads (gran 64: a lane whose read starts in a later granule
//     than its predecessor's fills the granules in between); for a bucket-sorted one it is the bin offsets of the sort (gran W);
//   * a workgroup takes a WINDOW TICKET, reads its range from the table and tallies the window's first `slice` candidates.  A
//     window with more (a deep small genome: C2 has 23 windows of 29 000 reads) goes onto the HOT LIST with a slice counter;
//     its owner keeps taking slices from the counter, and workgroups that find the tickets gone wait until every ticket holder
//     has published (a count: owners publish right after the dequeue, before they tally) and then help with the hot windows.
//     No scan, no item table, nothing sized by an upper bound on the items.
struct alignas(16) KdHot { uint32_t w, next, K, pad; kd_u64 lo, hi; };   // a window with K > 1 slices: next slice to take, its candidate range
struct KdWq {
    const uint32_t *bound32;   // k_prep's table (gran 64), or NULL:
    const kd_u64 *bound64;     //   the bin offsets of the bucket sort (entry b * reps = first slot of bin b)
    uint32_t gran, reps;
    uint32_t nb;               // last valid index of the table (= its entry for "behind the last site"; k_prep's table: its last WRITTEN entry)
    uint32_t jlo;              // first valid index (k_prep leaves the granules in front of the batch's first read unwritten: they read as this entry)
    KdHot *hot;                // [n_win]
    uint32_t n_win, span_slot;
    uint32_t cut;              // > 0: STATIC queue -- workgroup b tallies part (b mod cut) of window (b / cut), see k_window
};
__device__ __forceinline__ kd_u64 kd_wq_bound(const KdWq &Q, kd_u64 j) {
    j = j < Q.nb ? j : Q.nb;
    j = j > Q.jlo ? j : Q.jlo;
    return Q.bound32 ? (kd_u64)Q.bound32[j] : Q.bound64[j * Q.reps];
}
// candidate range of local window w: entries that start in [wlo - back, whi + maxlead), `back` = what an entry in front of the
// window can have left for it (OWNERSHIP below: nothing unless it is longer than H -- except for the plan's first window)
// (maxspan / maxlead: the batch's longest footprint / leading clip -- status[Q.span_slot], status[KDS_B_MAXLEAD], read once per workgroup)
__device__ __forceinline__ void kd_wq_range(const KdWq &Q, kd_u64 maxspan, kd_u64 maxlead, uint32_t w0, uint32_t w, uint32_t W, uint32_t H,
                                            kd_u64 &lo, kd_u64 &hi) {
    kd_u64 back = maxspan;
    if (w) back = back > H ? back - H : 0;
    const kd_u64 wlo = (kd_u64)(w0 + w) * W, whi = wlo + W;
    lo = kd_wq_bound(Q, (wlo > back ? wlo - back : 0) / Q.gran);
    hi = kd_wq_bound(Q, (whi + maxlead + Q.gran - 1) / Q.gran);
    if (hi < lo) hi = lo;      // (an unsorted batch's table is meaningless -- and unused; never a negative range)
}

// THE COLD TAIL (round 5).  The launch's persistent workgroups leave one by one as the window tickets run out, each after its last
// window (~0.1 ms of work at full size): for half a window's time on average a workgroup slot idles until the launch ends.  The
// clip counters and insertion events of the clipped / inserted reads (k_cold_lane: 0.12 ms of memory-bound work that needs
// nothing k_window produces) ride in the same launch as workgroups BEHIND the persistent ones -- `n_regions` more, one per
// record region: the dispatcher hands them out as slots come free, they fill the tail, and k_cold_lane's own launch is gone.
// (They hold no ticket and nobody waits for them; a slot taken by one before every persistent workgroup is resident delays that
// workgroup, nothing else.)  The batch's error classification, which must see every kernel's flags, is a launch of its own then.
// Measured: C3 step 1.589 -> 1.565 ms (k_window + k_cold_lane 0.932 + 0.131 -> 1.031), C4 2.858 -> 2.790, C2 0.253 -> 0.250, same FASTA,
// 359 parity tests.  Opt-in (KD_COLD_TAIL=1; kd_engine.h says why it is not the default): profiles/r05_cold_tail_ab.json.
struct KdColdTail {
    const KdColdRec *rec; const uint32_t *cnt; const kd_u64 *evbase, *poolbase;
    KdIns ins;
    uint32_t region_slots, n_regions, first_block;     // n_regions = 0: no tail; first_block = the persistent workgroups
};
template <bool ROWS>
__global__ void __launch_bounds__(KD_BLOCK, KD_WINDOW_OCC)   // 5 wavefronts per SIMD = the 5 workgroups per CU the LDS footprint allows
k_window(KdReads rd, const KdRInfo *rinfo, const uint32_t *order, KdTabs T, KdWq Q, uint32_t w0,
         uint32_t W, uint32_t H, uint32_t Wh_, uint32_t slice, kd_u64 *status, KdColdTail tail) {
    if (!ROWS && tail.n_regions && blockIdx.x >= tail.first_block) {     // (uniform over the workgroup)
        kd_cold_region(blockIdx.x - tail.first_block, rd, T, tail.ins, tail.rec, tail.cnt, tail.evbase, tail.poolbase, tail.region_slots, status);
        return;
    }
    // OWNERSHIP (round 3).  The histogram of window w covers the sites [wlo, whi + H): H sites more than the window.  An entry
    // is tallied by the window its START lies in, over [start, min(end, whi + H)) -- with H >= the longest footprint of the
    // batch that is the whole read, once, on the loop-free walk, whatever window edge it crosses (before, every read that
    // crossed an edge was walked twice, both times on the clipped path: 1.23 visits per read on the bench workload, the
    // crossing ones with the worst lane occupancy).  What a longer entry leaves over is tallied by the windows behind:
    // window w > s takes [max(start, wlo + H), min(end, whi + H)) ("early" entries: a shifted origin, the general walk), and
    // the reach of a leading clip back into the windows in FRONT of its read's start window is tallied there over
    // [wlo, whi) ("late" entries).  The flush adds the H extra sites to the tables like the others: table counters are
    // sums of work items anyway.
    // ROWS = false: `rinfo` describes the batch's reads (first pass, class REG = short regular reads).
    // ROWS = true (second pass): `rinfo` / rd.seq_off / rd.seq4 describe the ROWS of the batch's long reads (kd_long.h: one
    // symbol per site, the symbol is the LDS channel; k_long_reduce's entries are PLAIN runs thousands of sites long); H = 0:
    // every window tallies what lies inside it of every row that crosses it, on the plain walk; `order` is never NULL.
    // LDS channels of a row pass: nothing (0), A,T,G,C,N (1-5), deleted (6), the same seven with an insertion in front (7-13).
    KD_DYN_SHARED(uint32_t, hist);
    const int32_t Wh = (int32_t)Wh_;   // dwords per channel row (two u16 counters each, halos included; >= (W + 2*KD_HALO)/2)
    uint32_t *hist0 = hist + (KD_HALO / 2) * KD_HPITCH;   // pair of window-relative site 0
    // list entries: tile-relative read indices; list 0 (plain reads inside the histogram) also carries the read's window-relative
    // start (bits 10-19) and length (bits 20-31), which the classification has in registers: the walker does not fetch the
    // footprint record a second time
    uint32_t *l_plain = hist + (size_t)KD_HCH * Wh;
    uint16_t *l_cplx = reinterpret_cast<uint16_t *>(l_plain + KD_TILE);
    // the first KD_CMETA complex entries' footprints: window-relative start (bits 0-9), span (10-19), leading clip's reach (20-26),
    // CIGAR words (27-31); ~0: does not fit, the walker fetches the record
    uint32_t *l_cmeta = reinterpret_cast<uint32_t *>(l_cplx + KD_TILE);
    __shared__ kd_u64 s_first, s_last;
    __shared__ uint32_t s_win;
    // [tile parity][list]: 0 own plain entries that end inside the histogram (l_plain from the front), 1 own plain entries cut by
    // its end (l_plain from the back), 2 own complex entries (l_cplx from the front), 3 early / late entries (l_cplx from the back)
    __shared__ uint32_t s_cnt[2][4];
    __shared__ uint32_t s_row[2];       // next row of the tile to hand out
    __shared__ uint32_t s_gfirst[2], s_glast[2];   // G-starts of the tile's first and last candidate (a DEEP tile: see kd_walk_inner<true>)
    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & (KD_WAVE - 1);
    const uint32_t nh = (uint32_t)KD_HCH * (uint32_t)Wh;   // histogram dwords
    const int32_t Wi = (int32_t)W, We = (int32_t)(W + H);     // the window / the histogram's reach, in sites
    uint32_t *hist_early = hist0 + (H / 2) * KD_HPITCH;       // pair of site wlo + H: origin of the early entries' walk (H is even)
    KD_PHASE_DECL      // (profiling hooks, empty in the product: kd_common.h)
    // thread 0's view of the queue, kept across items -- in LDS, not in registers: every lane would carry them, and the walk
    // needs its 96
    const uint32_t NONE = 0xffffffffu, SCAN = 0xfffffffeu;
    __shared__ uint32_t q_hot_w, q_hot_j, q_hot_K;      // the hot window this workgroup is taking slices from
    __shared__ kd_u64 q_hot_lo, q_hot_hi;
    __shared__ uint32_t q_state, q_nhot, s_found;       // q_state: 0 window tickets may be left, 1 none left, 2 helping
    __shared__ kd_u64 q_maxspan, q_maxlead;       // the batch's longest footprint / leading clip (read once: two loads less per dequeue)
    if (t == 0) {
        q_hot_w = NONE; q_state = 0;
        q_maxspan = status[Q.span_slot]; q_maxlead = status[KDS_B_MAXLEAD];
    }
    for (;;) {
        if (t == 0) {
            uint32_t w = NONE, k = 0;
            kd_u64 lo = 0, hi = 0;
            if (Q.cut) {
                // STATIC queue (round 4): few windows, many workgroups (a deep small genome: C2 is 23 windows of 29 000 reads on
                // 1280 workgroups).  The hot list below served it -- and its wavefronts spent 47 % of their clocks in this loop
                // (phase clocks, profiles/r04_kwindow_experiments.json): 1280 tickets for 23 windows, everybody polling the
                // publication count, one contended counter per window.  With `cut` workgroups per window nothing needs to be
                // negotiated: workgroup b takes part (b mod cut) of window (b / cut) -- its candidate range from the boundary
                // table, cut evenly -- in pieces of at most `slice` (u16 counters), and leaves.
                if (q_state == 0) {
                    q_state = 3;
                    const uint32_t b = blockIdx.x;
                    if (b < Q.n_win * Q.cut) {
                        kd_wq_range(Q, q_maxspan, q_maxlead, w0, b / Q.cut, W, H, lo, hi);
                        const kd_u64 part = (((hi - lo + Q.cut - 1) / Q.cut) + 63) & ~(kd_u64)63;
                        const kd_u64 a = lo + (kd_u64)(b % Q.cut) * part;
                        q_hot_w = b / Q.cut; q_hot_lo = a < hi ? a : hi; q_hot_hi = a + part < hi ? a + part : hi;
                        if (b == 0) atomicAdd(&status[KDS_TOTAL_ITEMS], (kd_u64)Q.n_win * (Q.cut - 1u));      // (statistics: kd_get_batch_info)
                    }
                }
                if (q_hot_w != NONE && q_hot_lo < q_hot_hi) {
                    w = q_hot_w; lo = q_hot_lo; hi = q_hot_hi; k = 0;    // [lo, min(lo + slice, hi)) now, the rest on the next visit
                    q_hot_lo = lo + slice < hi ? lo + slice : hi;
                }
            } else
            for (;;) {
                if (q_hot_w != NONE) {                   // a window with several slices: the next one nobody has taken
                    k = atomicAdd(&Q.hot[q_hot_j].next, 1u);
                    if (k + 1u == q_hot_K) atomicAdd(&status[KDS_WQ_LEFT], ~0ULL);     // the window's last slice is taken
                    if (k < q_hot_K) { w = q_hot_w; lo = q_hot_lo; hi = q_hot_hi; break; }
                    q_hot_w = NONE;
                    continue;
                }
                if (q_state == 0) {
                    // (a ticket taken one item AHEAD, the next in flight while this window's range is looked up, was measured: +2.5 %
                    // -- a holder then publishes only when it reaches its window, and everybody else polls that much longer)
                    const kd_u64 tk = atomicAdd(&status[KDS_WQ_TICKET], 1ULL);
                    if (tk >= Q.n_win) { q_state = 1; continue; }
                    kd_wq_range(Q, q_maxspan, q_maxlead, w0, (uint32_t)tk, W, H, lo, hi);
                    const kd_u64 K = (hi - lo + slice - 1) / slice;
                    if (K > 1) {                         // publish the window for helpers; slice 0 is this workgroup's
                        const uint32_t j = (uint32_t)atomicAdd(&status[KDS_WQ_HOT], 1ULL);
                        KdHot e;
                        e.w = (uint32_t)tk; e.next = 1u; e.K = (uint32_t)(K < 0xfffffff0ULL ? K : 0xfffffff0ULL); e.pad = 0; e.lo = lo; e.hi = hi;
                        Q.hot[j] = e;
                        q_hot_w = e.w; q_hot_j = j; q_hot_K = e.K; q_hot_lo = lo; q_hot_hi = hi;
                        atomicAdd(&status[KDS_WQ_LEFT], 1ULL);
                        atomicAdd(&status[KDS_TOTAL_ITEMS], K - 1);      // (statistics: kd_get_batch_info)
                        __threadfence();
                    }
                    atomicAdd(&status[KDS_WQ_PUB], 1ULL);
                    if (K == 0) continue;                // nothing starts near this window
                    w = (uint32_t)tk; k = 0;
                    break;
                }
                if (q_state == 1) {                      // every ticket is taken: wait until their holders have published
                    while (kd_ld_acquire(&status[KDS_WQ_PUB]) < (kd_u64)Q.n_win) kd_spin_pause();
                    q_nhot = (uint32_t)kd_ld_acquire(&status[KDS_WQ_HOT]);
                    q_state = 2;
                }
                // helping: as long as some hot window has slices nobody has taken, look for one (all threads: below)
                if (q_nhot != 0 && kd_ld_acquire(&status[KDS_WQ_LEFT]) != 0) w = SCAN;
                break;                                   // (w == NONE: done)
            }
            s_win = w;
            s_first = lo + (kd_u64)k * slice;
            s_last = lo + (kd_u64)k * slice + slice < hi ? lo + (kd_u64)k * slice + slice : hi;
            s_cnt[0][0] = 0; s_cnt[0][1] = 0; s_cnt[0][2] = 0; s_cnt[0][3] = 0; s_row[0] = 0;
            s_found = NONE;
        }
        __syncthreads();
        if (s_win == SCAN) {
            // the hot list, KD_BLOCK entries at a time (workgroups start at different entries and go round): the first one with a
            // slice left; thread 0 then takes slices from it like its owner does
            const uint32_t nhot = q_nhot;
            for (uint32_t base = 0; base < nhot; base += KD_BLOCK) {      // (uniform trip count; two barriers per round)
                const uint32_t x = base + t;
                if (x < nhot) {
                    const KdHot *e = &Q.hot[(blockIdx.x + x) % nhot];
                    if (*(volatile const uint32_t *)&e->next < *(volatile const uint32_t *)&e->K) atomicMin(&s_found, x);
                }
                __syncthreads();
                const bool hit = s_found != NONE;
                __syncthreads();
                if (hit) break;
            }
            if (t == 0) {
                if (s_found == NONE) q_nhot = 0;         // nothing left anywhere: the next look at the queue says "done"
                else {
                    const uint32_t j = (blockIdx.x + s_found) % nhot;
                    const KdHot e = Q.hot[j];
                    q_hot_w = e.w; q_hot_j = j; q_hot_K = e.K; q_hot_lo = e.lo; q_hot_hi = e.hi;
                }
            }
            __syncthreads();
            continue;
        }
        __syncthreads();
        const uint32_t w = s_win;
        if (w == NONE) break;
        KD_MARK(c_deq)
        const kd_u64 wlo = (kd_u64)(w0 + w) * W, whi = wlo + W;
        const kd_u64 first = s_first, last = s_last;
        // The classification keys (start, span | flags, lead) of a tile are fetched ONE TILE AHEAD into registers:
        // the loads of tile k + 1 are in flight while tile k is walked.  `order`: bucket-sorted permutation.
        uint32_t p_gs[KD_TILE_PER_THREAD], p_sc[KD_TILE_PER_THREAD], p_ld[KD_TILE_PER_THREAD];   // (p_ld: the leading clip's reach, the CIGAR word count in its top byte)
#pragma unroll
        for (uint32_t u = 0; u < KD_TILE_PER_THREAD; u++) {
            const kd_u64 j = first + u * KD_BLOCK + t;
            p_sc[u] = KD_CLS_SKIP; p_gs[u] = 0; p_ld[u] = 0;
            if (j < last && !(KD_EXP_DELETE & 64)) {      // (& 64: TIMING EXPERIMENT, the tiles' keys are not fetched)
                const KdRInfo ri = KD_RI(rinfo, rd, order ? (kd_u64)order[j] : j);
                p_gs[u] = ri.gstart; p_sc[u] = ri.span_cls; p_ld[u] = (ri.lead & 0xffffffu) | (ri.pad & 0xff000000u);
            }
        }
        {   // Wh is a multiple of 4 (W is a multiple of 64): zero with 16-byte stores
            uint4 *h4 = reinterpret_cast<uint4 *>(hist);
            for (uint32_t x = t; x < ((KD_EXP_DELETE & 16) ? 0u : nh / 4); x += KD_BLOCK) h4[x] = make_uint4(0u, 0u, 0u, 0u);   // (& 16: TIMING EXPERIMENT, no zeroing)
        }
        KD_MARK(c_zero)
        uint32_t par = 0;
        for (kd_u64 tb = first; tb < last; tb += KD_TILE, par ^= 1u) {
            // classify the tile's reads: plain (single aligned run) / complex; drop those outside the window
#pragma unroll
            for (uint32_t u = 0; u < KD_TILE_PER_THREAD; u++) {
                if (KD_EXP_DELETE & 32) continue;      // (TIMING EXPERIMENT: nothing is classified, the lists stay empty)
                const kd_u64 gs = p_gs[u], span = p_sc[u] >> KD_SPAN_SHIFT;
                if (!ROWS) {
                    const kd_u64 j = tb + u * KD_BLOCK + t;
                    if (j == tb) s_gfirst[par] = (uint32_t)gs;
                    if (j + 1 == (tb + KD_TILE < last ? tb + KD_TILE : last)) s_glast[par] = (uint32_t)gs;
                }
                if (ROWS) {
                    if ((p_sc[u] & 3u) == KD_CLS_REG && gs < whi && gs + span > wlo)
                        l_plain[atomicAdd(&s_cnt[par][0], 1u)] = u * KD_BLOCK + t;
                } else if ((p_sc[u] & 3u) == KD_CLS_REG) {
                    const uint32_t rel = u * KD_BLOCK + t;
                    if (gs >= wlo && gs < whi) {             // starts here: this window's own
                        if (!(p_sc[u] & KD_INFO_PLAIN)) {
                            const uint32_t slot = atomicAdd(&s_cnt[par][2], 1u);
                            l_cplx[slot] = (uint16_t)rel;
#if KD_LIST_CARRY
                            if (slot < KD_CMETA) {
                                const uint32_t nc = p_ld[u] >> 24, ld = p_ld[u] & 0xffffffu;
                                l_cmeta[slot] = (gs - wlo < 1024u && span < 1024u && ld < 128u && nc < 32u)
                                                    ? (uint32_t)(gs - wlo) | (uint32_t)span << 10 | ld << 20 | nc << 27 : 0xffffffffu;
                            }
#endif
                        }
                        else if (gs + span <= whi + H)      // (a start or a length beyond the fields -- a hand-picked window of more than 1024 sites -- is not carried: length 0)
                            l_plain[atomicAdd(&s_cnt[par][0], 1u)] = (gs - wlo < 1024u && span < 4096u) ? rel | (uint32_t)(gs - wlo) << 10 | (uint32_t)span << 20 : rel;
                        else l_plain[KD_TILE - 1u - atomicAdd(&s_cnt[par][1], 1u)] = rel;
                    } else if (gs < wlo) {                   // starts in a window in front: what its owner(s) left of it for this one
                        // (the FIRST window of a shard's plan has no window in front: it takes such entries from its own first site)
                        if (gs + span > wlo + (w ? H : 0u)) l_cplx[KD_TILE - 1u - atomicAdd(&s_cnt[par][3], 1u)] = (uint16_t)(rel | 0x8000u);
                    } else if (gs - (p_ld[u] & 0xffffffu) < whi) {         // starts behind: its leading clip reaches back into this window
                        l_cplx[KD_TILE - 1u - atomicAdd(&s_cnt[par][3], 1u)] = (uint16_t)(rel | 0x4000u);
                    }
                }
            }
            __syncthreads();
            KD_MARK(c_cls)
            const uint32_t ni = s_cnt[par][0], np = s_cnt[par][1], ncx = s_cnt[par][2], nx = s_cnt[par][3];
            // DEEP tile (a sorted batch: the tile's first and last candidates bracket its start sites): KD_DEEP_READS_PER_SITE or more
            // reads per start site -- the plain reads are walked from different chunks
            const kd_u64 tile_n = (tb + KD_TILE < last ? tb + KD_TILE : last) - tb;
            const bool deep = !ROWS && !order && rd.osh == 0 && (kd_u64)(s_glast[par] - s_gfirst[par] + 1u) * KD_DEEP_READS_PER_SITE <= tile_n;
            if (t == 0) { s_cnt[par ^ 1u][0] = 0; s_cnt[par ^ 1u][1] = 0; s_cnt[par ^ 1u][2] = 0; s_cnt[par ^ 1u][3] = 0; s_row[par ^ 1u] = 0; }   // next tile's counters (idle until its classify)
#pragma unroll
            for (uint32_t u = 0; u < KD_TILE_PER_THREAD; u++) {
                const kd_u64 j = tb + KD_TILE + u * KD_BLOCK + t;
                p_sc[u] = KD_CLS_SKIP;
                if (j < last && !(KD_EXP_DELETE & 64)) {
                    const KdRInfo ri = KD_RI(rinfo, rd, order ? (kd_u64)order[j] : j);
                    p_gs[u] = ri.gstart; p_sc[u] = ri.span_cls; p_ld[u] = (ri.lead & 0xffffffu) | (ri.pad & 0xff000000u);
                }
            }
            // homogeneous wavefronts: first the plain reads, then the complex ones.  Lane l of a wavefront takes
            // list entries l*rows + r: neighbours in a wavefront are `rows` reads apart in the sorted batch,
            // which keeps them off the same LDS counters in the same instruction.
            const uint32_t rows_i = (ni + KD_WAVE - 1) / KD_WAVE, rows_p = (np + KD_WAVE - 1) / KD_WAVE, rows_c = (ncx + KD_WAVE - 1) / KD_WAVE;
            // ROWS HANDED OUT: a wavefront takes the next row of the tile from a counter -- the complex rows first (the longest), then
            // the plain rows inside the histogram, those cut by its end, the early / late ones -- instead of every fourth row of
            // every list: a wavefront whose rows waited longer for their bases takes fewer of them.
            const uint32_t rows_x = ROWS ? 0u : (nx + KD_WAVE - 1) / KD_WAVE;
            const uint32_t rows_all = (ROWS ? 0u : rows_c + rows_p + rows_x) + rows_i;
            for (;;) {
                uint32_t rr = 0;
                if (lane == 0) rr = atomicAdd(&s_row[par], 1u);
                rr = kd_readfirstlane(rr);
                if (rr >= rows_all) break;
                if (!ROWS && rr < rows_c) {
                    const uint32_t r = rr, e = lane * rows_c + r;
                    if (e < ncx) {
                        const kd_u64 j = tb + l_cplx[e], i = order ? (kd_u64)order[j] : j;
                        KdRInfo ri;
                        const uint32_t cm = (KD_LIST_CARRY && e < KD_CMETA) ? l_cmeta[e] : 0xffffffffu;
                        if (cm == 0xffffffffu) ri = KD_RI(rinfo, rd, i);
                        else { ri.gstart = (uint32_t)wlo + (cm & 1023u); ri.span_cls = ((cm >> 10) & 1023u) << KD_SPAN_SHIFT; ri.lead = (cm >> 20) & 127u; ri.pad = (cm >> 27) << 24; }
                        const int32_t grel = (int32_t)(ri.gstart - (uint32_t)wlo);
                        const int32_t foot_end = grel + (int32_t)(ri.span_cls >> KD_SPAN_SHIFT);
#if KD_EXP_DELETE & 4      // (TIMING EXPERIMENT: no complex walk)
                        (void)grel; (void)foot_end;
#else
                        if (!kd_walk_short(rd, i, ri, wlo, We, Wh, hist0)) {   // more than three segments: general walk
                            kd_walk_ops(rd, i, ri.pad >> 24, 0u, ri.pad >> 24, grel, 0, (int32_t)ri.lead, foot_end, We, Wh, hist0);
                        }
#endif
                    }
                    continue;
                }
                if (!ROWS) rr -= rows_c;
                if (rr < rows_i) {
#ifdef KD_EXP_ADJ      // (EXPERIMENT: a wavefront's lanes take ADJACENT list entries -- neighbouring reads, neighbouring memory -- instead of entries
                    //  `rows` apart; 1: with the chunk rotation of deep tiles, 2: without; 3 / 4: PAIRS / QUADS of lanes on adjacent entries, each
                    //  lane of a group from another chunk, the groups `rows` apart)
                    const uint32_t r = rr;
                    const uint32_t e = KD_EXP_ADJ == 3 ? 2u * ((lane >> 1) * rows_i + r) + (lane & 1u)
                                     : KD_EXP_ADJ == 4 ? 4u * ((lane >> 2) * rows_i + r) + (lane & 3u) : r * KD_WAVE + lane;
#else
                    const uint32_t r = rr, e = lane * rows_i + r;
#endif
                    if (e < ni) {
                        const uint32_t le = l_plain[e];
                        const kd_u64 j = tb + (le & 1023u), i = order ? (kd_u64)order[j] : j;
                        KdRInfo ri;
                        if (ROWS || !KD_LIST_CARRY || (le >> 20) == 0u) ri = KD_RI(rinfo, rd, i);
                        else { ri.gstart = (uint32_t)wlo + ((le >> 10) & 1023u); ri.span_cls = (le >> 20) << KD_SPAN_SHIFT; ri.lead = 0; ri.pad = 0; }
                        if (ROWS) kd_walk_row(rd, i, ri, wlo, Wi, lane + 17u * r, hist0);
#ifdef KD_EXP_ADJ
                        else if (KD_EXP_ADJ == 1) kd_walk_inner<true>(rd, i, ri, wlo, We, Wh, hist0, lane);
                        else if (KD_EXP_ADJ == 3) kd_walk_inner<true>(rd, i, ri, wlo, We, Wh, hist0, (lane & 1u) * 2u);
                        else if (KD_EXP_ADJ == 4) kd_walk_inner<true>(rd, i, ri, wlo, We, Wh, hist0, lane & 3u);
#endif
                        else if (deep) kd_walk_inner<true>(rd, i, ri, wlo, We, Wh, hist0, lane);
                        else kd_walk_inner<false>(rd, i, ri, wlo, We, Wh, hist0, 0u);
                    }
                    continue;
                }
                rr -= rows_i;
                if (rr < rows_p) {
                    const uint32_t r = rr, e = lane * rows_p + r;
                    if (e < np) {
                        const kd_u64 j = tb + l_plain[KD_TILE - 1u - e], i = order ? (kd_u64)order[j] : j;
                        kd_walk_plain(rd, i, KD_RI(rinfo, rd, i), wlo, We, Wh, hist0);
                    }
                    continue;
                }
                rr -= rows_p;
                {   // early / late entries (rare: an entry longer than the histogram's reach, a leading clip across the window's left
                    // edge): the general walk, early ones against the origin wlo + H over W sites, late ones against [wlo, whi)
                    const uint32_t e = rr * KD_WAVE + lane;
                    if (e < nx) {
                        const uint32_t code = l_cplx[KD_TILE - 1u - e];
                        const bool early = (code & 0x8000u) != 0;
                        const kd_u64 j = tb + (code & 0x3fffu), i = order ? (kd_u64)order[j] : j;
                        const KdRInfo ri = KD_RI(rinfo, rd, i);
                        const bool shifted = early && w != 0;          // early entry of a window that has windows in front
                        const kd_u64 org = shifted ? wlo + H : wlo;
                        uint32_t *h0 = shifted ? hist_early : hist0;
                        const int32_t Wx = (early && !shifted) ? We : Wi;
                        const int32_t grel = (int32_t)(ri.gstart - (uint32_t)org);
                        const int32_t foot_end = grel + (int32_t)(ri.span_cls >> KD_SPAN_SHIFT);
                        kd_walk_ops(rd, i, ri.pad >> 24, 0u, ri.pad >> 24, grel, 0, (int32_t)ri.lead, foot_end, Wx, Wh, h0);
                    }
                }
            }
            KD_MARK(c_cplx)
            __syncthreads();
            KD_MARK(c_wait)
        }
        // flush: lane = site pair.  A pair's 19 counters are consecutive in LDS: the thread requests them all at once (19 reads in
        // flight instead of one per loop turn), then walks the channels with the channel -> table-row mapping a compile-time
        // matter of the unrolled loop -- for a fixed channel consecutive lanes hit consecutive 8-byte words of one HBM row; zeros
        // are skipped.  (Rounds 1 - 3 looped channel by channel with the pair inside: 38 dependent LDS reads per thread, 9 % of
        // the wavefront clocks.)
        // LDS channel -> table channel (KD_CH_*): weights 0-4, deletions 5, csw 6-10, cew 11-15; 0xff = bad slot
        bool bad = false;
        for (uint32_t xw = t; xw < ((KD_EXP_DELETE & 8) ? 0u : (uint32_t)Wh); xw += KD_BLOCK) {     // (& 8: TIMING EXPERIMENT, no flush)
            uint32_t v[KD_HCH];
#pragma unroll
            for (uint32_t ch = 0; ch < KD_HCH; ch++) v[ch] = hist[xw * KD_HPITCH + ch];
            // the words hold window-relative sites s (low half) and s + 1 (high half); halo sites are dropped
            const int32_t sw = 2 * (int32_t)xw - KD_HALO;
            const kd_u64 g0 = wlo + (kd_u64)sw;   // even: W, the halo and the G-space rows are all even / 8-byte aligned
            const bool whole = sw >= 0 && sw + 1 < We && g0 + 1 < T.sites && kd_commit(T, g0) && kd_commit(T, g0 + 1);
#pragma unroll
            for (uint32_t ch = 0; ch < KD_HCH; ch++) {
                uint32_t tch, tch2 = 0xffu;
                if (ROWS) {   // row symbols (kd_common.h): 0 nothing, 1-5 A,T,G,C,N, 6 deleted; 7-13 the same with an insertion in front
                    if (ch == KD_ROW_SKIP || ch > KD_ROW_DEL + KD_ROW_INS) continue;
                    const uint32_t sym = ch >= KD_ROW_INS ? ch - KD_ROW_INS : ch;
                    tch = sym == KD_ROW_SKIP ? (uint32_t)KDC_INS_TOTAL : sym == KD_ROW_DEL ? (uint32_t)KDC_DEL : sym - 1u;
                    if (ch >= KD_ROW_INS && sym != KD_ROW_SKIP) tch2 = KDC_INS_TOTAL;
                } else {
                    tch = ch < 5 ? ch : ch == KD_HCH_DEL ? (uint32_t)KDC_DEL
                        : (ch >= 7 && ch < 12) ? ch - 1 : (ch >= 13 && ch < 18) ? ch - 2 : 0xffu;
                }
                const uint32_t x = v[ch];
                if (!x) continue;
                uint32_t *row = T.tab + (kd_u64)(tch == 0xffu ? 0u : tch) * T.stride;
                uint32_t *row2 = T.tab + (kd_u64)(tch2 == 0xffu ? 0u : tch2) * T.stride;
                if (tch != 0xffu && whole) {
                    // both sites of the word live: ONE 64-bit add on the two adjacent u32 counters (the low counter
                    // cannot carry into the high one: a u32 table counter never wraps)
                    atomicAdd(reinterpret_cast<kd_u64 *>(row + g0), (kd_u64)(x & 0xffffu) | ((kd_u64)(x >> 16) << 32));
                    if (ROWS && tch2 != 0xffu)
                        atomicAdd(reinterpret_cast<kd_u64 *>(row2 + g0), (kd_u64)(x & 0xffffu) | ((kd_u64)(x >> 16) << 32));
                    continue;
                }
                for (int hlf = 0; hlf < 2; hlf++) {
                    const uint32_t cnt = hlf ? x >> 16 : x & 0xffffu;
                    const int32_t sw2 = sw + hlf;
                    if (!cnt || sw2 < 0 || sw2 >= We) continue;
                    const kd_u64 g = wlo + (kd_u64)sw2;
                    if (tch == 0xffu) bad = true;
                    else if (g < T.sites && kd_commit(T, g)) {
                        atomicAdd(&row[g], cnt);
                        if (ROWS && tch2 != 0xffu) atomicAdd(&row2[g], cnt);
                    }
                }
            }
        }
        // a base outside A,C,G,T,N inside an aligned or clipped segment: k_errors (kd_find_bad_base) pins down the read
        if (bad) atomicAdd(&status[KDS_BAD_BASE], 1ULL);
        KD_MARK(c_flush)
        __syncthreads();
        KD_MARK(c_wait)
    }
    KD_PHASE_COMMIT(status, ROWS)
}
