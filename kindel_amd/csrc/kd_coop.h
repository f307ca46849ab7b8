// kd_coop.h -- k_window_coop: the LDS-histogram pileup with a COOPERATIVE walk (16 lanes per read), the default first pass.
// Part of the device code of kd_kernels.h (included from there, in order; not a stand-alone header).
//
// Why: profiles/valu_issue_calibration.json (scripts/issue_calib.hip, MI355X).  A wave64 ds_add_u32 occupies the CU's LDS
// pipeline for max(4, 2 x bank-conflict ways) cycles.  k_window's walk -- one lane per read, 64 lanes on 64 unrelated
// (site pair, channel) dwords -- is a ~4-way pattern: 3.7 ns per wave-instruction, 6.4 x 10^7 of them per launch on the
// bench workload = 0.92 of the kernel's 1.17 ms (SQ_ACTIVE_INST_LDS agrees: 81 % of the kernel).  Here consecutive lanes
// take consecutive BYTES of one read (two bases each), so the 16 lanes of a read aim at 16 consecutive site pairs, and
// the histogram rows are channel-major with a pitch that is a multiple of the 32 banks: bank = pair mod 32 whatever the
// base.  Four reads share a wavefront; two 16-runs per 32-lane LDS pass overlap at most 2-way, which costs nothing
// (measured: 1.93 ns per wave-instruction, the conflict-free price is 1.75).  Lanes are live 94 % of the time on 150-base
// reads (63 % in the lane-per-read walk), and the packed bases are fetched as coalesced byte runs, once per window they
// touch.
//
// LDS (dynamic): u16 counters, two sites per dword (low half = even site), ROW major, pitch P = W / 2 + KD_CPAD dwords:
//   rows  0..15  weights              row = the BAM nibble itself (A 1, C 2, G 4, T 8, N 15); a count in any other row is a
//                                     base outside A,C,G,T,N = KeyError in the reference (kindel.py:51-52), seen at flush
//   rows 16..23  clip_start_weights   row = nibble & 7 (A 1, C 2, G 4, T 0, N 7); the walker checks the nibble itself
//   rows 24..31  clip_end_weights     (clipped bases are ~1 % of the bases: their check costs nothing on the hot path)
//   row  32      deletions
// KD_CPAD dwords in front of row 0 and at the end of every row: the pair in front of / behind the window exists, so a
// byte that straddles a window edge adds its outside base with value 0 into padding and needs no branch.
// Work decomposition, planning, tiles and the flush are k_window's (kd_window.h); long-read SEGMENTS keep k_window.
#pragma once
#include "kd_common.h"
#include "kd_window.h"

#define KD_CG 16                         // lanes per read
#define KD_CGROUPS (KD_WAVE / KD_CG)     // reads per wavefront
#define KD_CROW_W 0u
#define KD_CROW_CSW 16u
#define KD_CROW_CEW 24u
#define KD_CROW_DEL 32u
#define KD_CNROWS 33u
#define KD_CPAD 32u
#define KD_CPRE 5                        // chunks (16 bytes = 32 bases per read) whose bytes are requested ahead: a 150-base read
#define KD_COOP_PITCH(W) ((W) / 2 + KD_CPAD)
#define KD_CT KD_BLOCK                    // reads per tile: one per thread of the workgroup
// histogram + the staged tile: per read a 16-byte record, its first four CIGAR words, KD_CPRE chunks of 16 packed-base bytes; two lists
#define KD_COOP_LDS_BYTES(P) ((size_t)(KD_CPAD + KD_CNROWS * (size_t)(P)) * 4 + (size_t)KD_CT * (16 + 16 + 16 * KD_CPRE + 4))
#define KD_NIB_VALID 0x8116u             // bit n set: nibble n is one of A, C, G, T, N

// One run of query bases [xa, xb) of a read that lands on consecutive sites of one row group: base x on window-relative
// site sx + x.  Lane k of the group owns byte 16 c + k of the read in chunk c (bases 2 (16 c + k) and + 1).
struct KdCoopRun {
    int32_t xa, xb;        // live query bases
    int32_t ba, bb;        // live bytes [ba, bb)
    int32_t c0, cb;        // first / last chunk with a live byte
    int32_t hA, hB;        // byte offsets into the histogram of the counters of this lane's two bases in chunk 0, row 0 of the group
    uint32_t vA, vB;       // add values (which u16 half)
};
__device__ __forceinline__ KdCoopRun kd_coop_run(uint32_t k, int32_t xa, int32_t xb, int32_t sx, uint32_t row, int32_t PB) {
    KdCoopRun r;
    r.xa = xa; r.xb = xb;
    r.ba = xa >> 1; r.bb = (xb + 1) >> 1;
    r.c0 = r.ba >> 4; r.cb = xb > xa ? (r.bb - 1) >> 4 : r.c0 - 1;
    const int32_t rowoff = (int32_t)(KD_CPAD * 4u) + (int32_t)row * PB;
    r.hA = rowoff + ((sx >> 1) + (int32_t)k) * 4;          // site sx + 2 (16 c + k)     -> pair (sx >> 1) + 16 c + k
    r.hB = rowoff + (((sx + 1) >> 1) + (int32_t)k) * 4;    // site sx + 2 (16 c + k) + 1
    const uint32_t p = (uint32_t)sx & 1u;
    r.vA = p ? 0x10000u : 1u; r.vB = p ? 1u : 0x10000u;
    return r;
}
// byte b of chunk c: both bases, dead ones (outside [xa, xb): a run end or a window edge inside the byte) add 0
template <bool CLIP>
__device__ __forceinline__ void kd_coop_add2(unsigned char *hb, const KdCoopRun &r, uint32_t b, int32_t c, uint32_t k, int32_t PB, bool &bad) {
    const int32_t x0 = 2 * (16 * c + (int32_t)k);
    const uint32_t a = (x0 >= r.xa && x0 < r.xb) ? r.vA : 0u, bv = (x0 + 1 >= r.xa && x0 + 1 < r.xb) ? r.vB : 0u;
    uint32_t hi = b >> 4, lo = b & 15u;
    if (CLIP) {   // the clip rows are indexed by nibble & 7: the nibble itself is checked here
        if ((a && !((KD_NIB_VALID >> hi) & 1u)) || (bv && !((KD_NIB_VALID >> lo) & 1u))) bad = true;
        hi &= 7u; lo &= 7u;
    }
    atomicAdd(reinterpret_cast<uint32_t *>(hb + r.hA + (int32_t)hi * PB + 64 * c), a);
    atomicAdd(reinterpret_cast<uint32_t *>(hb + r.hB + (int32_t)lo * PB + 64 * c), bv);
}

// What the walk knows about a read: window-relative start (may be negative), span | flags, leading-clip reach, query length |
// CIGAR words << 24 (KdRInfo::pad).  Staged in LDS per tile.
struct alignas(16) KdCoopRec { int32_t grel; uint32_t span_cls, lead, pad; };

// A read as the cooperative walk sees it.  PRE: this lane's byte of each of its (at most KD_CPRE) chunks and its (at most
// four) CIGAR words are in registers (read from the staged tile); otherwise seq / cg point at global memory.
struct KdCoopRead {
    int32_t grel;
    uint32_t span_cls, lead, n_cig;
    const uint8_t *seq;
    const uint32_t *cg;
    uint32_t by[KD_CPRE];
    uint32_t cw[4];
};
// the run, chunk by chunk.  PRE: every chunk of the read is in registers, no memory access here.  Otherwise a loop of loads.
template <bool CLIP, bool PRE>
__device__ __forceinline__ void kd_coop_walk_run(unsigned char *hb, const KdCoopRead &R, const KdCoopRun &r, uint32_t k, int32_t PB, bool &bad) {
    if (PRE) {
#pragma unroll
        for (int u = 0; u < KD_CPRE; u++) {
            const int32_t bi = 16 * u + (int32_t)k;
            if (u >= r.c0 && u <= r.cb && bi >= r.ba && bi < r.bb) kd_coop_add2<CLIP>(hb, r, R.by[u], u, k, PB, bad);
        }
    } else {
        int32_t c = r.c0;
        int32_t bi = 16 * c + (int32_t)k;
        uint32_t cur = (c <= r.cb && bi >= r.ba && bi < r.bb) ? R.seq[bi] : 0u;
        for (; c <= r.cb; c++) {   // the next chunk's byte in flight
            const int32_t bn = 16 * (c + 1) + (int32_t)k;
            uint32_t nxt = 0;
            if (c + 1 <= r.cb && bn >= r.ba && bn < r.bb) nxt = R.seq[bn];
            bi = 16 * c + (int32_t)k;
            if (bi >= r.ba && bi < r.bb) kd_coop_add2<CLIP>(hb, r, cur, c, k, PB, bad);
            cur = nxt;
        }
    }
}

// A regular read op by op (at most KD_PREP_MAX_OPS ops): the 16 lanes walk the CIGAR together (every lane decodes the same
// words), every M/=/X run and every soft clip is one kd_coop_walk_run, a deletion is one add per lane.
// Semantics as kd_walk_ops (kindel.py:49-81 for a REGULAR read: no wrap-around, nothing beyond the contig).
// PRE: bases and CIGAR words come from R's registers (<= 4 words, <= 32 KD_CPRE bases); otherwise they are loaded on the way.
template <bool PRE>
__device__ __forceinline__ void kd_coop_ops(unsigned char *hb, const KdCoopRead &R, int32_t Wi, int32_t PB, uint32_t k, bool &bad) {
    const uint32_t nc = R.n_cig;
    int32_t grel = R.grel, q = 0;
    const int32_t foot_end = grel + (int32_t)(R.span_cls >> KD_SPAN_SHIFT), lead = (int32_t)R.lead;
    for (uint32_t kk = 0; kk < nc; kk++) {
        uint32_t cw;
        if (PRE) cw = kk == 0 ? R.cw[0] : kk == 1 ? R.cw[1] : kk == 2 ? R.cw[2] : R.cw[3];
        else cw = R.cg[kk];
        const int32_t len = (int32_t)(cw >> 4);
        const uint32_t op = cw & 15u;
        if (op == 0 || op == 7 || op == 8) {
            const int32_t xa = grel < 0 ? q - grel : q, xb = Wi - grel < len ? q + (Wi - grel) : q + len;
            kd_coop_walk_run<false, PRE>(hb, R, kd_coop_run(k, xa, xb, grel - q, KD_CROW_W, PB), k, PB, bad);
            q += len; grel += len;
            if (grel >= Wi) break;
        } else if (op == 2) {
            const int32_t j0 = grel < 0 ? -grel : 0, j1 = Wi - grel < len ? Wi - grel : len;
            for (int32_t j = j0 + (int32_t)k; j < j1; j += KD_CG) {
                const int32_t s = grel + j;
                atomicAdd(reinterpret_cast<uint32_t *>(hb + (int32_t)(KD_CPAD * 4u) + (int32_t)KD_CROW_DEL * PB + (s >> 1) * 4), 1u << (16 * (s & 1)));
            }
            grel += len;
            if (grel >= Wi) break;
        } else if (op == 1) {
            q += len;
        } else if (op == 4) {
            if (kk == 0) {   // leading clip, kindel.py:64-73: base j -> site r - len + j, the last `lead` bases are kept
                const int32_t s_first = grel - len;
                const int32_t xa = -s_first > len - lead ? -s_first : len - lead, xb = Wi - s_first < len ? Wi - s_first : len;
                kd_coop_walk_run<true, PRE>(hb, R, kd_coop_run(k, xa, xb, s_first, KD_CROW_CEW, PB), k, PB, bad);
                q += len;
            } else {         // non-first clip, kindel.py:74-81: the last op of a regular read that moves r
                const int32_t n_adv = foot_end - grel;
                const int32_t xa = grel < 0 ? q - grel : q, xb = Wi - grel < n_adv ? q + (Wi - grel) : q + n_adv;
                kd_coop_walk_run<true, PRE>(hb, R, kd_coop_run(k, xa, xb, grel - q, KD_CROW_CSW, PB), k, PB, bad);
                break;
            }
        }
    }
}

// The staged tile in LDS (all threads of the workgroup write their own read's slots, the walk reads them 16 lanes per read)
struct KdCoopStage {
    const KdCoopRec *rec;      // [KD_CT]
    const uint4 *cig;          // [KD_CT] the first four CIGAR words
    const unsigned char *b8;   // [KD_CPRE][KD_CT][16] packed bases: chunk u of read t at (u * KD_CT + t) * 16
};

// PLAIN short reads (one M/=/X run over the whole read, at most 32 KD_CPRE bases).  Group g of a wavefront takes list entries
// g * rows + r: the four reads of a wavefront lie a quarter of the list apart (different sites: their 16-pair runs rarely
// meet in one LDS pass).  Everything the walk touches is in LDS.  Two paths per row, chosen for the whole wavefront:
//   FAST  every run of the row begins and ends on a byte boundary (or is cut by the window edge, where the half byte outside
//         lands in the row's padding and is never flushed): per chunk one range compare, two shifts / masks, two multiply-adds,
//         two ds_add_u32 with the chunk in the immediate offset;
//   EXACT a run that ends inside a byte (odd read length): per-base compares (kd_coop_add2).
__device__ __forceinline__ void kd_coop_list_plain(unsigned char *hb, const KdCoopStage &S, const uint16_t *list, uint32_t n, uint32_t r_first,
                                                   int32_t Wi, int32_t PB, uint32_t k, uint32_t grp, bool &bad) {
    if (!n) return;
    const uint32_t rows = (n + KD_CGROUPS - 1) / KD_CGROUPS;
    for (uint32_t r = r_first; r < rows; r += KD_WAVES_PER_BLOCK) {
        const uint32_t e = grp * rows + r;
        const bool valid = e < n;
        const uint32_t idx = list[valid ? e : 0u];
        const KdCoopRec rc = S.rec[idx];
        uint32_t by[KD_CPRE];
#pragma unroll
        for (int u = 0; u < KD_CPRE; u++) by[u] = S.b8[((uint32_t)u * KD_CT + idx) * 16u + k];
        const int32_t grel = rc.grel, len = (int32_t)(rc.span_cls >> KD_SPAN_SHIFT);
        const int32_t xa = grel < 0 ? -grel : 0, xb = Wi - grel < len ? Wi - grel : len;
        // a run end inside a byte that is NOT a window cut: only the read's own end (xb == len, len odd)
        const bool exact = valid && (len & 1) && xb == len;
        if (kd_ballot(exact)) {
            if (valid) {
                const KdCoopRun run = kd_coop_run(k, xa, xb, grel, KD_CROW_W, PB);
#pragma unroll
                for (int u = 0; u < KD_CPRE; u++) {
                    const int32_t bi = 16 * u + (int32_t)k;
                    if (u >= run.c0 && u <= run.cb && bi >= run.ba && bi < run.bb) kd_coop_add2<false>(hb, run, by[u], u, k, PB, bad);
                }
            }
        } else {
            // whole bytes [ba, bb): the window cuts are rounded OUTWARDS (the extra base sits on site -1 or W: padding)
            const int32_t ba = xa >> 1, bb = (xb + 1) >> 1;
            const uint32_t nlive = valid ? (uint32_t)(bb - ba) : 0u;
            const uint32_t t0 = (uint32_t)((int32_t)k - ba);          // byte 16 u + k is live iff (uint32_t)(16 u + t0) < nlive
            const uint32_t p = (uint32_t)grel & 1u;
            const uint32_t vA = 1u << (16u * p), vB = 0x10000u >> (16u * p);
            unsigned char *hA = hb + (int32_t)(KD_CPAD * 4u) + ((grel >> 1) + (int32_t)k) * 4;
            unsigned char *hB = hA + 4 * (int32_t)p;
#pragma unroll
            for (int u = 0; u < KD_CPRE; u++) {
                if (16u * (uint32_t)u + t0 < nlive) {
                    const uint32_t b = by[u];
                    atomicAdd(reinterpret_cast<uint32_t *>(hA + (int32_t)(b >> 4) * PB + 64 * u), vA);
                    atomicAdd(reinterpret_cast<uint32_t *>(hB + (int32_t)(b & 15u) * PB + 64 * u), vB);
                }
            }
        }
    }
}
// COMPLEX short reads (clips / indels, at most four CIGAR words, at most 32 KD_CPRE bases): the same rows, op by op.
__device__ __forceinline__ void kd_coop_list_complex(unsigned char *hb, const KdCoopStage &S, const uint16_t *list, uint32_t n, uint32_t r_first,
                                                     int32_t Wi, int32_t PB, uint32_t k, uint32_t grp, bool &bad) {
    if (!n) return;
    const uint32_t rows = (n + KD_CGROUPS - 1) / KD_CGROUPS;
    for (uint32_t r = r_first; r < rows; r += KD_WAVES_PER_BLOCK) {
        const uint32_t e = grp * rows + r;
        if (e >= n) continue;
        const uint32_t idx = list[e];
        const KdCoopRec rc = S.rec[idx];
        const uint4 cw = S.cig[idx];
        KdCoopRead R;
        R.grel = rc.grel; R.span_cls = rc.span_cls; R.lead = rc.lead; R.n_cig = rc.pad >> 24; R.seq = nullptr; R.cg = nullptr;
#pragma unroll
        for (int u = 0; u < KD_CPRE; u++) R.by[u] = S.b8[((uint32_t)u * KD_CT + idx) * 16u + k];
        R.cw[0] = cw.x; R.cw[1] = cw.y; R.cw[2] = cw.z; R.cw[3] = cw.w;
        kd_coop_ops<true>(hb, R, Wi, PB, k, bad);
    }
}
// The reads the staged walks do not take (more than 32 KD_CPRE bases or more than four CIGAR words: long plain reads, short
// reads with many indels): bases and CIGAR words loaded from global memory where they are needed.
__device__ __forceinline__ void kd_coop_list_general(unsigned char *hb, const KdReads &rd, const KdCoopStage &S, const uint32_t *order,
                                                     const uint16_t *list, uint32_t n, uint32_t r_first, kd_u64 tb, int32_t Wi, int32_t PB,
                                                     uint32_t k, uint32_t grp, bool &bad) {
    if (!n) return;
    const uint32_t rows = (n + KD_CGROUPS - 1) / KD_CGROUPS;
    for (uint32_t r = r_first; r < rows; r += KD_WAVES_PER_BLOCK) {
        const uint32_t e = grp * rows + r;
        if (e >= n) continue;
        const uint32_t idx = list[-(int32_t)e];          // (this list grows downwards from the end of the plain list's array)
        const kd_u64 j = tb + idx;
        const kd_u64 i = order ? (kd_u64)order[j] : j;
        const KdCoopRec rc = S.rec[idx];
        KdCoopRead R;
        R.grel = rc.grel; R.span_cls = rc.span_cls; R.lead = rc.lead; R.n_cig = rc.pad >> 24;
        R.seq = KD_SEQ_AT(rd, i);
        R.cg = rd.cigar + KD_COFF(rd, i);
        kd_coop_ops<false>(hb, R, Wi, PB, k, bad);
    }
}

// histogram row -> table channel (KDC_*); 0xff: a weights row no valid nibble selects (a count there = bad base); 0xfe: unused
__device__ __forceinline__ uint32_t kd_coop_row_channel(uint32_t row) {
    if (row < 16u) return row == 1u ? KDC_A : row == 8u ? KDC_T : row == 4u ? KDC_G : row == 2u ? KDC_C : row == 15u ? KDC_N : 0xffu;
    if (row == KD_CROW_DEL) return KDC_DEL;
    const uint32_t g = row < KD_CROW_CEW ? KDC_CSW : KDC_CEW, n7 = row & 7u;
    return n7 == 1u ? g + 0u : n7 == 0u ? g + 1u : n7 == 4u ? g + 2u : n7 == 2u ? g + 3u : n7 == 7u ? g + 4u : 0xfeu;
}

#ifndef KD_COOP_OCC
#define KD_COOP_OCC 2
#endif
// Per work item (window, slice of candidate reads), tiles of KD_CT = 256 reads:
//   * the LOADS are one LANE per read, like k_window's: thread t asks for read t's footprint and offsets two tiles ahead and
//     for its packed bases (KD_CPRE unaligned 16-byte chunks) and first CIGAR words one tile ahead -- 256 reads per workgroup
//     in flight, their latency behind the walk of the tiles before (a walk that loads as it goes has four reads in flight
//     per wavefront: 3 us per row of four reads, measured);
//   * at the tile's turn every thread writes its read's record / CIGAR words / chunks into LDS and enters it in a list
//     (plain, complex, general);
//   * the WALK is cooperative and touches only LDS (kd_coop_list_*).
__global__ void __launch_bounds__(KD_BLOCK, KD_COOP_OCC)
k_window_coop(KdReads rd, const KdRInfo *rinfo, const uint32_t *order, KdTabs T, const kd_u64 *win_lo, const kd_u64 *win_hi,
              const kd_u64 *item_off, const uint32_t *item_win, kd_u64 items_cap, uint32_t w0, uint32_t W, uint32_t P_, uint32_t slice,
              kd_u64 *status) {
    KD_DYN_SHARED(uint32_t, hist);
    const int32_t P = (int32_t)P_, PB = 4 * P;       // row pitch in dwords / bytes
    const uint32_t nh = KD_CPAD + KD_CNROWS * (uint32_t)P;
    unsigned char *hb = reinterpret_cast<unsigned char *>(hist);
    KdCoopRec *s_rec = reinterpret_cast<KdCoopRec *>(hist + nh);
    uint4 *s_cig = reinterpret_cast<uint4 *>(s_rec + KD_CT);
    uint4 *s_b = s_cig + KD_CT;                                     // [KD_CPRE][KD_CT]
    // tile-relative read indices: the plain short reads fill l_a from the front, the reads of the general walk from the back
    uint16_t *l_a = reinterpret_cast<uint16_t *>(s_b + KD_CPRE * KD_CT);
    uint16_t *l_cplx = l_a + KD_CT;
    KdCoopStage S;
    S.rec = s_rec; S.cig = s_cig; S.b8 = reinterpret_cast<const unsigned char *>(s_b);
    __shared__ kd_u64 s_item;
    __shared__ uint32_t s_cnt[3];   // plain, complex, general list lengths of the tile being walked
    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & (KD_WAVE - 1), wave = t / KD_WAVE;
    const uint32_t k = lane & (KD_CG - 1), grp = lane / KD_CG;
    const kd_u64 total = status[KDS_TOTAL_ITEMS];
    const int32_t Wi = (int32_t)W;
    bool bad = false;
    for (;;) {
        if (t == 0) s_item = atomicAdd(&status[KDS_NEXT_ITEM], 1ULL);
        __syncthreads();
        const kd_u64 item = s_item;
        if (item >= total || item >= items_cap) break;
        const uint32_t w = item_win[item];
        const kd_u64 wlo = (kd_u64)(w0 + w) * W, whi = wlo + W;
        const kd_u64 first = win_lo[w] + (item - item_off[w]) * slice;
        const kd_u64 last = first + slice < win_hi[w] ? first + slice : win_hi[w];
        const int64_t n_tiles = (int64_t)((last - first + KD_CT - 1) / KD_CT);
        {   // nh is a multiple of 4: zero with 16-byte stores
            uint4 *h4 = reinterpret_cast<uint4 *>(hist);
            for (uint32_t x = t; x < nh / 4; x += KD_BLOCK) h4[x] = make_uint4(0u, 0u, 0u, 0u);
        }
        // stage registers of this thread: first level (tile it + 2), second level (tile it + 1)
        KdRInfo a_ri; a_ri.gstart = 0; a_ri.span_cls = KD_CLS_SKIP; a_ri.lead = 0; a_ri.pad = 0;
        kd_u64 a_so = 0, a_co = 0;
        KdRInfo b_ri = a_ri;
        KdChunk b_ch[KD_CPRE];
        uint32_t b_cw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int u = 0; u < KD_CPRE; u++) { b_ch[u].x = 0; b_ch[u].y = 0; b_ch[u].z = 0; b_ch[u].w = 0; }
        for (int64_t it = -2; it < n_tiles; it++) {
            if (it >= 0) {
                // this thread's read of tile `it` (loaded over the last two turns) -> LDS, and into one of the lists
                if (t < 3) s_cnt[t] = 0;
                __syncthreads();
                KdCoopRec rc;
                rc.grel = (int32_t)(b_ri.gstart - (uint32_t)wlo); rc.span_cls = b_ri.span_cls; rc.lead = b_ri.lead; rc.pad = b_ri.pad;
                s_rec[t] = rc;
                s_cig[t] = make_uint4(b_cw[0], b_cw[1], b_cw[2], b_cw[3]);
#pragma unroll
                for (int u = 0; u < KD_CPRE; u++) s_b[u * KD_CT + t] = make_uint4(b_ch[u].x, b_ch[u].y, b_ch[u].z, b_ch[u].w);
                const kd_u64 gs = b_ri.gstart, span = b_ri.span_cls >> KD_SPAN_SHIFT;
                if ((b_ri.span_cls & 3u) == KD_CLS_REG && gs + span > wlo && gs - b_ri.lead < whi) {
                    // the staged walks take reads whose bases fit KD_CPRE chunks and whose CIGAR fits four words
                    const bool small = (b_ri.pad & 0xffffffu) <= 32u * KD_CPRE && (b_ri.pad >> 24) <= 4u;
                    if (!small) l_a[KD_CT - 1u - atomicAdd(&s_cnt[2], 1u)] = (uint16_t)t;
                    else if (b_ri.span_cls & KD_INFO_PLAIN) l_a[atomicAdd(&s_cnt[0], 1u)] = (uint16_t)t;
                    else l_cplx[atomicAdd(&s_cnt[1], 1u)] = (uint16_t)t;
                }
            }
            if (it + 1 >= 0 && it + 1 < n_tiles) {
                // second level for tile it + 1: the bases and CIGAR words of the read whose footprint / offsets arrived (unconditional
                // loads with clamped offsets: a chunk or word behind the read's end repeats its last one and is never used)
                b_ri = a_ri;
                const uint32_t nby = ((a_ri.pad & 0xffffffu) + 1u) >> 1, nc = a_ri.pad >> 24;
                const uint32_t last_chunk = nby ? ((nby - 1u) >> 4) : 0u;
                const uint8_t *sp = rd.seq4 + a_so;
#pragma unroll
                for (int u = 0; u < KD_CPRE; u++)
                    b_ch[u] = *reinterpret_cast<const KdChunk *>(sp + 16u * ((uint32_t)u < last_chunk ? (uint32_t)u : last_chunk));
                const uint32_t *cg = rd.cigar + (nc ? a_co : 0);
                const uint32_t lw = nc ? nc - 1u : 0u;
                b_cw[0] = cg[0]; b_cw[1] = cg[1u < lw ? 1u : lw]; b_cw[2] = cg[2u < lw ? 2u : lw]; b_cw[3] = cg[3u < lw ? 3u : lw];
            }
            if (it + 2 < n_tiles) {
                // first level for tile it + 2
                const kd_u64 j = first + (kd_u64)(it + 2) * KD_CT + t;
                a_ri.span_cls = KD_CLS_SKIP; a_ri.pad = 0; a_so = 0; a_co = 0;
                if (j < last) {
                    const kd_u64 i = order ? (kd_u64)order[j] : j;
                    a_ri = KD_RI(rinfo, rd, i);
                    a_so = KD_SOFF(rd, i); a_co = KD_COFF(rd, i);
                    if ((a_ri.span_cls & 3u) != KD_CLS_REG) { a_ri.pad = 0; a_so = 0; a_co = 0; }   // (pad means something else for long reads)
                }
            }
            if (it >= 0) {
                __syncthreads();
                const kd_u64 tb = first + (kd_u64)it * KD_CT;
                const uint32_t np = s_cnt[0], ncx = s_cnt[1], ng = s_cnt[2];
                // plain reads, then the complex ones, then the rest (each list's rows start at the wavefront after the one that took
                // the previous list's last row)
                const uint32_t rows_p = (np + KD_CGROUPS - 1) / KD_CGROUPS, rows_c = (ncx + KD_CGROUPS - 1) / KD_CGROUPS;
                const uint32_t w_c = (wave + KD_WAVES_PER_BLOCK - rows_p % KD_WAVES_PER_BLOCK) % KD_WAVES_PER_BLOCK;
                const uint32_t w_g = (wave + 2 * KD_WAVES_PER_BLOCK - (rows_p + rows_c) % KD_WAVES_PER_BLOCK) % KD_WAVES_PER_BLOCK;
                kd_coop_list_plain(hb, S, l_a, np, wave, Wi, PB, k, grp, bad);
                kd_coop_list_complex(hb, S, l_cplx, ncx, w_c, Wi, PB, k, grp, bad);
                kd_coop_list_general(hb, rd, S, order, l_a + KD_CT - 1, ng, w_g, tb, Wi, PB, k, grp, bad);
                __syncthreads();
            }
        }
        // flush: row by row, lane = site pair: consecutive lanes -> consecutive dwords of the LDS row (conflict free) and of
        // the HBM channel row (coalesced); zeros are skipped.  Rows no valid base selects are only checked.
        for (uint32_t row = 0; row < KD_CNROWS; row++) {
            const uint32_t tch = kd_coop_row_channel(row);
            if (tch == 0xfeu) continue;
            uint32_t *trow = T.tab + (kd_u64)(tch == 0xffu ? 0u : tch) * T.stride;
            const uint32_t *hrow = hist + KD_CPAD + row * (uint32_t)P;
            for (uint32_t xw = t; xw < W / 2; xw += KD_BLOCK) {
                const uint32_t v = hrow[xw];
                if (!v) continue;
                if (tch == 0xffu) { bad = true; continue; }
                const kd_u64 g0 = wlo + 2 * (kd_u64)xw;   // even: 8-byte aligned in the channel row
                if (g0 + 1 < T.sites && kd_commit(T, g0) && kd_commit(T, g0 + 1)) {
                    atomicAdd(reinterpret_cast<kd_u64 *>(trow + g0), (kd_u64)(v & 0xffffu) | ((kd_u64)(v >> 16) << 32));
                    continue;
                }
                if ((v & 0xffffu) && g0 < T.sites && kd_commit(T, g0)) atomicAdd(&trow[g0], v & 0xffffu);
                if ((v >> 16) && g0 + 1 < T.sites && kd_commit(T, g0 + 1)) atomicAdd(&trow[g0 + 1], v >> 16);
            }
        }
        __syncthreads();
    }
    // a base outside A,C,G,T,N inside an aligned or clipped segment: k_errors (kd_find_bad_base) pins down the read
    if (bad) atomicAdd(&status[KDS_BAD_BASE], 1ULL);
}
