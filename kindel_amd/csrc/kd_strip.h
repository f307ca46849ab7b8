// kd_strip.h -- k_strip: the SITE-MAJOR pileup of short regular reads (first pass of the record loop,
// /root/reference/kindel/kindel.py:40-81).  Part of the device code of kd_kernels.h.
#pragma once
#include "kd_common.h"

// k_window (kd_window.h) scatters: one lane per read, one LDS atomic per base.  k_strip gathers: a WAVEFRONT owns a
// STRIP of 64 consecutive G-space sites, lane l owns site s0 + l and keeps that site's tallies in REGISTERS; no LDS
// atomics, no workgroup barriers.  Per batch of up to 64 reads that overlap the strip:
//   stage       lane j converts read j's packed bases into one CODE BYTE per base (code = bit offset of the base's
//               counter field) and writes them into row j of a wavefront-private LDS tile, REALIGNED so that row byte
//               KD_ROW_BIAS + l is the read's base on site s0 + l; bytes the read does not cover hold the TRASH code.
//               8 bases (one source dword) -> two v_perm table look-ups per nibble half + one select + interleave,
//               v_alignbyte to the row's byte phase, a byte-compare mask against the covered range.
//   accumulate  for every row: ONE ds_read_u8 (the 64 lanes read 64 consecutive bytes: conflict-free, immediate row
//               offset, the same address register for every row) and ONE v_lshl_add_u32: acc += 1 << code.  `acc` packs
//               five 6-bit counters (A,T,G,C,N at bits 0,6,..,24); the trash code 30 lands in bits 30-31, which
//               overflow harmlessly.  Every <= 63 rows the fields are spilled into five 32-bit registers.
// Reads with clips / indels (<= KD_PREP_MAX_OPS ops) take the same route: their CIGAR is walked once per (read, strip),
// every M/=/X run is put into the row (read-modify-write at the run's edge dwords), leading / trailing soft clips become
// rows of their own in two further sub-passes (clip_end_weights / clip_start_weights accumulators), deleted sites are
// counted in a 64-entry LDS array (rare).  One flush per work item: 16 coalesced 32-bit atomicAdds of the non-zero
// counters.  Work items = (strip, slice of its candidate reads) from k_plan_* with a window of 64 sites; wavefronts pull
// pairs of consecutive items from eight queues (one per XCD: neighbouring strips -- which share most of their reads --
// are staged out of the same L2).
#define KD_STRIP 64u
#define KD_ROW_DW 21u          // dwords per staged row: odd -> lane-per-row dword stores are bank-conflict free
#define KD_ROW_BYTES (4u * KD_ROW_DW)
#define KD_ROW_BIAS 8          // row byte of strip-relative site 0 (dwords 0-1 and 18-20 are slack for whole-dword stores)
#define KD_TRASH4 0x1e1e1e1eu  // code 30 x 4
#define KD_BAD4 0x80808080u    // a code byte with bit 7 set: base outside A,C,G,T,N (KeyError in the reference)
#define KD_LIST 128u           // ring of pending candidates per class (< 64 pending + <= 64 new)
#ifndef KD_GRAB
#define KD_GRAB 2u             // consecutive work items per dequeue
#endif
#ifndef KD_STRIP_WGS
#define KD_STRIP_WGS 6         // workgroups per CU the kernel is built for (24 wavefronts: LDS 6 x 24 KB, <= 80 VGPRs)
#endif
#define KD_NQ 8u               // work queues (XCDs)

struct KdStripLds {
    uint32_t rows[KD_STRIP * KD_ROW_DW];
    uint16_t plain[KD_LIST], cplx[KD_LIST];   // item-relative candidate indices
    uint32_t dels[KD_STRIP];
};

struct __attribute__((packed, aligned(1))) KdWord { uint32_t v; };

// 8 packed bases (BAM nibbles, high nibble first) -> 8 code bytes: c0 = bases 0-3, c1 = bases 4-7.
// code = 6 * channel (A,T,G,C,N = 0..4, the reference's dict order, kindel.py:29); anything else = 0x9e (trash + bad).
// Two 8-entry byte tables (bit 3 of the nibble clear / set) looked up with v_perm_b32, a third v_perm selects per byte.
__device__ __forceinline__ void kd_conv8(uint32_t v, uint32_t &c0, uint32_t &c1) {
    const uint32_t TL_LO = 0x9e12009eu, TL_HI = 0x9e9e9e0cu;   // nibbles 0-7:  '=',A,C,M,G,R,S,V
    const uint32_t TH_LO = 0x9e9e9e06u, TH_HI = 0x189e9e9eu;   // nibbles 8-15: T,W,Y,H,K,D,B,N
    const uint32_t tl = v & 0x07070707u, th = (v >> 4) & 0x07070707u;
    const uint32_t sl = ((v >> 1) & 0x04040404u) | 0x03020100u, sh = ((v >> 5) & 0x04040404u) | 0x03020100u;
    const uint32_t rl = kd_perm(kd_perm(TH_HI, TH_LO, tl), kd_perm(TL_HI, TL_LO, tl), sl);   // bases 1,3,5,7
    const uint32_t rh = kd_perm(kd_perm(TH_HI, TH_LO, th), kd_perm(TL_HI, TL_LO, th), sh);   // bases 0,2,4,6
    c0 = kd_perm(rh, rl, 0x01050004u);
    c1 = kd_perm(rh, rl, 0x03070206u);
}

// value bytes at row positions inside [lo_b, hi_b), `other` bytes elsewhere.  pos = the four row positions of the dword,
// A = (0x80 - lo_b) * 0x01010101, B = (0x80 - hi_b) * 0x01010101: bit 7 of a byte of pos + A is set iff position >= lo_b.
__device__ __forceinline__ uint32_t kd_mask_bytes(uint32_t value, uint32_t other, uint32_t pos, uint32_t A, uint32_t B) {
    const uint32_t t = (pos + A) & ~(pos + B);
    return kd_perm(value, other, ((t >> 5) & 0x04040404u) | 0x03020100u);
}

// A PLAIN read (one M/=/X run = the whole read): query base x lies on strip-relative site p + x.  All 16 data dwords of
// the row are written exactly once (bytes outside the read = trash): nine source dwords, two row dwords each.
__device__ __forceinline__ void kd_stage_plain(uint32_t *row, const uint8_t *seq, int32_t p, int32_t len, uint32_t &bad) {
    const int32_t lo_b = KD_ROW_BIAS + (p > 0 ? p : 0);
    const int32_t hi_b = KD_ROW_BIAS + (p + len < (int32_t)KD_STRIP ? p + len : (int32_t)KD_STRIP);
    const int32_t delta = KD_ROW_BIAS + p;                 // row byte of query base 0
    const int32_t fd = delta >> 2, adj = (delta & 3) ? 1 : 0;
    const uint32_t e = (4u - ((uint32_t)delta & 3u)) & 3u;
    // row dword K = alignbyte(conv[M + 1], conv[M], e) with M = K - fd - adj; conv[2d], conv[2d + 1] = source dword d
    const int32_t ds = (2 - fd - adj) >> 1;                // first source dword (floor)
    int32_t K = 2 * ds - 1 + fd + adj;                     // 0 or 1: the first outputs fall into the slack dwords
    const int32_t dmax = (len - 1) >> 3;
    const uint32_t A = (uint32_t)(0x80 - lo_b) * 0x01010101u, B = (uint32_t)(0x80 - hi_b) * 0x01010101u;
    uint32_t pos = 0x03020100u + (uint32_t)K * 0x04040404u;
    uint32_t src[9];
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int32_t d = ds + t;
        src[t] = 0u;
        if (d >= 0 && d <= dmax) src[t] = reinterpret_cast<const KdWord *>(seq + 4 * (int64_t)d)->v;
    }
    uint32_t prev = 0u;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        uint32_t c0, c1;
        kd_conv8(src[t], c0, c1);
        const uint32_t o0 = kd_mask_bytes(kd_alignbyte(c0, prev, e), KD_TRASH4, pos, A, B);
        const uint32_t o1 = kd_mask_bytes(kd_alignbyte(c1, c0, e), KD_TRASH4, pos + 0x04040404u, A, B);
        prev = c1;
        bad |= o0 | o1;
        row[K] = o0; row[K + 1] = o1;
        K += 2; pos += 0x08080808u;
    }
}

// One M/=/X (or soft-clip) run of a read into a row that already holds trash / earlier runs: row bytes [b0, b0 + n) <-
// codes of query bases [q0, q0 + n).  KD_ROW_BIAS <= b0, b0 + n <= KD_ROW_BIAS + 64, n >= 1.  Interior dwords are
// stored whole, the two edge dwords are merged with what the row holds.
__device__ __forceinline__ void kd_put_run(uint32_t *row, const uint8_t *seq, int32_t q0, int32_t b0, int32_t n, int32_t dmax,
                                           uint32_t &bad) {
    const int32_t hi_b = b0 + n;
    const int32_t delta = b0 - q0;
    const int32_t fd = delta >> 2, adj = (delta & 3) ? 1 : 0;
    const uint32_t e = (4u - ((uint32_t)delta & 3u)) & 3u;
    const int32_t Kf = b0 >> 2, Kl = (hi_b - 1) >> 2;
    const int32_t ds = (Kf - fd - adj) >> 1, de = (Kl - fd - adj + 1) >> 1;
    int32_t K = 2 * ds - 1 + fd + adj;
    const uint32_t A = (uint32_t)(0x80 - b0) * 0x01010101u, B = (uint32_t)(0x80 - hi_b) * 0x01010101u;
    uint32_t pos = 0x03020100u + (uint32_t)K * 0x04040404u;
    uint32_t prev = 0u;
    for (int32_t d = ds; d <= de; d++) {
        uint32_t v = 0u, c0, c1;
        if (d >= 0 && d <= dmax) v = reinterpret_cast<const KdWord *>(seq + 4 * (int64_t)d)->v;
        kd_conv8(v, c0, c1);
        const uint32_t a0 = kd_alignbyte(c0, prev, e), a1 = kd_alignbyte(c1, c0, e);
        prev = c1;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int32_t Kh = K + h;
            if (Kh >= Kf && Kh <= Kl) {
                uint32_t o = h ? a1 : a0;
                if (Kh == Kf || Kh == Kl) o = kd_mask_bytes(o, row[Kh], pos + (h ? 0x04040404u : 0u), A, B);
                bad |= o;
                row[Kh] = o;
            }
        }
        K += 2; pos += 0x08080808u;
    }
}

// one group of counters: five 6-bit fields in `acc`, spilled into w[0..4] before a field can reach 64
struct KdAcc {
    uint32_t acc, since;
    uint32_t w[5];
};
__device__ __forceinline__ void kd_acc_init(KdAcc &a) { a.acc = 0; a.since = 0; for (int f = 0; f < 5; f++) a.w[f] = 0; }
__device__ __forceinline__ void kd_acc_spill(KdAcc &a) {
#pragma unroll
    for (int f = 0; f < 5; f++) a.w[f] += (a.acc >> (6 * f)) & 63u;
    a.acc = 0; a.since = 0;
}
// col = the lane's column of the tile: byte KD_ROW_BIAS + lane of row 0
__device__ __forceinline__ void kd_accumulate(const uint8_t *col, uint32_t nrows, KdAcc &a) {
    uint32_t r = 0;
    for (; r + 8 <= nrows; r += 8) {
        if (a.since + 8 > 63) kd_acc_spill(a);
        const uint8_t *c = col + r * KD_ROW_BYTES;
        uint32_t x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = c[k * KD_ROW_BYTES];
#pragma unroll
        for (int k = 0; k < 8; k++) a.acc += 1u << (x[k] & 31u);
        a.since += 8;
    }
    for (; r < nrows; r++) {
        if (a.since + 1 > 63) kd_acc_spill(a);
        a.acc += 1u << (col[r * KD_ROW_BYTES] & 31u);
        a.since += 1;
    }
}

__device__ __forceinline__ void kd_row_trash(uint32_t *row) {
#pragma unroll
    for (int k = 2; k < 18; k++) row[k] = KD_TRASH4;
}

// One work item: strip [s0, s0 + 64) against the candidates [first, last) of `rinfo` (through `order` if not NULL).
__device__ __forceinline__ void kd_strip_item(const KdReads &rd, const KdRInfo *rinfo, const uint32_t *order, const KdTabs &T,
                                              KdStripLds &L, uint32_t lane, kd_u64 s0, kd_u64 first, kd_u64 last, uint32_t &bad) {
    KdAcc aw, ae, as;   // weights, clip_end_weights, clip_start_weights
    kd_acc_init(aw); kd_acc_init(ae); kd_acc_init(as);
    L.dels[lane] = 0u;
    uint32_t n_pl = 0, n_cx = 0, d_pl = 0, d_cx = 0;   // appended to / consumed from the two rings (wave-uniform)
    const kd_u64 s1 = s0 + KD_STRIP;
    uint32_t *const row = &L.rows[lane * KD_ROW_DW];
    const uint8_t *const col = reinterpret_cast<const uint8_t *>(L.rows) + KD_ROW_BIAS + lane;

    // nr plain reads of the ring, from entry d_pl on: stage, accumulate
    auto plain_batch = [&](uint32_t nr) {
        if (lane < nr) {
            const kd_u64 j = first + L.plain[(d_pl + lane) & (KD_LIST - 1u)];
            const kd_u64 i = order ? (kd_u64)order[j] : j;
            const KdRInfo ri = rinfo[i];
            kd_stage_plain(row, rd.seq4 + rd.seq_off[i], (int32_t)(ri.gstart - (uint32_t)s0),
                           (int32_t)(ri.span_cls >> KD_SPAN_SHIFT), bad);
        }
        KD_WAVE_SYNC();
        kd_accumulate(col, nr, aw);
        KD_WAVE_SYNC();
        d_pl += nr;
    };
    // nr reads with clips / indels: CIGAR walk, weights rows; then the clip rows in two compacted sub-passes
    auto cplx_batch = [&](uint32_t nr) {
        int32_t cew_s = 0, cew_q = 0, cew_n = 0, csw_s = 0, csw_q = 0, csw_n = 0, dmax = 0;
        const uint8_t *seq = rd.seq4;
        if (lane < nr) {
            const kd_u64 j = first + L.cplx[(d_cx + lane) & (KD_LIST - 1u)];
            const kd_u64 i = order ? (kd_u64)order[j] : j;
            const KdRInfo ri = rinfo[i];
            const uint32_t nc = rd.n_cig[i];
            const uint32_t *cg = rd.cigar + rd.cig_off[i];
            const KdChunk pre = kd_load_cigar4(cg, 0u, nc);
            seq = rd.seq4 + rd.seq_off[i];
            dmax = ((int32_t)rd.seq_len[i] - 1) >> 3;
            int32_t grel = (int32_t)(ri.gstart - (uint32_t)s0), q = 0;   // strip-relative reference cursor, query cursor
            const int32_t foot_end = grel + (int32_t)(ri.span_cls >> KD_SPAN_SHIFT);
            const int32_t lead = (int32_t)ri.lead;
            kd_row_trash(row);
            for (uint32_t k = 0; k < nc; k++) {
                const uint32_t cw = k == 0 ? pre.x : k == 1 ? pre.y : k == 2 ? pre.z : k == 3 ? pre.w : cg[k];
                const int32_t len = (int32_t)(cw >> 4);
                const uint32_t op = cw & 15u;
                if (op == 0 || op == 7 || op == 8) {          // kindel.py:49-54
                    const int32_t i0 = grel < 0 ? -grel : 0, i1 = (int32_t)KD_STRIP - grel < len ? (int32_t)KD_STRIP - grel : len;
                    if (i1 > i0) kd_put_run(row, seq, q + i0, KD_ROW_BIAS + grel + i0, i1 - i0, dmax, bad);
                    q += len; grel += len;
                } else if (op == 2) {                         // kindel.py:59-62
                    const int32_t a = grel < 0 ? 0 : grel, b = grel + len < (int32_t)KD_STRIP ? grel + len : (int32_t)KD_STRIP;
                    for (int32_t s = a; s < b; s++) atomicAdd(&L.dels[s], 1u);
                    grel += len;
                } else if (op == 1) {
                    q += len;
                } else if (op == 4) {
                    if (k == 0) {   // leading clip, kindel.py:64-73: its last `lead` bases lie on the sites before the read
                        cew_s = grel - lead; cew_q = len - lead; cew_n = lead;
                        q += len;
                    } else {        // non-first clip, kindel.py:74-81: the last op of a regular read that moves r
                        csw_s = grel; csw_q = q; csw_n = foot_end - grel;
                        break;
                    }
                }
                if (grel >= (int32_t)KD_STRIP) break;   // everything further right lies outside the strip
            }
        }
        KD_WAVE_SYNC();
        kd_accumulate(col, nr, aw);
        KD_WAVE_SYNC();
        for (int pass = 0; pass < 2; pass++) {
            const int32_t cs = pass ? csw_s : cew_s, cq = pass ? csw_q : cew_q, cn = pass ? csw_n : cew_n;
            const bool has = cn > 0 && cs < (int32_t)KD_STRIP && cs + cn > 0;
            const kd_u64 m = kd_ballot(has);
            const uint32_t nrow = (uint32_t)kd_popcll(m);
            if (!nrow) continue;
            if (has) {
                uint32_t *crow = &L.rows[kd_mbcnt(m) * KD_ROW_DW];
                kd_row_trash(crow);
                const int32_t i0 = cs < 0 ? -cs : 0, i1 = (int32_t)KD_STRIP - cs < cn ? (int32_t)KD_STRIP - cs : cn;
                kd_put_run(crow, seq, cq + i0, KD_ROW_BIAS + cs + i0, i1 - i0, dmax, bad);
            }
            KD_WAVE_SYNC();
            kd_accumulate(col, nrow, pass ? as : ae);
            KD_WAVE_SYNC();
        }
        d_cx += nr;
    };

    for (kd_u64 tb = first; tb < last; tb += KD_STRIP) {
        const kd_u64 j = tb + lane;
        bool isp = false, isc = false;
        if (j < last) {
            const KdRInfo ri = rinfo[order ? (kd_u64)order[j] : j];
            const kd_u64 gs = ri.gstart, span = ri.span_cls >> KD_SPAN_SHIFT;
            if ((ri.span_cls & 3u) == KD_CLS_REG && gs + span > s0 && gs - ri.lead < s1) {
                isp = (ri.span_cls & KD_INFO_PLAIN) != 0;
                isc = !isp;
            }
        }
        const kd_u64 mp = kd_ballot(isp), mc = kd_ballot(isc);
        if (isp) L.plain[(n_pl + kd_mbcnt(mp)) & (KD_LIST - 1u)] = (uint16_t)(j - first);
        if (isc) L.cplx[(n_cx + kd_mbcnt(mc)) & (KD_LIST - 1u)] = (uint16_t)(j - first);
        n_pl += (uint32_t)kd_popcll(mp); n_cx += (uint32_t)kd_popcll(mc);
        KD_WAVE_SYNC();
        if (n_pl - d_pl >= KD_STRIP) plain_batch(KD_STRIP);
        if (n_cx - d_cx >= KD_STRIP) cplx_batch(KD_STRIP);
    }
    if (n_pl > d_pl) plain_batch(n_pl - d_pl);
    if (n_cx > d_cx) cplx_batch(n_cx - d_cx);

    // flush: lane l = site s0 + l; consecutive lanes -> consecutive dwords of one channel row
    kd_acc_spill(aw); kd_acc_spill(ae); kd_acc_spill(as);
    KD_WAVE_SYNC();
    const kd_u64 g = s0 + lane;
    if (g < T.stride && kd_commit(T, g)) {
        uint32_t *t0 = T.tab + g;
#pragma unroll
        for (int f = 0; f < 5; f++) {
            if (aw.w[f]) atomicAdd(t0 + (kd_u64)(KDC_A + f) * T.stride, aw.w[f]);
            if (as.w[f]) atomicAdd(t0 + (kd_u64)(KDC_CSW + f) * T.stride, as.w[f]);
            if (ae.w[f]) atomicAdd(t0 + (kd_u64)(KDC_CEW + f) * T.stride, ae.w[f]);
        }
        const uint32_t nd = L.dels[lane];
        if (nd) atomicAdd(t0 + (kd_u64)KDC_DEL * T.stride, nd);
    }
    KD_WAVE_SYNC();   // dels / the rings are reused by the next item
}

__global__ void __launch_bounds__(KD_BLOCK, KD_STRIP_WGS)
k_strip(KdReads rd, const KdRInfo *rinfo, const uint32_t *order, KdTabs T, const kd_u64 *win_lo, const kd_u64 *win_hi,
        const kd_u64 *item_off, const uint32_t *item_win, kd_u64 items_cap, uint32_t w0, uint32_t slice, kd_u64 *status) {
    __shared__ KdStripLds lds_all[KD_WAVES_PER_BLOCK];
    const uint32_t lane = threadIdx.x & (KD_WAVE - 1), wave = threadIdx.x / KD_WAVE;
    KdStripLds &L = lds_all[wave];
    kd_u64 total = status[KDS_TOTAL_ITEMS];
    if (total > items_cap) total = items_cap;   // (k_plan_items has raised KDS_INTERNAL)
    const kd_u64 per = (total + KD_NQ - 1) / KD_NQ;
    uint32_t bad = 0;
    // own queue first (blockIdx % 8 = the XCD the workgroup runs on, a speed assumption only), then the others
    for (uint32_t qi = 0; qi < KD_NQ; qi++) {
        const uint32_t q = (blockIdx.x + qi) % KD_NQ;
        const kd_u64 qlo = (kd_u64)q * per, qhi = qlo + per < total ? qlo + per : total;
        if (qlo >= qhi) continue;
        for (;;) {
            kd_u64 it = 0;
            if (lane == 0) it = atomicAdd(&status[KDS_QUEUE0 + q], (kd_u64)KD_GRAB);
            it = kd_readfirstlane64(it) + qlo;
            if (it >= qhi) break;
            const kd_u64 it_end = it + KD_GRAB < qhi ? it + KD_GRAB : qhi;
            for (; it < it_end; it++) {
                const uint32_t w = item_win[it];
                const kd_u64 first = win_lo[w] + (it - item_off[w]) * slice;
                const kd_u64 last = first + slice < win_hi[w] ? first + slice : win_hi[w];
                kd_strip_item(rd, rinfo, order, T, L, lane, (kd_u64)(w0 + w) * KD_STRIP, first, last, bad);
            }
        }
    }
    // a base outside A,C,G,T,N inside an aligned or clipped segment: k_find_bad_base pins down the read
    if (bad & KD_BAD4) atomicAdd(&status[KDS_BAD_BASE], 1ULL);
}
