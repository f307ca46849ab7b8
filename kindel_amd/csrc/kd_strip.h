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
#define KD_STRIP_WGS 5         // workgroups per CU the kernel is built for (20 wavefronts: LDS 5 x 32 KB, <= 96 VGPRs)
#endif
#define KD_NQ 8u               // work queues (XCDs)

// a pending PLAIN read: everything its staging needs (no dependent global loads between the scan and the base loads)
struct __attribute__((aligned(16))) KdPlainEnt { uint32_t gstart, len; kd_u64 seq_off; };

struct KdStripLds {
    KdPlainEnt plain[KD_LIST];                // ring of pending plain reads
    uint32_t rows[KD_STRIP * KD_ROW_DW];
    uint32_t cplx[KD_LIST];                   // ring of pending reads with clips / indels (read indices)
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

// ONE RUN per lane into the lane's row: row bytes [b0, b0 + n) <- codes of query bases [q0, q0 + n) of the read at `seq`
// (KD_ROW_BIAS <= b0, b0 + n <= KD_ROW_BIAS + 64; n == 0: the lane has no run).  The 18 row dwords are produced in nine
// steps t of two dwords from nine source dwords (8 bases each); the work of a step is decided for the WHOLE wavefront:
//   no lane needs source dword t   -> nothing is loaded or converted; `first`: the two dwords become trash
//   every lane covers both dwords  -> convert, realign, store
//   otherwise                      -> convert, realign, byte-select against trash (`first`) or against the row's content
// `first`: the rows are being initialised (every data dword is written); else the run is merged into rows that hold
// earlier runs of the same reads.  `active` = the lane's row is in use (idle lanes neither veto nor report bad bases).
// Reads of a coordinate-sorted batch start a few sites apart, so most steps are one of the two cheap kinds.
__device__ __forceinline__ void kd_stage_run(uint32_t *row, const uint8_t *seq, int32_t dmax, int32_t q0, int32_t b0, int32_t n,
                                             bool first, bool active, uint32_t &bad) {
    const int32_t hi_b = b0 + n;
    const int32_t delta = b0 - q0;                         // row byte of query base 0
    const int32_t fd = delta >> 2, adj = (delta & 3) ? 1 : 0;
    const uint32_t e = (4u - ((uint32_t)delta & 3u)) & 3u;
    // row dword K = alignbyte(conv[M + 1], conv[M], e) with M = K - fd - adj; conv[2d], conv[2d + 1] = source dword d
    const int32_t ds = (2 - fd - adj) >> 1;                // source dword of step 0 (floor)
    const int32_t Kb = 2 * ds - 1 + fd + adj;              // row dword of step 0: 0 or 1 (slack dwords)
    uint32_t need = 0u, full = active ? 0u : 0x1ffu;
    if (n > 0) {
        // steps whose source dword holds run bases (step t produces conv[2(ds+t)-1 .. 2(ds+t)+1] realigned, so its output
        // may lag its source by a dword), and steps whose two output dwords intersect the run
        const int32_t sA = (q0 >> 3) - ds, sB = ((q0 + n - 1) >> 3) - ds;
        const int32_t oA = (b0 - 4 * Kb) >> 3, oB = (hi_b - 1 - 4 * Kb) >> 3;
        const int32_t tA = sA < oA ? sA : oA, tB = sB > oB ? sB : oB;
        need = (2u << tB) - (1u << tA);
        const int32_t f0 = (b0 - 4 * Kb + 7) >> 3, f1 = (hi_b - 4 * Kb - 8) >> 3;   // steps whose two dwords lie inside the run
        full = f1 >= f0 ? (2u << f1) - (1u << f0) : 0u;
    }
    const uint32_t red = kd_wave_or(need | ((~full & 0x1ffu) << 16));
    const uint32_t NEED = red & 0x1ffu, FULL = ~(red >> 16) & 0x1ffu;
    uint32_t src[9];
#pragma unroll
    for (int t = 0; t < 9; t++) {
        src[t] = 0u;
        if ((NEED >> t) & 1u) {
            int32_t d = ds + t;
            d = d < 0 ? 0 : d > dmax ? dmax : d;           // a clamped dword only feeds bytes outside the run
            src[t] = reinterpret_cast<const KdWord *>(seq + 4 * (int64_t)d)->v;
        }
    }
    // the nine steps write dwords Kb .. Kb + 17; a later merge round of the same row may have the other Kb: make sure
    // the slack dword it would read (and report as "bad base" garbage) is defined
    if (first) { row[0] = KD_TRASH4; row[18] = KD_TRASH4; }
    const uint32_t A = (uint32_t)(0x80 - b0) * 0x01010101u, B = (uint32_t)(0x80 - hi_b) * 0x01010101u;
    const uint32_t pos0 = 0x03020100u + (uint32_t)Kb * 0x04040404u;
    uint32_t prev = 0u, lbad = 0u;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        uint32_t *r2 = row + Kb + 2 * t;
        if (!((NEED >> t) & 1u)) {
            if (first) { r2[0] = KD_TRASH4; r2[1] = KD_TRASH4; }
            continue;
        }
        uint32_t c0, c1;
        kd_conv8(src[t], c0, c1);
        uint32_t o0 = kd_alignbyte(c0, prev, e), o1 = kd_alignbyte(c1, c0, e);
        prev = c1;
        if (!((FULL >> t) & 1u)) {
            const uint32_t pos = pos0 + (uint32_t)t * 0x08080808u;
            const uint32_t x0 = first ? KD_TRASH4 : r2[0], x1 = first ? KD_TRASH4 : r2[1];
            o0 = kd_mask_bytes(o0, x0, pos, A, B);
            o1 = kd_mask_bytes(o1, x1, pos + 0x04040404u, A, B);
        }
        lbad |= o0 | o1;
        r2[0] = o0; r2[1] = o1;
    }
    if (active) bad |= lbad;
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

    // nr plain reads of the ring, from entry d_pl on: one run each (the whole read, cut to the strip)
    auto plain_batch = [&](uint32_t nr) {
        const bool act = lane < nr;
        const uint8_t *seq = rd.seq4;
        int32_t dmax = 0, q0 = 0, b0 = KD_ROW_BIAS, n = 0;
        if (act) {
            const KdPlainEnt en = L.plain[(d_pl + lane) & (KD_LIST - 1u)];
            const int32_t p = (int32_t)(en.gstart - (uint32_t)s0), len = (int32_t)en.len;
            seq = rd.seq4 + en.seq_off;
            dmax = (len - 1) >> 3;
            q0 = p < 0 ? -p : 0;
            b0 = KD_ROW_BIAS + p + q0;
            n = (p + len < (int32_t)KD_STRIP ? p + len : (int32_t)KD_STRIP) - (p + q0);
        }
        kd_stage_run(row, seq, dmax, q0, b0, n, true, act, bad);
        KD_WAVE_SYNC();
        kd_accumulate(col, nr, aw);
        KD_WAVE_SYNC();
        d_pl += nr;
    };
    // nr reads with clips / indels: every lane walks its CIGAR from run to run; the wavefront stages one run per lane
    // and round (a read with one indel has two), then the clip rows in two compacted sub-passes
    auto cplx_batch = [&](uint32_t nr) {
        const bool act = lane < nr;
        int32_t cew_s = 0, cew_q = 0, cew_n = 0, csw_s = 0, csw_q = 0, csw_n = 0, dmax = 0;
        const uint8_t *seq = rd.seq4;
        const uint32_t *cg = rd.cigar;
        KdChunk pre; pre.x = pre.y = pre.z = pre.w = 0u;
        uint32_t nc = 0, k = 0;
        int32_t grel = 0, q = 0, foot_end = 0, lead = 0;
        if (act) {
            const kd_u64 i = L.cplx[(d_cx + lane) & (KD_LIST - 1u)];
            const KdRInfo ri = rinfo[i];
            nc = rd.n_cig[i];
            cg = rd.cigar + rd.cig_off[i];
            pre = kd_load_cigar4(cg, 0u, nc);
            seq = rd.seq4 + rd.seq_off[i];
            dmax = ((int32_t)rd.seq_len[i] - 1) >> 3;
            grel = (int32_t)(ri.gstart - (uint32_t)s0);          // strip-relative reference cursor
            foot_end = grel + (int32_t)(ri.span_cls >> KD_SPAN_SHIFT);
            lead = (int32_t)ri.lead;
        }
        // phase 0: the weights rows, one M/=/X run per lane and round (a read with one indel has two); phases 1, 2: the
        // clip_end_weights / clip_start_weights rows of the lanes with a leading / trailing clip, compacted
#pragma nounroll
        for (int phase = 0; phase < 3; phase++) {
            uint32_t nrow = nr;
            uint32_t *trow = row;
            bool tact = act;
            int32_t cq = 0, cb = KD_ROW_BIAS, cn = 0;
            if (phase > 0) {
                const int32_t cs = phase == 2 ? csw_s : cew_s, cl = phase == 2 ? csw_n : cew_n;
                tact = cl > 0 && cs < (int32_t)KD_STRIP && cs + cl > 0;
                const kd_u64 m = kd_ballot(tact);
                nrow = (uint32_t)kd_popcll(m);
                if (!nrow) continue;
                if (tact) {
                    const int32_t i0 = cs < 0 ? -cs : 0, i1 = (int32_t)KD_STRIP - cs < cl ? (int32_t)KD_STRIP - cs : cl;
                    cq = (phase == 2 ? csw_q : cew_q) + i0; cb = KD_ROW_BIAS + cs + i0; cn = i1 - i0;
                }
                // the r-th lane with a clip takes row r; idle lanes point at the last row and stay idle in the staging
                trow = &L.rows[(tact ? kd_mbcnt(m) : KD_STRIP - 1u) * KD_ROW_DW];
            }
            for (uint32_t round = 0;; round++) {
                int32_t rq = 0, rb = KD_ROW_BIAS, rn = 0;
                if (phase > 0) {
                    if (round == 0) { rq = cq; rb = cb; rn = cn; }
                } else {
                    // advance to the lane's next M/=/X run that reaches into the strip
                    while (k < nc && rn == 0) {
                        const uint32_t cw = k == 0 ? pre.x : k == 1 ? pre.y : k == 2 ? pre.z : k == 3 ? pre.w : cg[k];
                        const int32_t len = (int32_t)(cw >> 4);
                        const uint32_t op = cw & 15u;
                        k++;
                        if (op == 0 || op == 7 || op == 8) {          // kindel.py:49-54
                            const int32_t i0 = grel < 0 ? -grel : 0, i1 = (int32_t)KD_STRIP - grel < len ? (int32_t)KD_STRIP - grel : len;
                            if (i1 > i0) { rq = q + i0; rb = KD_ROW_BIAS + grel + i0; rn = i1 - i0; }
                            q += len; grel += len;
                        } else if (op == 2) {                         // kindel.py:59-62
                            const int32_t a = grel < 0 ? 0 : grel, b = grel + len < (int32_t)KD_STRIP ? grel + len : (int32_t)KD_STRIP;
                            for (int32_t s = a; s < b; s++) atomicAdd(&L.dels[s], 1u);
                            grel += len;
                        } else if (op == 1) {
                            q += len;
                        } else if (op == 4) {
                            if (k == 1) {   // leading clip, kindel.py:64-73: its last `lead` bases lie on the sites before the read
                                cew_s = grel - lead; cew_q = len - lead; cew_n = lead;
                                q += len;
                            } else {        // non-first clip, kindel.py:74-81: the last op of a regular read that moves r
                                csw_s = grel; csw_q = q; csw_n = foot_end - grel;
                                k = nc;
                            }
                        }
                        if (grel >= (int32_t)KD_STRIP) k = nc;   // everything further right lies outside the strip
                    }
                }
                if (round > 0 && kd_ballot(rn > 0) == 0) break;
                kd_stage_run(trow, seq, dmax, rq, rb, rn, round == 0, tact, bad);
            }
            KD_WAVE_SYNC();
            if (phase == 0) kd_accumulate(col, nrow, aw);
            else if (phase == 1) kd_accumulate(col, nrow, ae);
            else kd_accumulate(col, nrow, as);
            KD_WAVE_SYNC();
        }
        d_cx += nr;
    };

    for (kd_u64 tb = first; tb < last; tb += KD_STRIP) {
        const kd_u64 j = tb + lane;
        bool isp = false, isc = false;
        KdPlainEnt en; en.gstart = 0; en.len = 0; en.seq_off = 0;
        uint32_t ridx = 0;
        if (j < last) {
            const kd_u64 i = order ? (kd_u64)order[j] : j;
            const KdRInfo ri = rinfo[i];
            const kd_u64 so = rd.seq_off[i];
            const kd_u64 gs = ri.gstart, span = ri.span_cls >> KD_SPAN_SHIFT;
            if ((ri.span_cls & 3u) == KD_CLS_REG && gs + span > s0 && gs - ri.lead < s1) {
                isp = (ri.span_cls & KD_INFO_PLAIN) != 0;
                isc = !isp;
                en.gstart = ri.gstart; en.len = (uint32_t)span; en.seq_off = so;
                ridx = (uint32_t)i;
            }
        }
        const kd_u64 mp = kd_ballot(isp), mc = kd_ballot(isc);
        if (isp) L.plain[(n_pl + kd_mbcnt(mp)) & (KD_LIST - 1u)] = en;
        if (isc) L.cplx[(n_cx + kd_mbcnt(mc)) & (KD_LIST - 1u)] = ridx;
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
    if (g < T.sites && kd_commit(T, g)) {
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
            if (lane == 0) it = atomicAdd(&status[KDS_QUEUE0 + q * KDS_STRIDE], (kd_u64)KD_GRAB);
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
    // a base outside A,C,G,T,N inside an aligned or clipped segment: k_errors (kd_find_bad_base) pins down the read
    if (bad & KD_BAD4) atomicAdd(&status[KDS_BAD_BASE], 1ULL);
}
