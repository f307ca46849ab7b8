// kd_prep.h -- k_prep: classify reads, footprints, stats, insertion slots.
// Part of the device code of kd_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
#include "kd_common.h"

// ---------------------------------------------------------------------------------------
// k_prep: classify reads, compute the reference span of their table writes, count events.
// One lane per read, one wavefront per workgroup, KD_PREP_PER_THREAD reads per lane so that the
// wavefront's totals (stats, list / slot reservations) cost one atomic per counter and 4096 reads.
// ---------------------------------------------------------------------------------------
#ifndef KD_PREP_PER_THREAD
#define KD_PREP_PER_THREAD 64
#endif
#define KD_COLD_REGION (KD_WAVE * KD_PREP_PER_THREAD)   // record slots per wavefront of k_prep (one per read it classifies)
#ifndef KD_PREP_UNROLL
#define KD_PREP_UNROLL 2   // reads whose loads are issued together (a divisor of KD_PREP_PER_THREAD; 4: 128 registers + scratch, 0.236 vs 0.219 ms)
#endif
#define KD_PREP_MAX_OPS 16

// result of scanning one CIGAR
struct KdScan {
    uint32_t cls, cold, lead, n_ins, ins_bases;
    kd_u64 span, aligned, walked;
};

// Serial scan of ops [0, nc) of a read with at most KD_PREP_MAX_OPS words.  "Regular" means: k_window / the COLD pass can
// process the read with plain G-space arithmetic and no Python wrap-around or exception can occur (bad bases aside).
// `pre` = the first 4 CIGAR words, already in registers (loaded together with those of other reads).
// The rules are those of kindel.py:40-81 seen from the cursors; the arithmetic is branch-free per op (the lanes of a
// wavefront hold different op kinds: a branch per kind runs every branch; round 2's version did that in 64-bit integers
// throughout, 410 lane-instructions per read) and 32-bit but for one value:
//   rem  sites left between the reference cursor and the contig's end (L - r; 64-bit, may be negative): every rule about r
//        is a comparison against it;
//   q    query cursor CLAMPED to the read length (qc = min(q, sl)) + `over` (q > sl): what min(q, sl), q + len > sl and
//        sl - q need, exact for any length (an irregular read's insertion slots are reserved from these counts).
#ifndef KD_SCAN_EXACT_INLINE
#define KD_SCAN_EXACT_INLINE __forceinline__   // (out of line -- one copy, called for the few reads kd_scan_cigar_inside leaves over -- measured 0.50 vs 0.37 ms: the call's spills)
#endif
__device__ KD_SCAN_EXACT_INLINE KdScan kd_scan_cigar(const uint32_t *cg, uint32_t nc, int32_t pos0, uint32_t sl, uint32_t L,
                                                     const uint32_t *pre) {
    KdScan s;
    s.cold = 0; s.lead = 0; s.n_ins = 0; s.ins_bases = 0; s.aligned = 0; s.walked = 0;
    const int64_t rem0 = (int64_t)L - (int64_t)pos0;
    int64_t rem = rem0, rem_hot = rem0;          // rem_hot: rem behind the last op that wrote (M, D, non-first S)
    bool regular = pos0 >= 0, seen_nfs = false, over = false;
    uint32_t qc = 0;
    for (uint32_t k = 0; k < nc; k++) {
        const uint32_t c = k < 4 ? (k == 0 ? pre[0] : k == 1 ? pre[1] : k == 2 ? pre[2] : pre[3]) : cg[k];
        const uint32_t len = c >> 4, op = c & 15u;
        const bool m = (0x181u >> op) & 1u, i_ = op == 1, d = op == 2, sc = op == 4, sf = sc && k == 0, sn = sc && k != 0;
        const uint32_t room = sl - qc;                          // query bases left (qc <= sl)
        // non-first clip: sites it advances over (kindel.py:74-81): min(len, L - r) while r < L
        const uint32_t n_adv = rem > 0 ? ((int64_t)len < rem ? len : (uint32_t)rem) : 0u;
        bool irr = seen_nfs && (m || i_ || d || sn);
        irr = irr || (m && ((int64_t)len > rem || over || len > room));            // r + len > L, q + len > sl
        irr = irr || (i_ && rem < 0);                                               // r > L
        irr = irr || (d && (int64_t)len > rem + 1);                                 // r + len > L + 1
        irr = irr || (sf && (rem < 0 || len > sl));
        irr = irr || (sn && (rem < -1 || over || n_adv > room || (len > n_adv && n_adv >= room)));   // clip_starts[r - 1]; q + n_adv vs sl
        regular = regular && !irr;
        if (m) s.aligned += len;
        if (m || i_ || d || sc) s.walked += len;
        if (i_) { s.n_ins++; s.ins_bases += over ? 0u : (len < room ? len : room); }   // min(q + len, sl) - min(q, sl)
        if (i_ || sc) s.cold = KD_INFO_COLD;
        if (sf) s.lead = (uint32_t)(pos0 > 0 ? ((int64_t)len < (int64_t)pos0 ? (int32_t)len : pos0) : 0);
        const uint32_t radv = (m || d) ? len : sn ? n_adv : 0u;
        const uint32_t qadv = (m || i_ || sf) ? len : sn ? n_adv : 0u;
        rem -= (int64_t)radv;
        over = over || qadv > room;
        qc = over ? sl : qc + qadv;
        if (m || d || sn) rem_hot = rem;
        seen_nfs = seen_nfs || sn;
    }
    s.cls = regular ? KD_CLS_REG : KD_CLS_IRREG;
    s.span = (kd_u64)(rem0 - rem_hot);           // sites from pos0 to the end of the last write
    return s;
}

// The same scan for the read every aligner writes: one that lies INSIDE its contig and whose CIGAR consumes no more query
// than the read has.  With  rem = sites left between the reference cursor and the contig's end (L - pos0 at the start),
// Q = sum of the M / = / X / I / S lengths:
//      0 <= pos0 <= L,   every M / = / X / D no longer than rem where it stands,   Q <= sl
// no rule of kd_scan_cigar about the contig's or the query's end can fire (never `over`; min(len, room) = len; a non-first
// soft clip advances min(len, rem) <= room sites, so a read whose trailing clip hangs over the contig's end -- what an aligner
// makes of a read that does -- stays regular as kindel.py:74-81 has it), the only way left to be irregular is an op that
// writes behind a non-first soft clip, and every count is a plain sum: 32-bit adds behind one-bit tests of the op kind (16
// words of < 2^28 sum to < 2^32), fifteen instructions per op where the exact scan needs sixty and 64-bit compares.
// Returns false when the premise does not hold (an aligned run or a deletion over a contig's end, a CIGAR longer than its
// read, ...): the caller then leaves the read to kd_scan_cigar, which decides.
__device__ __forceinline__ bool kd_scan_cigar_inside(const uint32_t *cg, uint32_t nc, int32_t pos0, uint32_t sl, uint32_t L,
                                                     const uint32_t *pre, KdScan &s) {
    const uint32_t rem0 = L - (uint32_t)pos0;     // (meaningful when 0 <= pos0 <= L: checked at the end)
    uint32_t rem = rem0, Q = 0, aligned = 0, walked = 0, n_ins = 0, insb = 0, cold = 0, lead = 0, nfs = 0, bad = 0;
#define KD_SCAN_STEP(c, first)                                                                        \
    {                                                                                                 \
        const uint32_t len = (c) >> 4, op = (c) & 15u;                                                \
        const uint32_t m = (0x181u >> op) & 1u, id = (0x006u >> op) & 1u, i_ = (0x002u >> op) & 1u;   \
        const uint32_t sc = (0x010u >> op) & 1u, sn = (first) ? 0u : sc, md = m | (id & ~i_);         \
        bad |= nfs & (m | id | sn);                                                                   \
        bad |= (md && len > rem) ? 1u : 0u;                                                           \
        aligned += m ? len : 0u;                                                                      \
        walked += (m | id | sc) ? len : 0u;                                                           \
        n_ins += i_;                                                                                  \
        insb += i_ ? len : 0u;                                                                        \
        cold |= i_ | sc;                                                                              \
        rem -= md ? len : sn ? (len < rem ? len : rem) : 0u;                                          \
        Q += (m | i_ | sc) ? len : 0u;                                                                \
        nfs |= sn;                                                                                    \
    }
    {   // word 0: the only place a soft clip is a LEADING one
        const uint32_t c = nc > 0 ? pre[0] : 0x3u;     // (a padding word: N, length 0, moves nothing)
        if ((c & 15u) == 4u) lead = pos0 > 0 ? ((c >> 4) < (uint32_t)pos0 ? (c >> 4) : (uint32_t)pos0) : 0u;
        KD_SCAN_STEP(c, true)
    }
#pragma unroll
    for (uint32_t k = 1; k < 4; k++) {
        const uint32_t c = k < nc ? pre[k] : 0x3u;
        KD_SCAN_STEP(c, false)
    }
    for (uint32_t k = 4; k < nc; k++) {
        const uint32_t c = cg[k];
        KD_SCAN_STEP(c, false)
    }
#undef KD_SCAN_STEP
    s.cls = KD_CLS_REG; s.cold = cold ? KD_INFO_COLD : 0u; s.lead = lead; s.n_ins = n_ins; s.ins_bases = insb;
    s.span = rem0 - rem; s.aligned = aligned; s.walked = walked;
    return pos0 >= 0 && (uint32_t)pos0 <= L && !bad && Q <= sl;
}

#ifndef KD_PREP_OCC
#define KD_PREP_OCC 4     // wavefronts per SIMD the register budget is set for
#endif
// ---------------------------------------------------------------------------------------------------------------------
// k_prep, round 3.  Where the round-2 kernel's wavefronts spent their clocks (s_memtime marks, forced waits; C3, 0.37 ms):
// issuing loads 18 %, waiting for them 12 %, classifying 49 %, and 21 % BEHIND the loop -- at the workgroup's barriers, waiting
// for its slowest wavefront and for a dozen returning atomics.  The bytes were never the problem (the same seven arrays
// stream through a kernel that only adds them up in 0.15 ms, scripts/stream_calib.hip), and neither was the CIGAR scan (four
// fifths of its instructions removed: no change).  What was, in the order found (profiles/r03_kprep_experiments.json):
//   * the status words.  Every workgroup ended with 14 atomics on words of ONE 128-byte line, and atomics on different words
//     of a line queue up behind each other like atomics on one word (~4.4 ns each: +0.25 ms for every 4 068 wavefronts more
//     that end that way).  One line per counter (KDS_STRIDE, kd_common.h): 0.37 -> 0.32 ms for the old kernel;
//   * the barriers: one WAVEFRONT per workgroup -- no barrier, no LDS; a wavefront reduces its counts by cross-lane adds and
//     one lane per counter issues the atomic (the returning ones -- event / pool / list slots -- all in flight together);
//   * the scalar unit and the register budget: the classification was a dozen bool conditions per read, i.e. 1 500 s_and /
//     s_or / s_cbranch per 4 reads on the CU's ONE scalar unit, and 128 registers with spills.  Now the read every aligner
//     writes is decided by kd_scan_cigar_inside (sums: no rule about the contig's / the query's end can fire), conditions
//     are 0 / 1 integers combined in vector registers, and what the sums cannot decide is DEFERRED to a loop behind the
//     main one: the exact scan (kd_scan_cigar) counts it and it takes the general walk (class IRREG: k_pileup_wave follows
//     the reference statement by statement, so that is always right; only the malformed and the reads hanging over a
//     contig's end go there).  2 reads per step at 96 registers (5 wavefronts per SIMD): 0.32 -> 0.22 ms.
// ---------------------------------------------------------------------------------------------------------------------
#define KD_PREP_BLOCK KD_WAVE
__global__ void __launch_bounds__(KD_PREP_BLOCK, KD_PREP_OCC)
k_prep(KdReads rd, KdTabs T, KdRInfo *rinfo, KdColdRec *cold_rec, uint32_t *cold_cnt, kd_u64 *cold_evbase, kd_u64 *cold_poolbase,
       uint32_t *irreg_list, uint32_t *long_list,
       uint32_t *read_ev, kd_u64 *read_pool, kd_u64 *status, uint32_t per_thread, uint32_t *bound, uint32_t nb) {
    // per_thread: reads per lane, a multiple of KD_PREP_UNROLL up to KD_PREP_PER_THREAD (= 64: the bits of the list masks)
    KD_PREP_CLK_DECL      // (profiling hooks, empty in the product: kd_common.h)
    const uint32_t t = threadIdx.x;                      // = the lane
    const kd_u64 chunk0 = (kd_u64)blockIdx.x * KD_PREP_BLOCK * per_thread;
    uint32_t a_reads = 0, a_ins = 0, a_reg = 0, a_unsorted = 0, a_ins_tail = 0;    // (counts over <= 64 reads of <= 16 ops)
    kd_u64 a_aligned = 0, a_walked = 0, a_insb = 0, a_insb_tail = 0;
    uint32_t a_maxspan = 0, a_maxlead = 0;
    kd_u64 m_irreg = 0, m_long = 0, m_ins = 0, m_defer = 0;   // bit `it` = this lane's it-th read is in the list
    // Compact records of the clipped / inserted reads (for k_cold_lane): the wavefront owns a region of the record array
    // (one slot per read it classifies), the lanes whose read qualifies take consecutive slots (__ballot + mbcnt) and store
    // straight to HBM: neighbouring records from neighbouring lanes, no atomics, no staging.  cold_cnt[region] = records written.
    const kd_u64 wave_region = blockIdx.x;
    KdColdRec *wave_rec = cold_rec + wave_region * ((kd_u64)KD_WAVE * per_thread);
    uint32_t wcount = 0;
    // The insertion-event slots and pool bytes of these reads are handed out here too: a wavefront-wide prefix sum per read
    // gives every read its offset in the wavefront's range, the range's base follows from ONE returning atomic per wavefront.
    uint32_t w_ev_total = 0;
    kd_u64 w_pool_total = 0;
    const bool tail_wave = chunk0 < rd.n && chunk0 + (kd_u64)KD_PREP_BLOCK * per_thread >= rd.n;
    bool fill_on = true;                                         // this wavefront still writes k_window's boundary table
    uint32_t c_cached = 0xffffffffu;                             // one-entry cache of the contig table
    uint32_t cb_cached = 0, L_cached = 0;                         // (G-space fits 32 bits)
    for (int it0 = 0; it0 < (int)per_thread; it0 += KD_PREP_UNROLL) {
        // KD_PREP_UNROLL reads per step: all of their metadata loads are issued before any is consumed, none inside a branch
        uint32_t v_c[KD_PREP_UNROLL], v_nc[KD_PREP_UNROLL], v_fl[KD_PREP_UNROLL], v_sl[KD_PREP_UNROLL], v_pc[KD_PREP_UNROLL];
        int32_t v_pos[KD_PREP_UNROLL], v_ppos[KD_PREP_UNROLL];
        kd_u64 v_coff[KD_PREP_UNROLL];
#pragma unroll
        for (int u = 0; u < KD_PREP_UNROLL; u++) {
            const kd_u64 i = chunk0 + (kd_u64)(it0 + u) * KD_PREP_BLOCK + t;
            const kd_u64 j = i < rd.n ? i : 0;
            v_c[u] = rd.contig[j]; v_pos[u] = rd.pos0[j];
            v_sl[u] = rd.seq_len[j]; v_nc[u] = rd.n_cig[j]; v_fl[u] = rd.flag[j]; v_coff[u] = rd.cig_off[j];
            // the record in front of this one (sortedness, first record of a contig): the neighbouring lane holds it, but for
            // lane 0, which loads it (the other lanes load their own record again: the same line)
            const kd_u64 jp = (t == 0 && j > 0) ? j - 1 : j;
            v_pc[u] = rd.contig[jp]; v_ppos[u] = rd.pos0[jp];
        }
        // second level: the first 4 CIGAR words of each read as ONE unaligned 16-byte load at the read's first word -- or, for the
        // last reads of the batch, at the array's last four words (rd.n_cigar >= 4: the engine sees to it), shifted into place below
        uint32_t v_cw[KD_PREP_UNROLL][4], v_cd[KD_PREP_UNROLL];
#pragma unroll
        for (int u = 0; u < KD_PREP_UNROLL; u++) {
            const kd_u64 i = chunk0 + (kd_u64)(it0 + u) * KD_PREP_BLOCK + t;
            const kd_u64 last = rd.n_cigar - 4;
            const kd_u64 at = i < rd.n ? (v_coff[u] < last ? v_coff[u] : last) : 0;
            const kd_u64 d64 = v_coff[u] - at;
            v_cd[u] = (i < rd.n && d64 < 4) ? (uint32_t)d64 : 4u;          // 0 but for the batch's last reads (4: no word of it)
            const KdChunk w = *reinterpret_cast<const KdChunk *>(rd.cigar + at);
            v_cw[u][0] = w.x; v_cw[u][1] = w.y; v_cw[u][2] = w.z; v_cw[u][3] = w.w;
        }
#pragma unroll
        for (int u = 0; u < KD_PREP_UNROLL; u++) {
            const int it = it0 + u;
            const kd_u64 i = chunk0 + (kd_u64)it * KD_PREP_BLOCK + t;
            const bool ok = i < rd.n;   // (no early exit for a lane past the end of the batch: the wavefront votes below)
            const uint32_t c = v_c[u];
            if (c != c_cached) { c_cached = c; cb_cached = (uint32_t)T.contig_base[c]; L_cached = T.contig_len[c]; }
            const int32_t pos0 = v_pos[u];
            const uint32_t pc_n = kd_shfl_up(c, 1), ppos_n = kd_shfl_up((uint32_t)pos0, 1);
            const uint32_t pc = t == 0 ? v_pc[u] : pc_n;             // (the batch's first read is its own predecessor)
            const int32_t ppos = t == 0 ? v_ppos[u] : (int32_t)ppos_n;
            // first record of its contig in the file (any record with that RNAME counts, kindel.py:143-145): only where
            // the contig changes from one record to the next can a contig appear for the first time
            if (ok && (i == 0 || pc != c)) atomicMin(&T.first_idx[c], rd.base_index + i);
            const uint32_t gkey = cb_cached + (uint32_t)(pos0 > 0 ? pos0 : 0);
            {   // sortedness of G-start over ALL reads of the batch (window ranges rely on it)
                const uint32_t pcb = pc == c ? cb_cached : (uint32_t)T.contig_base[pc];
                const uint32_t pk = pcb + (uint32_t)(ppos > 0 ? ppos : 0);
                a_unsorted += (ok && pk > gkey) ? 1u : 0u;
                // k_window's BOUNDARY TABLE (kd_window.h: KdWq): bound[j] = first read that starts at or behind site 64 j.  A read in
                // a later granule than its predecessor fills the granules in between -- by itself when that is one or two entries
                // (deep coverage: every granule holds reads), the whole wavefront together when it is a gap in the coverage.
                // The granules in FRONT of the batch's first read and BEHIND its last one are not written at all (round 6: on a
                // shard of a strong-scaling run they are seven eighths of the table, filled by the first and the last wavefront
                // while every other one had left -- a third of this kernel's time there, scripts/exp/prep_slowest_wave.sh): two
                // status words name the table's first and last written entry, a reader clamps to them (kd_wq_bound).
                // Meaningless for an unsorted batch (which is bucket-sorted instead, the table unused) -- and there every forward
                // jump would look like a gap to fill: a wavefront that sees ONE read start in front of its predecessor stops
                // filling for good (random order: the first 64 reads it looks at), and fills of more than 64 entries draw on a
                // budget of 4 x the table (a sorted batch fills every entry once).
                if (bound) {
                    if (kd_ballot(ok && pk > gkey)) fill_on = false;          // (wave-uniform)
                    // (a POS behind the last contig's end -- the read is an IndexError further down -- must not index past the
                    // table's nb + 1 entries: both granules are clamped to nb, which leaves such a read nothing to fill)
                    const uint32_t kj = (gkey >> 6) < nb ? (gkey >> 6) : nb;
                    const uint32_t pj = (pk >> 6) < nb ? (pk >> 6) : nb;
                    const uint32_t b0 = i == 0 ? kj : pj + 1u;                // first granule to fill (the batch's first read: its own)
                    if (ok && i == 0) status[KDS_B_BOUND_LO] = (kd_u64)kj;
                    uint32_t cnt = (fill_on && ok && kj >= b0) ? kj - b0 + 1u : 0u;
                    if (cnt <= 2u) {
                        if (cnt) bound[b0] = (uint32_t)i;
                        if (cnt == 2u) bound[b0 + 1u] = (uint32_t)i;
                        cnt = 0;
                    }
                    for (unsigned long long m = kd_ballot(cnt != 0); m; m &= m - 1) {
                        const uint32_t l = (uint32_t)__builtin_ctzll(m);
                        const uint32_t f0 = kd_shfl(b0, l), fn = kd_shfl(cnt, l), fv = kd_shfl((uint32_t)i, l);
                        if (fn > 64u) {
                            kd_u64 spent = 0;
                            if (t == 0) spent = atomicAdd(&status[KDS_B_FILL], (kd_u64)fn);
                            if (kd_shfl64(spent, 0) + fn > 4ull * nb) { fill_on = false; break; }   // (spent BEFORE this fill + this fill)
                        }
                        for (uint32_t x = t; x < fn; x += KD_WAVE) bound[f0 + x] = fv;
                    }
                    // behind the batch's last read: ONE entry "none" = the number of reads (a last read in the table's last granule
                    // leaves that granule's entry as it is: the reader's clamp ends there anyway)
                    if (tail_wave && ok && i + 1 == rd.n) {
                        if (kj < nb) bound[kj + 1u] = (uint32_t)rd.n;
                        status[KDS_B_BOUND_HI1] = (kd_u64)(kj < nb ? kj + 1u : nb) + 1ull;
                    }
                }
            }
            const uint32_t sl = v_sl[u], nc = v_nc[u];
            // the read's first four words in place (the batch's last reads were loaded from in front of their first word)
            uint32_t cw[4];
            {
                const uint32_t d = v_cd[u];
                if (d) {   // (rare: at most the last four reads of the batch)
                    uint32_t sh[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++)
                        sh[k] = d == 1 ? (k + 1 < 4 ? v_cw[u][(k + 1) & 3] : 0u) : d == 2 ? (k + 2 < 4 ? v_cw[u][(k + 2) & 3] : 0u)
                                       : d == 3 ? (k + 3 < 4 ? v_cw[u][(k + 3) & 3] : 0u) : 0u;
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) v_cw[u][k] = sh[k];
                }
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) cw[k] = k < nc ? v_cw[u][k] : 0x3u;   // (behind the read's words: N, length 0)
            }
            // class: 0 / 1 integers, combined in vector registers
            const uint32_t skip = (!ok || (v_fl[u] & 4u) || sl <= 1) ? 1u : 0u;       // kindel.py:43-46
            const uint32_t star = (!skip && nc == 0) ? 1u : 0u;                        // CIGAR '*': k_pileup_wave raises KD_E_CIGAR
            const uint32_t lng = (!skip && nc > KD_PREP_MAX_OPS) ? 1u : 0u;
            const uint32_t scanned = (skip | star | lng) ^ 1u;
            KdScan s;
            const bool inside = kd_scan_cigar_inside(rd.cigar + v_coff[u], scanned ? nc : 0u, pos0, sl, L_cached, cw, s);
            // regular: decided by the sums, footprint and length within what the window kernels index with
            const uint32_t reg = (scanned && inside && s.span <= 0x07ffffffULL && sl < KD_COLD_MAX_SEQ) ? 1u : 0u;
            const uint32_t defer = (scanned && !inside) ? 1u : 0u;                     // the exact scan decides, behind the loop
            const uint32_t irr = star | ((scanned & (reg | defer)) ^ scanned);            // scanned, inside, but too long / too wide
            const uint32_t counted = scanned & (defer ^ 1u);                           // its counts are the sums'
            a_reads += (skip ^ 1u) & (defer ^ 1u);
            a_aligned += counted ? s.aligned : 0ULL; a_walked += counted ? s.walked : 0ULL;
            a_ins += counted ? s.n_ins : 0u; a_insb += counted ? s.ins_bases : 0u;
            a_reg += reg;
            const uint32_t span = reg ? (uint32_t)s.span : 0u, lead = reg ? s.lead : 0u;
            a_maxspan = span > a_maxspan ? span : a_maxspan;
            a_maxlead = lead > a_maxlead ? lead : a_maxlead;
            const uint32_t cold = counted ? s.cold : 0u;
            const uint32_t has_ins = (counted && s.n_ins) ? 1u : 0u;
            {
                const bool is_cold = reg && cold;
                const unsigned long long cm = kd_ballot(is_cold);
                const bool reg_ins = is_cold && has_ins;
                uint32_t ev_rel = 0, pool_rel = 0;
                if (kd_ballot(reg_ins)) {   // (wave-uniform) two DPP prefix sums: events (<= 16 per read) and pool bytes (< 2^20 per
                                            // regular read: 64 of them stay below 2^32)
                    const uint32_t ve = reg_ins ? s.n_ins : 0u, vb = reg_ins ? s.ins_bases : 0u;
                    const uint32_t ie = kd_wave_scan_add(ve), ib = kd_wave_scan_add(vb);
                    ev_rel = w_ev_total + (ie - ve);
                    pool_rel = (uint32_t)(w_pool_total + (ib - vb));
                    w_ev_total += kd_readlane(ie, KD_WAVE - 1);
                    w_pool_total += kd_readlane(ib, KD_WAVE - 1);
                }
                if (is_cold) {
                    KdColdRec cr;
                    cr.cig_off = v_coff[u]; cr.read = (uint32_t)i; cr.pos0 = (uint32_t)pos0; cr.contig = c;
                    cr.len_ops = sl | (nc << 20) | (has_ins ? KD_COLD_HAS_INS : 0u);
                    cr.ev_rel = ev_rel; cr.pool_rel = pool_rel;
                    wave_rec[wcount + kd_mbcnt(cm)] = cr;
                }
                wcount += (uint32_t)kd_popcll(cm);
            }
            const kd_u64 bit = 1ULL << it;
            if (irr) m_irreg |= bit;
            if (lng) m_long |= bit;
            if (defer) m_defer |= bit;
            if (irr && has_ins) {   // (an irregular read with insertions: counted by the sums, too long for the window path) its slots
                m_ins |= bit; read_ev[i] = s.n_ins; read_pool[i] = s.ins_bases;     //   come from the last loop
                a_ins_tail += s.n_ins; a_insb_tail += s.ins_bases;
            }
            KdRInfo ri;
            ri.gstart = gkey;
            // plain: the whole read is ONE aligned run: a single op whose aligned length is the read length
            const uint32_t plain = (reg && nc == 1 && !cold && s.aligned == (kd_u64)sl && s.span == s.aligned) ? KD_INFO_PLAIN : 0u;
            const uint32_t cls = reg ? KD_CLS_REG : lng ? KD_CLS_LONG : (irr | defer) ? KD_CLS_IRREG : KD_CLS_SKIP;
            ri.span_cls = (span << KD_SPAN_SHIFT) | plain | (has_ins ? KD_INFO_INS : 0u) | cold | cls;
            // pad of a REGULAR short-CIGAR read: query length | CIGAR words << 24 (k_window_coop routes on them without touching
            // seq_len / n_cig again; a regular read is shorter than KD_COLD_MAX_SEQ = 2^20 bases and has <= 16 words)
            ri.lead = lead; ri.pad = reg ? (sl | (nc << 24)) : 0u;
            if (ok) rinfo[i] = ri;
        }
    }
    KD_PREP_CLK_LOOP_END
    // The reads the sums could not decide (rare: at a contig's end, CIGAR and read length at odds, POS 0): the exact scan counts
    // them -- an irregular read's insertion slots are reserved from its counts -- and they take the general walk.
    for (kd_u64 todo = m_defer; todo; todo &= todo - 1) {
        const int it = __builtin_ctzll(todo);
        const kd_u64 i = chunk0 + (kd_u64)it * KD_PREP_BLOCK + t;
        const uint32_t c = rd.contig[i], nc = rd.n_cig[i];
        const kd_u64 coff = rd.cig_off[i];
        uint32_t pre[4];
        for (uint32_t k = 0; k < 4; k++) pre[k] = k < nc ? rd.cigar[coff + k] : 0u;
        const KdScan s = kd_scan_cigar(rd.cigar + coff, nc, rd.pos0[i], rd.seq_len[i], T.contig_len[c], pre);
        a_reads++; a_aligned += s.aligned; a_walked += s.walked; a_ins += s.n_ins; a_insb += s.ins_bases;
        m_irreg |= 1ULL << it;
        if (s.n_ins) {
            m_ins |= 1ULL << it; read_ev[i] = s.n_ins; read_pool[i] = s.ins_bases;
            a_ins_tail += s.n_ins; a_insb_tail += s.ins_bases;
            rinfo[i].span_cls |= KD_INFO_INS | s.cold;
        } else if (s.cold) rinfo[i].span_cls |= s.cold;
    }
    // ---- the wavefront's totals: cross-lane sums, then ONE lane per counter issues the atomic ----
    const uint32_t n_irreg = (uint32_t)kd_popcll(m_irreg), n_long = (uint32_t)kd_popcll(m_long);
    const uint32_t i_irreg = kd_wave_scan_add(n_irreg), i_long = kd_wave_scan_add(n_long);      // inclusive prefix sums over the lanes
    const uint32_t i_evt = kd_wave_scan_add(a_ins_tail);
    const uint32_t t_irreg = kd_readlane(i_irreg, KD_WAVE - 1), t_long = kd_readlane(i_long, KD_WAVE - 1), t_evt = kd_readlane(i_evt, KD_WAVE - 1);
    kd_u64 i_poolt = a_insb_tail;    // (64-bit: an irregular read's insertion may be 2^28 bases long)
#pragma unroll
    for (uint32_t d = 1; d < KD_WAVE; d <<= 1) { const kd_u64 up = kd_shfl_up64(i_poolt, d); if (t >= d) i_poolt += up; }
    const kd_u64 t_poolt = kd_shfl64(i_poolt, KD_WAVE - 1);
    const uint32_t r_reads = kd_wave_sum(a_reads), r_ins = kd_wave_sum(a_ins), r_reg = kd_wave_sum(a_reg), r_uns = kd_wave_sum(a_unsorted);
    const uint32_t r_span = kd_wave_max(a_maxspan), r_lead = kd_wave_max(a_maxlead);
    kd_u64 r_al = a_aligned, r_wk = a_walked, r_ib = a_insb;
#pragma unroll
    for (uint32_t m = 1; m < KD_WAVE; m <<= 1) {
        r_al += kd_shfl64(r_al, t ^ m); r_wk += kd_shfl64(r_wk, t ^ m); r_ib += kd_shfl64(r_ib, t ^ m);
    }
    const kd_u64 ev_all = (kd_u64)w_ev_total + t_evt, pool_all = w_pool_total + t_poolt;   // regular reads' slots first, then the tail's
    kd_u64 got = 0;      // lane k's returning atomic (all in flight together)
    if (t == 0 && r_reads) atomicAdd(&status[KDS_ST_READS], (kd_u64)r_reads);
    if (t == 1 && r_al) atomicAdd(&status[KDS_ST_ALIGNED], r_al);
    if (t == 2 && r_wk) atomicAdd(&status[KDS_ST_WALKED], r_wk);
    if (t == 3 && r_ins) { atomicAdd(&status[KDS_ST_INS], (kd_u64)r_ins); atomicAdd(&status[KDS_B_INS_OPS], (kd_u64)r_ins); }
    if (t == 4 && r_ib) atomicAdd(&status[KDS_B_INS_BASES], r_ib);
    if (t == 5 && ev_all) got = atomicAdd(&status[KDS_N_EV], ev_all);
    if (t == 6 && ev_all) got = atomicAdd(&status[KDS_POOL], pool_all);
    if (t == 7 && r_reg) atomicAdd(&status[KDS_B_N_REG], (kd_u64)r_reg);
    if (t == 8 && r_uns) atomicAdd(&status[KDS_B_UNSORTED], (kd_u64)r_uns);
    if (t == 9 && r_span) atomicMax(&status[KDS_B_MAXSPAN], (kd_u64)r_span);
    if (t == 10 && r_lead) atomicMax(&status[KDS_B_MAXLEAD], (kd_u64)r_lead);
    if (t == 11 && wcount) atomicAdd(&status[KDS_B_N_COLD], (kd_u64)wcount);
    if (t == 12 && t_irreg) got = atomicAdd(&status[KDS_B_N_IRREG], (kd_u64)t_irreg);
    if (t == 13 && t_long) got = atomicAdd(&status[KDS_B_N_LONG], (kd_u64)t_long);
    const kd_u64 b_ev = kd_shfl64(got, 5), b_pool = kd_shfl64(got, 6), b_irreg = kd_shfl64(got, 12), b_long = kd_shfl64(got, 13);
    if (t == 0) { cold_cnt[wave_region] = wcount; cold_evbase[wave_region] = b_ev; cold_poolbase[wave_region] = b_pool; }
    if (m_irreg | m_long | m_ins) {
        kd_u64 w_irreg = b_irreg + (i_irreg - n_irreg), w_long = b_long + (i_long - n_long);
        kd_u64 w_ev = b_ev + w_ev_total + (i_evt - a_ins_tail), w_pool = b_pool + w_pool_total + (i_poolt - a_insb_tail);
        for (kd_u64 todo = m_irreg | m_long | m_ins; todo; todo &= todo - 1) {
            const int it = __builtin_ctzll(todo);
            const kd_u64 i = chunk0 + (kd_u64)it * KD_PREP_BLOCK + t;
            const kd_u64 bit = 1ULL << it;
            if (m_ins & bit) {  // the first pass left the read's event / base COUNTS here: turn them into its slots
                const uint32_t n_ev = read_ev[i];
                const kd_u64 n_b = read_pool[i];
                read_ev[i] = (uint32_t)w_ev; read_pool[i] = w_pool;
                w_ev += n_ev; w_pool += n_b;
            }
            if (m_irreg & bit) irreg_list[w_irreg++] = (uint32_t)i;
            if (m_long & bit) long_list[w_long++] = (uint32_t)i;
        }
    }
    KD_PREP_CLK_COMMIT(status)
}

// (long-CIGAR reads -- k_prep_long, k_long_reduce, k_long_expand -- live in kd_long.h)
