// kd_prep.h -- k_prep: classify reads, footprints, stats, insertion slots.
// Part of the device code of kd_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
#include "kd_common.h"

// ---------------------------------------------------------------------------------------
// k_prep: classify reads, compute the reference span of their table writes, count events.
// One lane per read, KD_PREP_PER_THREAD reads per lane so that the per-block reductions
// (stats, list reservations) cost one global atomic per 8192 reads.
// ---------------------------------------------------------------------------------------
#ifndef KD_PREP_PER_THREAD
#define KD_PREP_PER_THREAD 32
#endif
#define KD_PREP_CHUNK (KD_BLOCK * KD_PREP_PER_THREAD)
#define KD_COLD_REGION (KD_WAVE * KD_PREP_PER_THREAD)   // record slots per wavefront of k_prep (one per read it classifies)
#ifndef KD_PREP_UNROLL
#define KD_PREP_UNROLL 4   // reads whose loads are issued together (a divisor of KD_PREP_PER_THREAD)
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
__device__ __forceinline__ KdScan kd_scan_cigar(const uint32_t *cg, uint32_t nc, int32_t pos0, uint32_t sl, uint32_t L,
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

#ifndef KD_PREP_OCC
#define KD_PREP_OCC 4     // workgroups per CU the register budget is set for (5: measured slower)
#endif
__global__ void __launch_bounds__(KD_BLOCK, KD_PREP_OCC)
k_prep(KdReads rd, KdTabs T, KdRInfo *rinfo, KdColdRec *cold_rec, uint32_t *cold_cnt, kd_u64 *cold_evbase, kd_u64 *cold_poolbase,
       uint32_t *irreg_list, uint32_t *long_list,
       uint32_t *read_ev, kd_u64 *read_pool, kd_u64 *status, uint32_t per_thread) {
    // per_thread: reads per lane, a multiple of KD_PREP_UNROLL up to KD_PREP_PER_THREAD (the engine picks fewer for small batches so
    // that the launch still fills the chip: 32 reads per lane of a 2 M-read shard are 254 workgroups on 256 CUs)
    __shared__ kd_u64 s_red[8];       // reads, aligned, walked, ins_ops, ins_bases, n_reg, unsorted
    __shared__ uint32_t s_maxspan, s_maxlead;
    __shared__ uint32_t s_cnt[3];     // cold, irreg, long (block totals / running offsets)
    __shared__ kd_u64 s_base[5];
    __shared__ kd_u64 s_ins[2];       // insertion events / insertion bases of the block's short-CIGAR reads
    const uint32_t t = threadIdx.x;
    if (t < 8) s_red[t] = 0;
    if (t < 3) s_cnt[t] = 0;
    if (t < 2) s_ins[t] = 0;
    if (t == 0) { s_maxspan = 0; s_maxlead = 0; }
    __syncthreads();
    const kd_u64 chunk0 = (kd_u64)blockIdx.x * KD_BLOCK * per_thread;
    kd_u64 a_reads = 0, a_aligned = 0, a_walked = 0, a_ins = 0, a_insb = 0, a_reg = 0, a_unsorted = 0;
    kd_u64 a_ins_tail = 0, a_insb_tail = 0;
    uint32_t a_maxspan = 0, a_maxlead = 0, n_irreg = 0, n_long = 0;
    uint32_t m_irreg = 0, m_long = 0, m_ins = 0;  // bit `it` = this thread's it-th read is in the list
    // Compact records of the clipped / inserted reads (for k_cold_lane): every WAVEFRONT owns a region of the record array
    // (one slot per read it classifies), the lanes whose read qualifies take consecutive slots (__ballot + mbcnt) and store
    // straight to HBM: neighbouring records from neighbouring lanes, no atomics, no staging, no limit on how many reads
    // are clipped.  cold_cnt[region] = records written.
    const kd_u64 wave_region = (kd_u64)blockIdx.x * KD_WAVES_PER_BLOCK + t / KD_WAVE;
    KdColdRec *wave_rec = cold_rec + wave_region * (KD_WAVE * per_thread);
    uint32_t wcount = 0;
    // The insertion-event slots and pool bytes of these reads are handed out here too: a wavefront-wide prefix sum per read
    // gives every read its offset in the wavefront's range, the range's base follows when the block has reserved its share
    // (one returning atomic per block, as before).  Only irregular reads with insertions (rare) still go through the last loop.
    uint32_t w_ev_total = 0;
    kd_u64 w_pool_total = 0;
    uint32_t c_cached = 0xffffffffu;                             // one-entry cache of the contig table
    kd_u64 cb_cached = 0;
    int64_t L_cached = 0;
    // 4 reads per step: all of their metadata loads are issued before any is consumed
    for (int it0 = 0; it0 < (int)per_thread; it0 += KD_PREP_UNROLL) {
        uint32_t v_c[KD_PREP_UNROLL], v_pc[KD_PREP_UNROLL], v_nc[KD_PREP_UNROLL], v_fl[KD_PREP_UNROLL];
        int64_t v_pos[KD_PREP_UNROLL], v_ppos[KD_PREP_UNROLL], v_sl[KD_PREP_UNROLL];
        kd_u64 v_coff[KD_PREP_UNROLL];
        bool v_ok[KD_PREP_UNROLL];
#pragma unroll
        for (int u = 0; u < KD_PREP_UNROLL; u++) {
            const kd_u64 i = chunk0 + (kd_u64)(it0 + u) * KD_BLOCK + t;
            v_ok[u] = i < rd.n;
            const kd_u64 j = v_ok[u] ? i : 0, jp = (v_ok[u] && i > 0) ? i - 1 : j;
            v_c[u] = rd.contig[j]; v_pos[u] = rd.pos0[j];
            v_pc[u] = rd.contig[jp]; v_ppos[u] = rd.pos0[jp];
            v_sl[u] = rd.seq_len[j]; v_nc[u] = rd.n_cig[j]; v_fl[u] = rd.flag[j]; v_coff[u] = rd.cig_off[j];
        }
        // second level: the first 4 CIGAR words of each of the 4 reads, again all in flight together
        uint32_t v_cw[KD_PREP_UNROLL][4];
#pragma unroll
        for (int u = 0; u < KD_PREP_UNROLL; u++) {
            const uint32_t *cgp = rd.cigar + v_coff[u];
            const uint32_t ncu = v_ok[u] ? v_nc[u] : 0u;
#pragma unroll
            for (int k = 0; k < 4; k++) v_cw[u][k] = (uint32_t)k < ncu ? cgp[k] : 0u;
        }
#pragma unroll
        for (int u = 0; u < KD_PREP_UNROLL; u++) {
            const bool ok = v_ok[u];   // (no early exit for a lane past the end of the batch: the wavefront votes below)
            const int it = it0 + u;
            const kd_u64 i = chunk0 + (kd_u64)it * KD_BLOCK + t;
            const uint32_t c = v_c[u];
            if (c != c_cached) { c_cached = c; cb_cached = T.contig_base[c]; L_cached = (int64_t)T.contig_len[c]; }
            const int64_t pos0 = v_pos[u];
            // first record of its contig in the file (any record with that RNAME counts, kindel.py:143-145): only where
            // the contig changes from one record to the next can a contig appear for the first time
            if (ok && (i == 0 || v_pc[u] != c)) atomicMin(&T.first_idx[c], rd.base_index + i);
            const kd_u64 gkey = cb_cached + (kd_u64)(pos0 > 0 ? pos0 : 0);
            {   // sortedness of G-start over ALL reads of the batch (window ranges rely on it)
                const kd_u64 pcb = v_pc[u] == c ? cb_cached : T.contig_base[v_pc[u]];
                const kd_u64 pk = pcb + (kd_u64)(v_ppos[u] > 0 ? v_ppos[u] : 0);
                if (ok && pk > gkey) a_unsorted++;
            }
            const int64_t sl = v_sl[u];
            const uint32_t nc = v_nc[u];
            uint32_t cls, cold = 0, lead = 0;
            kd_u64 span = 0, al = 0, n_ins_r = 0, n_insb_r = 0;
            bool has_ins = false, has_ins_tail = true;   // has_ins_tail: the read's slots come from the last loop
            if (!ok || (v_fl[u] & 4u) || sl <= 1) {
                cls = KD_CLS_SKIP;
            } else if (nc == 0) {
                cls = KD_CLS_IRREG;  // CIGAR '*': k_pileup_wave raises KD_E_CIGAR
                a_reads++;
            } else if (nc > KD_PREP_MAX_OPS) {
                cls = KD_CLS_LONG;
                a_reads++;
            } else {
                KdScan s = kd_scan_cigar(rd.cigar + v_coff[u], nc, (int32_t)pos0, (uint32_t)sl, (uint32_t)L_cached, v_cw[u]);
                cls = s.cls; cold = s.cold; span = s.span; lead = s.lead; has_ins = s.n_ins != 0; al = s.aligned;
                n_ins_r = s.n_ins; n_insb_r = s.ins_bases;
                a_reads++; a_aligned += s.aligned; a_walked += s.walked; a_ins += s.n_ins; a_insb += s.ins_bases;
            }
            if (span > 0x07ffffffULL || (cls == KD_CLS_REG && sl >= (int64_t)KD_COLD_MAX_SEQ)) { cls = KD_CLS_IRREG; span = 0; }
            if (cls == KD_CLS_REG) {
                a_reg++;
                if ((uint32_t)span > a_maxspan) a_maxspan = (uint32_t)span;
                if (lead > a_maxlead) a_maxlead = lead;
            }
            {
                const bool is_cold = cls == KD_CLS_REG && cold != 0;
                const unsigned long long cm = kd_ballot(is_cold);
                const bool reg_ins = is_cold && has_ins;
                uint32_t ev_rel = 0, pool_rel = 0;
                if (kd_ballot(reg_ins)) {   // (wave-uniform) inclusive scan of (pool bytes << 16 | events): <= 16 events per read
                    const kd_u64 v = reg_ins ? ((kd_u64)n_insb_r << 16) | n_ins_r : 0ULL;
                    kd_u64 incl = v;
#pragma unroll
                    for (uint32_t d = 1; d < KD_WAVE; d <<= 1) {
                        const kd_u64 up = kd_shfl_up64(incl, d);
                        if ((t & (KD_WAVE - 1)) >= d) incl += up;
                    }
                    const kd_u64 tot = kd_shfl64(incl, KD_WAVE - 1);
                    ev_rel = w_ev_total + (uint32_t)((incl - v) & 0xffffu);
                    pool_rel = (uint32_t)(w_pool_total + ((incl - v) >> 16));
                    w_ev_total += (uint32_t)(tot & 0xffffu);
                    w_pool_total += tot >> 16;
                }
                if (is_cold) {
                    KdColdRec cr;
                    cr.cig_off = v_coff[u]; cr.read = (uint32_t)i; cr.pos0 = (uint32_t)pos0; cr.contig = c;
                    cr.len_ops = (uint32_t)sl | (nc << 20) | (has_ins ? KD_COLD_HAS_INS : 0u);
                    cr.ev_rel = ev_rel; cr.pool_rel = pool_rel;
                    wave_rec[wcount + kd_mbcnt(cm)] = cr;
                }
                wcount += (uint32_t)kd_popcll(cm);
                if (reg_ins) has_ins_tail = false;
            }
            if (cls == KD_CLS_IRREG) { n_irreg++; m_irreg |= 1u << it; }
            if (cls == KD_CLS_LONG) { n_long++; m_long |= 1u << it; }
            if (has_ins && has_ins_tail) {   // (an irregular read with insertions) counts, see the last loop
                m_ins |= 1u << it; read_ev[i] = (uint32_t)n_ins_r; read_pool[i] = n_insb_r;
                a_ins_tail += n_ins_r; a_insb_tail += n_insb_r;
            }
            KdRInfo ri;
            ri.gstart = (uint32_t)gkey;
            // plain: the whole read is ONE aligned run: a single op whose aligned length is the read length
            const uint32_t plain = (cls == KD_CLS_REG && nc == 1 && !cold && al == (kd_u64)sl && span == al) ? KD_INFO_PLAIN : 0u;
            ri.span_cls = ((uint32_t)span << KD_SPAN_SHIFT) | plain | (has_ins ? KD_INFO_INS : 0u) | cold | cls;
            // pad of a REGULAR short-CIGAR read: query length | CIGAR words << 24 (k_window_coop routes on them without touching
            // seq_len / n_cig again; a regular read is shorter than KD_COLD_MAX_SEQ = 2^20 bases and has <= 16 words)
            ri.lead = lead; ri.pad = cls == KD_CLS_REG ? ((uint32_t)sl | (nc << 24)) : 0u;
            if (ok) rinfo[i] = ri;
        }
    }
    kd_u64 wave_off_ev = 0, wave_off_pool = 0;     // the wavefront's range inside the block's reservation (lane 0)
    if ((t & (KD_WAVE - 1)) == 0) {
        cold_cnt[wave_region] = wcount;
        if (wcount) atomicAdd(&s_cnt[0], wcount);
        if (w_ev_total) { wave_off_ev = atomicAdd(&s_ins[0], (kd_u64)w_ev_total); wave_off_pool = atomicAdd(&s_ins[1], w_pool_total); }
    }
    // block reduction through LDS atomics, then one global atomic per word per block
    if (a_reads) atomicAdd(&s_red[0], a_reads);
    if (a_aligned) atomicAdd(&s_red[1], a_aligned);
    if (a_walked) atomicAdd(&s_red[2], a_walked);
    if (a_ins) atomicAdd(&s_red[3], a_ins);
    if (a_insb) atomicAdd(&s_red[4], a_insb);
    if (a_reg) atomicAdd(&s_red[5], a_reg);
    if (a_unsorted) atomicAdd(&s_red[6], a_unsorted);
    if (a_maxspan) atomicMax(&s_maxspan, a_maxspan);
    if (a_maxlead) atomicMax(&s_maxlead, a_maxlead);
    // list slots: thread-local offset inside the block
    uint32_t o_irreg = n_irreg ? atomicAdd(&s_cnt[1], n_irreg) : 0;
    uint32_t o_long = n_long ? atomicAdd(&s_cnt[2], n_long) : 0;
    // insertion event / pool slots: thread-local offsets inside the block, one global reservation per block
    // (a global counter bumped per event serialises at ~11 ns per returning atomic on one address)
    const kd_u64 o_ev = a_ins_tail ? atomicAdd(&s_ins[0], a_ins_tail) : 0;
    const kd_u64 o_pool = a_ins_tail ? atomicAdd(&s_ins[1], a_insb_tail) : 0;
    __syncthreads();
    // one global atomic per word and block, issued by DIFFERENT threads (a dozen returning atomics in a row from one
    // thread are a dozen round trips the other 255 threads wait for)
    if (t == 0 && s_red[0]) atomicAdd(&status[KDS_ST_READS], s_red[0]);
    if (t == 1 && s_red[1]) atomicAdd(&status[KDS_ST_ALIGNED], s_red[1]);
    if (t == 2 && s_red[2]) atomicAdd(&status[KDS_ST_WALKED], s_red[2]);
    if (t == 3 && s_red[3]) { atomicAdd(&status[KDS_ST_INS], s_red[3]); atomicAdd(&status[KDS_B_INS_OPS], s_red[3]); }
    if (t == 4 && s_red[4]) atomicAdd(&status[KDS_B_INS_BASES], s_red[4]);
    if (t == 5) s_base[3] = s_ins[0] ? atomicAdd(&status[KDS_N_EV], s_ins[0]) : 0;
    if (t == 6) s_base[4] = s_ins[0] ? atomicAdd(&status[KDS_POOL], s_ins[1]) : 0;
    if (t == 7 && s_red[5]) atomicAdd(&status[KDS_B_N_REG], s_red[5]);
    if (t == 8 && s_red[6]) atomicAdd(&status[KDS_B_UNSORTED], s_red[6]);
    if (t == 9 && s_maxspan) atomicMax(&status[KDS_B_MAXSPAN], (kd_u64)s_maxspan);
    if (t == 10 && s_maxlead) atomicMax(&status[KDS_B_MAXLEAD], (kd_u64)s_maxlead);
    if (t == 64 && s_cnt[0]) atomicAdd(&status[KDS_B_N_COLD], (kd_u64)s_cnt[0]);
    if (t == 65) s_base[1] = s_cnt[1] ? atomicAdd(&status[KDS_B_N_IRREG], (kd_u64)s_cnt[1]) : 0;
    if (t == 66) s_base[2] = s_cnt[2] ? atomicAdd(&status[KDS_B_N_LONG], (kd_u64)s_cnt[2]) : 0;
    __syncthreads();
    if ((t & (KD_WAVE - 1)) == 0) { cold_evbase[wave_region] = s_base[3] + wave_off_ev; cold_poolbase[wave_region] = s_base[4] + wave_off_pool; }
    if (m_irreg | m_long | m_ins) {
        kd_u64 w_irreg = s_base[1] + o_irreg, w_long = s_base[2] + o_long;
        kd_u64 w_ev = s_base[3] + o_ev, w_pool = s_base[4] + o_pool;
        for (uint32_t todo = m_irreg | m_long | m_ins; todo; todo &= todo - 1) {
            const int it = __builtin_ctz(todo);
            const kd_u64 i = chunk0 + (kd_u64)it * KD_BLOCK + t;
            const uint32_t bit = 1u << it;
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
}

// (long-CIGAR reads -- k_prep_long, k_long_reduce, k_long_expand -- live in kd_long.h)
