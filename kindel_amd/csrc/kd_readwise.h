// kd_readwise.h -- k_pileup_wave (exact semantics, wavefront per read), k_cold_lane.
// Part of the device code of kd_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
#include "kd_common.h"

// ---------------------------------------------------------------------------------------
// k_pileup_wave<HOT, COLD>: one wavefront per read, exact reference semantics
// (kindel.py:40-81 incl. Python negative-index wrap-around), 32-bit atomics into HBM.
//   HOT : commit M/=/X and D tallies (weights, deletions)
//   COLD: commit soft-clip tables and emit insertion events
// <true,true> is what runs: irregular reads, unsorted batches and KD_MODE_GLOBAL.
// All control flow is wave-uniform (every value steering it comes from uniform loads).
// ---------------------------------------------------------------------------------------
template <bool HOT, bool COLD>
__global__ void __launch_bounds__(KD_BLOCK)
k_pileup_wave(KdReads rd, KdTabs T, KdIns ins, const uint32_t *list, kd_u64 n_list, const KdRInfo *rinfo,
              kd_u64 *status) {
    const uint32_t lane = threadIdx.x & (KD_WAVE - 1);
    const kd_u64 slot = (kd_u64)blockIdx.x * KD_WAVES_PER_BLOCK + (threadIdx.x / KD_WAVE);
    if (slot >= n_list) return;
    const kd_u64 i = list ? (kd_u64)list[slot] : slot;
    const int64_t sl = rd.seq_len[i];
    if ((rd.flag[i] & 4u) || sl <= 1) return;  // kindel.py:43-46
    const kd_u64 gidx = rd.base_index + i;
    const uint32_t nc = rd.n_cig[i];
    const uint32_t c = rd.contig[i];
    if (nc == 0) { if (lane == 0) kd_flag_error(T, status, c, gidx); return; }  // kindel.py:47
    const int64_t L = T.contig_len[c];
    const kd_u64 cb = T.contig_base[c];
    const uint8_t *seq = rd.seq4 + rd.seq_off[i];
    const uint32_t *cg = rd.cigar + rd.cig_off[i];
    uint32_t *tab = T.tab;
    const kd_u64 S = T.stride;
    int64_t r = rd.pos0[i], q = 0;  // kindel.py:41-42
    kd_u64 ev_next = 0, pool_next = 0;  // this read's reserved insertion slots (k_prep), loaded at its first I
    bool ev_loaded = false;
    for (uint32_t k = 0; k < nc; k++) {
        const uint32_t w = cg[k];
        const int64_t len = w >> 4;
        const uint32_t op = w & 15u;
        if (op == 0 || op == 7 || op == 8) {  // M = X  kindel.py:49-54
            if (len > 0 && (q + len > sl || r + len > L || r < -L)) { if (lane == 0) kd_flag_error(T, status, c, gidx); return; }
            if (HOT) {
                for (int64_t j = lane; j < len; j += KD_WAVE) {
                    int64_t idx = r + j;
                    if (idx < 0) idx += L;
                    const uint32_t ch = kd_chan(kd_nib(seq, q + j));
                    const kd_u64 g = cb + (kd_u64)idx;
                    if (ch == 7u) kd_flag_error(T, status, c, gidx);
                    else if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)ch * S + g], 1u);
                }
            }
            r += len; q += len;
        } else if (op == 1) {  // I  kindel.py:55-58
            if (r > L || r < -(L + 1)) { if (lane == 0) kd_flag_error(T, status, c, gidx); return; }
            if (COLD) {
                if (!ev_loaded) { ev_next = ins.read_ev[i]; pool_next = ins.read_pool[i]; ev_loaded = true; }
                const int64_t q0 = q < sl ? q : sl, q1 = q + len < sl ? q + len : sl;
                const kd_u64 n = (kd_u64)(q1 - q0);
                const kd_u64 e = ev_next, po = pool_next;
                ev_next += 1; pool_next += n;
                if (lane == 0) {
                    const int64_t idx = r < 0 ? r + L + 1 : r;
                    const kd_u64 g = cb + (kd_u64)idx;
                    if (e >= ins.ev_cap || po + n > ins.pool_cap) {
                        atomicAdd(&status[KDS_INTERNAL], 1ULL);
                    } else if (kd_commit(T, g)) {
                        ins.ev_site[e] = (uint32_t)g; ins.ev_len[e] = (uint32_t)n; ins.ev_off[e] = po;
                        for (kd_u64 b = 0; b < n; b++) ins.pool[po + b] = (uint8_t)kd_nib(seq, q0 + (int64_t)b);
                        atomicAdd(&tab[(kd_u64)KDC_INS_TOTAL * S + g], 1u);
                    } else {
                        ins.ev_site[e] = KD_EV_DROPPED; ins.ev_len[e] = 0; ins.ev_off[e] = po;  // other shard's site
                    }
                }
            }
            q += len;
        } else if (op == 2) {  // D  kindel.py:59-62
            if (len > 0 && (r + len - 1 > L || r < -(L + 1))) { if (lane == 0) kd_flag_error(T, status, c, gidx); return; }
            if (HOT) {
                for (int64_t j = lane; j < len; j += KD_WAVE) {
                    int64_t idx = r + j;
                    if (idx < 0) idx += L + 1;
                    const kd_u64 g = cb + (kd_u64)idx;
                    if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)KDC_DEL * S + g], 1u);
                }
            }
            r += len;
        } else if (op == 4) {  // S
            if (k == 0) {  // kindel.py:64-73
                if (r > L || r < -(L + 1) || len > sl) { if (lane == 0) kd_flag_error(T, status, c, gidx); return; }
                if (COLD) {
                    if (lane == 0) {
                        const kd_u64 g = cb + (kd_u64)(r < 0 ? r + L + 1 : r);
                        if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)KDC_CLIP_ENDS * S + g], 1u);
                    }
                    for (int64_t j = lane; j < len; j += KD_WAVE) {
                        const int64_t rel = r - len + j;
                        if (rel >= 0) {
                            const uint32_t ch = kd_chan(kd_nib(seq, j));
                            const kd_u64 g = cb + (kd_u64)rel;
                            if (ch == 7u) kd_flag_error(T, status, c, gidx);
                            else if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)(KDC_CEW + ch) * S + g], 1u);
                        }
                    }
                }
                q += len;
            } else {  // kindel.py:74-81
                const int64_t x = r - 1;
                if (x > L || x < -(L + 1)) { if (lane == 0) kd_flag_error(T, status, c, gidx); return; }
                const int64_t n_adv = r < L ? (len < L - r ? len : L - r) : 0;
                // (the reference looks seq[q_pos] up in EVERY one of the op's `len` turns, also when r_pos has reached the contig's end --
                // and in none when len is 0: a "0S" behind a query cursor that an insertion carried past the read's end is legal)
                if (len > 0 && (n_adv > sl - q || (len > n_adv && q + n_adv >= sl) || (n_adv > 0 && r < -L))) {
                    if (lane == 0) kd_flag_error(T, status, c, gidx);
                    return;
                }
                if (COLD) {
                    if (lane == 0) {
                        const kd_u64 g = cb + (kd_u64)(x < 0 ? x + L + 1 : x);
                        if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)KDC_CLIP_STARTS * S + g], 1u);
                    }
                    for (int64_t j = lane; j < n_adv; j += KD_WAVE) {
                        int64_t idx = r + j;
                        if (idx < 0) idx += L;
                        const uint32_t ch = kd_chan(kd_nib(seq, q + j));
                        const kd_u64 g = cb + (kd_u64)idx;
                        if (ch == 7u) kd_flag_error(T, status, c, gidx);
                        else if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)(KDC_CSW + ch) * S + g], 1u);
                    }
                }
                r += n_adv; q += n_adv;
            }
        }
        // H, N, P, anything else: ignored entirely
    }
    (void)rinfo;
}

// k_cold_lane: what is left of the soft-clip / insertion side of REGULAR reads (kindel.py:55-58, :63-81) once
// k_window has tallied the clipped bases: the clip_ends / clip_starts counters (one 32-bit atomic each) and
// the insertion events, written into the slots k_prep reserved for the read.  One LANE per read of the cold
// list.  Regular reads cannot raise and never wrap (k_prep checked), so this is plain G-space arithmetic.
// The workgroup behind the last record region is the batch's error classification (kd_errors.h): the other kernels of the
// batch have finished, this kernel flags nothing.
// kd_cold_region: the records of ONE region (= of one wavefront of k_prep), all threads of a 256-thread workgroup.  Called by
// k_cold_lane (a workgroup per region) and -- round 5 -- by the workgroups k_window's launch carries behind its persistent ones
// (kd_window.h: KdColdTail): the work fills the chip while the last windows are still being tallied.
__device__ __forceinline__ void kd_cold_region(uint32_t region, const KdReads &rd, const KdTabs &T, const KdIns &ins, const KdColdRec *rec,
                                               const uint32_t *cold_cnt, const kd_u64 *cold_evbase, const kd_u64 *cold_poolbase,
                                               uint32_t region_slots, kd_u64 *status) {
    // one workgroup per record region (= per wavefront of k_prep): cnt records, usually fewer than 256
    const uint32_t cnt = cold_cnt[region];
    const KdColdRec *reg = rec + (kd_u64)region * region_slots;
    const kd_u64 ev_base = cold_evbase[region], pool_base = cold_poolbase[region];   // the region's insertion slots
  for (uint32_t k0 = 0; k0 < cnt; k0 += KD_BLOCK) {     // (uniform trip count: the wavefront meets again behind each walk)
    const uint32_t slot = k0 + threadIdx.x;
    const bool live = slot < cnt;
    const KdColdRec cr = reg[live ? slot : 0];    // k_prep's record of the read: two coalesced 16-byte loads
    const kd_u64 i = cr.read;
    const int64_t sl = cr.len_ops & (KD_COLD_MAX_SEQ - 1u);
    const uint32_t nc = live ? ((cr.len_ops >> 20) & 31u) : 0u;
    const uint32_t c = cr.contig;
    // everything the walk may need is requested up front (first four CIGAR words in one load, the read's packed bases'
    // offset): the kernel is a chain of dependent round trips otherwise
    const uint32_t *cg = rd.cigar + cr.cig_off;
    const KdChunk pre = kd_load_cigar4(cg, 0u, nc);
    // (the packed bases are only read for insertions: a read that is merely clipped does not ask where they lie -- one request to
    // a line of its own less for half of the records)
    const uint8_t *seq = rd.seq4 + ((cr.len_ops & KD_COLD_HAS_INS) ? rd.seq_off[i] : 0ULL);
    kd_u64 ev_next = ev_base + cr.ev_rel, pool_next = pool_base + cr.pool_rel;   // (unused by a read without insertions)
    const int64_t L = T.contig_len[c];
    const kd_u64 cb = T.contig_base[c];
    uint32_t *tab = T.tab;
    const kd_u64 S = T.stride;
    int64_t r = cr.pos0, q = 0;
    // the counters this read bumps -- its clip_ends site, its clip_starts site, the site of its first insertion -- are
    // committed behind the walk, where neighbouring lanes (reads sorted by position) aiming at the same site add once
    const kd_u64 NONE = ~0ULL;
    kd_u64 g_ce = NONE, g_cs = NONE, g_in = NONE;
    for (uint32_t k = 0; k < nc; k++) {
        const uint32_t w = k == 0 ? pre.x : k == 1 ? pre.y : k == 2 ? pre.z : k == 3 ? pre.w : cg[k];
        const int64_t len = w >> 4;
        const uint32_t op = w & 15u;
        if (op == 0 || op == 7 || op == 8) { r += len; q += len; }
        else if (op == 2) { r += len; }
        else if (op == 1) {
            const int64_t q0 = q < sl ? q : sl, q1 = q + len < sl ? q + len : sl;
            const kd_u64 n = (kd_u64)(q1 - q0);
            const kd_u64 e = ev_next, po = pool_next;
            ev_next += 1; pool_next += n;
            const kd_u64 g = cb + (kd_u64)r;  // 0 <= r <= L for a regular read
            if (e >= ins.ev_cap || po + n > ins.pool_cap) {
                atomicAdd(&status[KDS_INTERNAL], 1ULL);
            } else if (kd_commit(T, g)) {
                ins.ev_site[e] = (uint32_t)g; ins.ev_len[e] = (uint32_t)n; ins.ev_off[e] = po;
                for (kd_u64 b = 0; b < n; b++) ins.pool[po + b] = (uint8_t)kd_nib(seq, q0 + (int64_t)b);
                if (g_in == NONE) g_in = g;
                else atomicAdd(&tab[(kd_u64)KDC_INS_TOTAL * S + g], 1u);
            } else {
                ins.ev_site[e] = KD_EV_DROPPED; ins.ev_len[e] = 0; ins.ev_off[e] = po;
            }
            q += len;
        } else if (op == 4) {
            if (k == 0) {  // kindel.py:64-73
                const kd_u64 g = cb + (kd_u64)r;
                if (kd_commit(T, g)) g_ce = g;
                // query bases [xa, len) land on sites r - len + x  (those with r - len + x >= 0)
                // (clip_end_weights of these bases are tallied by k_window)
                q += len;
            } else {  // kindel.py:74-81; regular: the last op that touches r
                const int64_t x = r - 1;
                const kd_u64 g = cb + (kd_u64)(x < 0 ? x + L + 1 : x);
                if (kd_commit(T, g)) {              // (a regular read has one such clip)
                    if (g_cs == NONE) g_cs = g;
                    else atomicAdd(&tab[(kd_u64)KDC_CLIP_STARTS * S + g], 1u);
                }
                const int64_t n_adv = r < L ? (len < L - r ? len : L - r) : 0;
                // query bases [q, q + n_adv) land on sites r + (x - q)
                // clip_start_weights are tallied by k_window (LDS)
                r += n_adv; q += n_adv;
            }
        }
    }
    uint32_t hl, n;
    if ((n = kd_run_heads(g_ce != NONE, g_ce, hl))) atomicAdd(&tab[(kd_u64)KDC_CLIP_ENDS * S + g_ce], n);
    if ((n = kd_run_heads(g_cs != NONE, g_cs, hl))) atomicAdd(&tab[(kd_u64)KDC_CLIP_STARTS * S + g_cs], n);
    if ((n = kd_run_heads(g_in != NONE, g_in, hl))) atomicAdd(&tab[(kd_u64)KDC_INS_TOTAL * S + g_in], n);
  }
}
__global__ void __launch_bounds__(KD_BLOCK)
k_cold_lane(KdReads rd, KdTabs T, KdIns ins, const KdColdRec *rec, const uint32_t *cold_cnt, const kd_u64 *cold_evbase,
            const kd_u64 *cold_poolbase, uint32_t region_slots, kd_u64 *status, const KdRInfo *rinfo, uint32_t n_contigs) {
    if (blockIdx.x + 1 == gridDim.x) { kd_errors(rd, T, rinfo, n_contigs, status, true); return; }
    kd_cold_region(blockIdx.x, rd, T, ins, rec, cold_cnt, cold_evbase, cold_poolbase, region_slots, status);
}

// k_cold_slots: the same records -> read_ev[] / read_pool[] of the reads with insertions.  Only for the paths that walk regular
// reads through k_pileup_wave (KD_MODE_GLOBAL, a batch without window work), which looks a read's slots up by read index.
__global__ void __launch_bounds__(KD_BLOCK)
k_cold_slots(const KdColdRec *rec, const uint32_t *cold_cnt, const kd_u64 *cold_evbase, const kd_u64 *cold_poolbase,
             uint32_t region_slots, uint32_t *read_ev, kd_u64 *read_pool) {
    const uint32_t cnt = cold_cnt[blockIdx.x];
    const KdColdRec *reg = rec + (kd_u64)blockIdx.x * region_slots;
    for (uint32_t k = threadIdx.x; k < cnt; k += KD_BLOCK) {
        const KdColdRec cr = reg[k];
        if (!(cr.len_ops & KD_COLD_HAS_INS)) continue;
        read_ev[cr.read] = (uint32_t)(cold_evbase[blockIdx.x] + cr.ev_rel);
        read_pool[cr.read] = cold_poolbase[blockIdx.x] + cr.pool_rel;
    }
}
