// kd_ins.h -- k_ins_*: insertion events -> hash multiset -> per-site unique majority / tie.
// Part of the device code of kd_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
#include "kd_common.h"

// ---------------------------------------------------------------------------------------
// Insertion multiset: insertions[site][string] += 1 (kindel.py:55-58) and
// consensus(insertions[site]) (kindel.py:420-421) -> per site: unique majority string or tie.
// ---------------------------------------------------------------------------------------
// Per site the reduction leaves TWO maxima over the site's hash slots (k_ins_verify_max): best_a = max (count << 32 | slot),
// best_b = max (count << 32 | ~slot).  Both carry the top count; best_a names the largest slot holding it, best_b the
// smallest: the majority string is unique iff they name the same slot (kd_cns.h reads them -- round 4: no k_ins_pick pass,
// no win[] array).  0 = nothing reduced on the site.

struct KdInsTab {
    kd_u64 *key;     // [cap] 0 = empty
    uint32_t *cnt;   // [cap]
    uint32_t *rep;   // [cap] representative event of the key: the one that claimed the slot
    uint32_t *ev_slot;  // [n_ev]
    kd_u64 cap;      // power of two
    kd_u64 seed;
    kd_u64 key_mask; // ~0; the tests narrow the key for the first attempt so that the collision / re-seed path runs
    kd_u64 sites;    // G-space sites (bound of a valid event site)
};
// ev_slot[e] before k_ins_insert: KD_EV_TAKE = the event takes part (its site is flagged), KD_EV_DROPPED = it does not
#define KD_EV_TAKE 0xfffffffeu

// An insertion is emitted at a site only if 2 * ins_total > min(aligned_depth, aligned_depth_next) (kindel.py:411-412, :419);
// the deletion / min_depth tests before it can only remove sites.  On sequencing data that is a handful of sites, while
// EVERY read with an I op contributes an event: flag the sites first, reduce only the events on flagged sites.
// One thread per 4 sites (16-byte loads per channel), flag[g] = 1 / 0.
// Round 4: the kernel also leaves the words the kernels behind it start from -- the hash verification's collision counter,
// and the consensus run's per-contig depth range (min = ~0, max = 0) and output offsets -- which were a memset and an upload
// of their own between the kernels.
__global__ void __launch_bounds__(KD_BLOCK)
k_ins_flag(KdTabs T, kd_u64 g_first, kd_u64 g_end, uint8_t *flag, kd_u64 *status, kd_u64 *contig_off, uint32_t *depth_minmax,
           uint32_t n_contigs) {
    const kd_u64 gt = (kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x;
    if (gt == 0) status[KDS_INS_COLLISION] = 0ULL;
    if (depth_minmax)
        for (kd_u64 c = gt; c <= n_contigs; c += (kd_u64)gridDim.x * KD_BLOCK) {
            contig_off[c] = 0ULL;
            if (c < n_contigs) { depth_minmax[2 * c] = 0xffffffffu; depth_minmax[2 * c + 1] = 0u; }
        }
    const kd_u64 g0 = g_first + gt * 4;
    if (g0 >= g_end || !flag) return;
    uint32_t v[5][5];
    const int chs[5] = {KDC_A, KDC_T, KDC_G, KDC_C, KDC_INS_TOTAL};
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const uint32_t *row = T.tab + (kd_u64)chs[c] * T.stride;
        const uint4 x = *reinterpret_cast<const uint4 *>(row + g0);     // the allocation covers whole 1024-site tiles + slack
        v[c][0] = x.x; v[c][1] = x.y; v[c][2] = x.z; v[c][3] = x.w;
        v[c][4] = (c < 4 && g0 + 4 < T.sites) ? row[g0 + 4] : 0u;
    }
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const kd_u64 ad = (kd_u64)v[0][k] + v[1][k] + v[2][k] + v[3][k];
        const kd_u64 adn = (kd_u64)v[0][k + 1] + v[1][k + 1] + v[2][k + 1] + v[3][k + 1];   // 0 behind a contig's last site
        const kd_u64 m = ad < adn ? ad : adn;
        if (v[4][k] && 2ULL * v[4][k] > m) out |= 1u << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(flag + g0) = out;
}
__device__ __forceinline__ kd_u64 kd_mix64(kd_u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// k_ins_insert: lane per event.  An event takes part iff its site is flagged (no compaction: the others leave after two
// coalesced loads and one byte; a compacted list needed one atomic per wavefront on one counter, 0.5 ms on C4); round 4: the
// marking pass (k_ins_filter) is part of this kernel -- `flag` != NULL: decide here and record the decision in ev_slot;
// NULL (the re-seeded repeat after a hash collision): ev_slot holds KD_EV_TAKE / KD_EV_DROPPED from the first attempt.
// (events of one deep site sit next to each other in event order and carry the same insertion: neighbouring lanes with
//  the same 64-bit key probe once and add their number, kd_run_heads)
// Round 5: where a batch has FEW events for its sites (short reads: C3 has 1.3 M events on 5 M sites) the SITE TEST is made here, per
// event, from the tables themselves (ten 4-byte loads for the event's site and the one behind it) -- k_ins_flag's pass over
// every site of the shard (0.022 ms on C3 and a dependent launch: 0.03 ms of every step, a tenth of C2's) is gone, and this
// kernel leaves the words a consensus run starts from (the collision counter, the per-contig depth ranges and output offsets).
// With many events for the sites (long reads: C5 has 10 M events on 1 M sites; measured 0.172 against 0.042 + 0.008 ms)
// k_ins_flag tests every site once and the events look their site's byte up, as in rounds 2 - 4.  `first` = the first attempt of a reduction: decide and
// record the decision in ev_slot; else (the re-seeded repeat after a hash collision) ev_slot holds KD_EV_TAKE / KD_EV_DROPPED.
__device__ __forceinline__ bool kd_ins_site_emits(const KdTabs &T, kd_u64 g) {
    const uint32_t it = T.tab[(kd_u64)KDC_INS_TOTAL * T.stride + g];
    if (!it) return false;
    kd_u64 ad = 0, adn = 0;
    const int chs[4] = {KDC_A, KDC_T, KDC_G, KDC_C};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t *row = T.tab + (kd_u64)chs[c] * T.stride;
        ad += row[g];
        adn += g + 1 < T.sites ? row[g + 1] : 0u;      // (0 behind a contig's last site: its slot L holds no weights)
    }
    return 2ULL * it > (ad < adn ? ad : adn);
}
__global__ void __launch_bounds__(KD_BLOCK)
k_ins_insert(KdIns ins, KdInsTab H, kd_u64 n_ev, KdTabs T, uint32_t first, kd_u64 *status, kd_u64 *contig_off, uint32_t *depth_minmax,
             uint32_t n_contigs, const uint8_t *flag) {     // flag != NULL: k_ins_flag has tested the sites (a batch with many events for its sites: kd_engine.h)
    const kd_u64 e = (kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x;
    if (first && !flag) {     // (what k_ins_flag leaves for the kernels behind it)
        if (e == 0) { status[KDS_INS_COLLISION] = 0ULL; }
        for (kd_u64 c = e; c <= n_contigs; c += (kd_u64)gridDim.x * KD_BLOCK) {
            contig_off[c] = 0ULL;
            if (c < n_contigs) { depth_minmax[2 * c] = 0xffffffffu; depth_minmax[2 * c + 1] = 0u; }
        }
    }
    bool take = false;
    uint32_t site = 0, len = 0;
    if (e < n_ev) {
        if (first) {
            site = ins.ev_site[e]; len = ins.ev_len[e];
            // a reserved slot that was never written (its read raised a reference exception half way: the batch is rejected by
            // kd_finalize anyway) holds stale data: keep it out unless it is at least in bounds
            take = site != KD_EV_DROPPED && site < H.sites && ins.ev_off[e] + len <= ins.pool_cap &&
                   (flag ? flag[site] != 0 : (kd_commit(T, site) && kd_ins_site_emits(T, site)));
            if (!take) H.ev_slot[e] = KD_EV_DROPPED;
        } else {
            take = H.ev_slot[e] == KD_EV_TAKE;
            if (take) { site = ins.ev_site[e]; len = ins.ev_len[e]; }
        }
    }
    kd_u64 h = 0;
    if (take) {
        const uint8_t *p = ins.pool + ins.ev_off[e];
        h = kd_mix64(H.seed ^ ((kd_u64)site << 32 | len));
        for (uint32_t b = 0; b < len; b++) h = (h ^ p[b]) * 0x100000001b3ULL;
        h = (kd_mix64(h) & H.key_mask) | 1ULL;
    }
    uint32_t head_lane;
    const uint32_t run = kd_run_heads(take, h, head_lane);
    kd_u64 s = 0;
    if (run) {
        s = (h >> 1) & (H.cap - 1);
        for (;;) {
            kd_u64 cur = H.key[s];
            if (cur == 0) {
                cur = atomicCAS(&H.key[s], 0ULL, h);
                // the event that claims the slot is its representative (any member would do: k_ins_verify proves all
                // members byte-identical); a plain store instead of one more scattered atomic per event
                if (cur == 0) { H.rep[s] = (uint32_t)e; break; }
            }
            if (cur == h) break;
            s = (s + 1) & (H.cap - 1);
        }
        atomicAdd(&H.cnt[s], run);
    }
    s = kd_shfl64(s, head_lane);
    if (take) H.ev_slot[e] = (uint32_t)s;
}

// The three kernels below are chains of dependent scattered loads (event -> slot -> representative -> site -> counter):
// every thread handles KD_INS_PER_THREAD events and issues each level of the chain for all of them before using any.
#define KD_INS_PER_THREAD 4
#define KD_INS_CHUNK (KD_BLOCK * KD_INS_PER_THREAD)

// exactness: every event must be byte-identical to the representative of its slot (k_ins_insert has finished: the counts
// are final).  The representative itself nominates its slot for its site: best[site] = max over the site's slots of
// (count << 32 | slot).  Per EVENT: nothing here is proportional to the table capacity or to the sites.
__global__ void __launch_bounds__(KD_BLOCK)
k_ins_verify_max(KdIns ins, KdInsTab H, kd_u64 n_ev, kd_u64 *best_a, kd_u64 *best_b, kd_u64 *status) {
    const kd_u64 k0 = (kd_u64)blockIdx.x * KD_INS_CHUNK + threadIdx.x;
    uint32_t ev[KD_INS_PER_THREAD], s[KD_INS_PER_THREAD], r[KD_INS_PER_THREAD], site[KD_INS_PER_THREAD], cnt[KD_INS_PER_THREAD];
#pragma unroll
    for (int k = 0; k < KD_INS_PER_THREAD; k++) { const kd_u64 j = k0 + (kd_u64)k * KD_BLOCK; ev[k] = j < n_ev ? (uint32_t)j : 0xffffffffu; }
#pragma unroll
    for (int k = 0; k < KD_INS_PER_THREAD; k++) s[k] = ev[k] != 0xffffffffu ? H.ev_slot[ev[k]] : KD_EV_DROPPED;
#pragma unroll
    for (int k = 0; k < KD_INS_PER_THREAD; k++) {
        const kd_u64 e = ev[k];
        r[k] = 0; site[k] = 0; cnt[k] = 0;
        if (s[k] != KD_EV_DROPPED) { r[k] = H.rep[s[k]]; site[k] = ins.ev_site[e]; cnt[k] = H.cnt[s[k]]; }
    }
#pragma unroll
    for (int k = 0; k < KD_INS_PER_THREAD; k++) {
        const kd_u64 e = ev[k];
        if (s[k] == KD_EV_DROPPED) continue;
        if (r[k] == (uint32_t)e) {   // the representative nominates its slot: largest / smallest slot with the top count
            atomicMax(&best_a[site[k]], ((kd_u64)cnt[k] << 32) | s[k]);
            atomicMax(&best_b[site[k]], ((kd_u64)cnt[k] << 32) | (uint32_t)~s[k]);
            continue;
        }
        const uint32_t rr = r[k];
        bool same = site[k] == ins.ev_site[rr] && ins.ev_len[e] == ins.ev_len[rr];
        if (same) {
            const uint8_t *a = ins.pool + ins.ev_off[e], *b = ins.pool + ins.ev_off[rr];
            for (uint32_t j = 0; j < ins.ev_len[e]; j++) if (a[j] != b[j]) { same = false; break; }
        }
        if (!same) atomicAdd(&status[KDS_INS_COLLISION], 1ULL);
    }
}
// undo what the events of the last reduction left in the hash table and in best_a[] / best_b[] (all of them are zero between
// reductions: no capacity- or site-proportional memset per kd_finalize)
__global__ void __launch_bounds__(KD_BLOCK)
k_ins_cleanup(KdIns ins, KdInsTab H, kd_u64 n_ev, kd_u64 *best_a, kd_u64 *best_b,
              kd_u64 *status, kd_u64 *first_idx, kd_u64 *err_first, uint32_t *err_code, uint32_t n_contigs) {
    // (kd_reset: the status words and the per-contig first-record / first-error state on the way -- k_reset's job, one launch)
    if (status) {
        const kd_u64 i = (kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x;
        if (i < KDS_COUNT) status[i] = i == KDS_ERR_READ ? ~0ULL : 0ULL;
        if (i < n_contigs) { first_idx[i] = ~0ULL; err_first[i] = ~0ULL; err_code[i] = 0u; }
    }
    const kd_u64 k0 = (kd_u64)blockIdx.x * KD_INS_CHUNK + threadIdx.x;
    uint32_t ev[KD_INS_PER_THREAD], s[KD_INS_PER_THREAD], site[KD_INS_PER_THREAD];
#pragma unroll
    for (int k = 0; k < KD_INS_PER_THREAD; k++) {
        const kd_u64 j = k0 + (kd_u64)k * KD_BLOCK;
        ev[k] = j < n_ev ? (uint32_t)j : 0xffffffffu;
        s[k] = ev[k] != 0xffffffffu ? H.ev_slot[ev[k]] : KD_EV_DROPPED;
        site[k] = ev[k] != 0xffffffffu ? ins.ev_site[ev[k]] : 0u;
    }
#pragma unroll
    for (int k = 0; k < KD_INS_PER_THREAD; k++) {
        if (s[k] >= KD_EV_TAKE) continue;   // dropped, or marked but never inserted
        H.key[s[k]] = 0ULL; H.cnt[s[k]] = 0u;
        best_a[site[k]] = 0ULL; best_b[site[k]] = 0ULL;
        // the event takes part again should the reduction be repeated (re-seeded after a hash collision); a later
        // kd_finalize marks all events afresh (k_ins_insert with the site flags)
        H.ev_slot[ev[k]] = KD_EV_TAKE;
    }
}
