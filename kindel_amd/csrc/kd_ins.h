// kd_ins.h -- k_ins_*: insertion events -> hash multiset -> per-site unique majority / tie.
// Part of the device code of kd_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
#include "kd_common.h"

// ---------------------------------------------------------------------------------------
// Insertion multiset: insertions[site][string] += 1 (kindel.py:55-58) and
// consensus(insertions[site]) (kindel.py:420-421) -> per site: unique majority string or tie.
// ---------------------------------------------------------------------------------------
// win[site]: 0 = no insertion string, event index + 1 = the unique majority string, KD_INS_TIE = several strings
// share the top count.  Ordered so that one atomicMax per hash slot settles it (TIE beats a winner beats NONE).
#define KD_INS_NONE 0u
#define KD_INS_TIE 0xffffffffu

struct KdInsTab {
    kd_u64 *key;     // [cap] 0 = empty
    uint32_t *cnt;   // [cap]
    uint32_t *rep;   // [cap] representative event of the key: the one that claimed the slot
    uint32_t *ev_slot;  // [n_ev]
    kd_u64 cap;      // power of two
    kd_u64 seed;
    kd_u64 sites;    // G-space sites (bound of a valid event site)
};

__device__ __forceinline__ kd_u64 kd_mix64(kd_u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

__global__ void __launch_bounds__(KD_BLOCK)
k_ins_insert(KdIns ins, KdInsTab H, kd_u64 n_ev) {
    const kd_u64 e = (kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x;
    if (e >= n_ev) return;
    const uint32_t site = ins.ev_site[e], len = ins.ev_len[e];
    // a reserved slot that was never written (its read raised a reference exception half way: the batch is rejected by
    // kd_finalize anyway) holds stale data: keep it out of the table unless it is at least in bounds
    if (site == KD_EV_DROPPED || site >= H.sites || ins.ev_off[e] + len > ins.pool_cap) { H.ev_slot[e] = KD_EV_DROPPED; return; }
    const uint8_t *p = ins.pool + ins.ev_off[e];
    kd_u64 h = kd_mix64(H.seed ^ ((kd_u64)site << 32 | len));
    for (uint32_t b = 0; b < len; b++) h = (h ^ p[b]) * 0x100000001b3ULL;
    h = kd_mix64(h) | 1ULL;
    kd_u64 s = (h >> 1) & (H.cap - 1);
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
    atomicAdd(&H.cnt[s], 1u);
    H.ev_slot[e] = (uint32_t)s;
}

// exactness: every event must be byte-identical to the representative of its slot (k_ins_insert has finished: the counts
// are final).  The representative itself nominates its slot for its site: best[site] = max over the site's slots of
// (count << 32 | slot).  One thread per EVENT: nothing here is proportional to the table capacity or to the sites.
__global__ void __launch_bounds__(KD_BLOCK)
k_ins_verify_max(KdIns ins, KdInsTab H, kd_u64 n_ev, kd_u64 *best, kd_u64 *status) {
    const kd_u64 e = (kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x;
    if (e >= n_ev) return;
    const uint32_t s = H.ev_slot[e];
    if (s == KD_EV_DROPPED) return;
    const uint32_t r = H.rep[s];
    if (r == (uint32_t)e) {
        atomicMax(&best[ins.ev_site[e]], ((kd_u64)H.cnt[s] << 32) | s);
        return;
    }
    bool same = ins.ev_site[e] == ins.ev_site[r] && ins.ev_len[e] == ins.ev_len[r];
    if (same) {
        const uint8_t *a = ins.pool + ins.ev_off[e], *b = ins.pool + ins.ev_off[r];
        for (uint32_t k = 0; k < ins.ev_len[e]; k++) if (a[k] != b[k]) { same = false; break; }
    }
    if (!same) atomicAdd(&status[KDS_INS_COLLISION], 1ULL);
}
// the best slot of a site nominates its representative event; any OTHER slot of the site with the same count makes it
// a tie (kindel.py:377, :421).  One thread per event, only representatives act.
__global__ void __launch_bounds__(KD_BLOCK)
k_ins_pick(KdIns ins, KdInsTab H, kd_u64 n_ev, const kd_u64 *best, uint32_t *win) {
    const kd_u64 e = (kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x;
    if (e >= n_ev) return;
    const uint32_t s = H.ev_slot[e];
    if (s == KD_EV_DROPPED || H.rep[s] != (uint32_t)e) return;
    const uint32_t site = ins.ev_site[e];
    const kd_u64 b = best[site];
    if ((uint32_t)(b >> 32) != H.cnt[s]) return;
    atomicMax(&win[site], (uint32_t)b == s ? (uint32_t)e + 1u : KD_INS_TIE);
}
// undo what the events of the last reduction left in the hash table and in best[] / win[] (all of them are zero between
// reductions: no capacity- or site-proportional memset per kd_finalize)
__global__ void __launch_bounds__(KD_BLOCK)
k_ins_cleanup(KdIns ins, KdInsTab H, kd_u64 n_ev, kd_u64 *best, uint32_t *win) {
    const kd_u64 e = (kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x;
    if (e >= n_ev) return;
    const uint32_t s = H.ev_slot[e];
    if (s == KD_EV_DROPPED) return;
    H.key[s] = 0ULL; H.cnt[s] = 0u;
    const uint32_t site = ins.ev_site[e];
    best[site] = 0ULL; win[site] = KD_INS_NONE;
}
