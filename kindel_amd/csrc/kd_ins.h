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
    if (site == KD_EV_DROPPED) { H.ev_slot[e] = KD_EV_DROPPED; return; }
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

// exactness: every event must be byte-identical to the representative of its slot
__global__ void __launch_bounds__(KD_BLOCK)
k_ins_verify(KdIns ins, KdInsTab H, kd_u64 n_ev, kd_u64 *status) {
    const kd_u64 e = (kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x;
    if (e >= n_ev || H.ev_slot[e] == KD_EV_DROPPED) return;
    const uint32_t r = H.rep[H.ev_slot[e]];
    if (r == (uint32_t)e) return;
    bool same = ins.ev_site[e] == ins.ev_site[r] && ins.ev_len[e] == ins.ev_len[r];
    if (same) {
        const uint8_t *a = ins.pool + ins.ev_off[e], *b = ins.pool + ins.ev_off[r];
        for (uint32_t k = 0; k < ins.ev_len[e]; k++) if (a[k] != b[k]) { same = false; break; }
    }
    if (!same) atomicAdd(&status[KDS_INS_COLLISION], 1ULL);
}

// pass 1: best[site] = max over slots of (count << 32 | slot)
__global__ void __launch_bounds__(KD_BLOCK)
k_ins_site_max(KdIns ins, KdInsTab H, kd_u64 *best) {
    const kd_u64 s = (kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x;
    if (s >= H.cap || H.key[s] == 0) return;
    atomicMax(&best[ins.ev_site[H.rep[s]]], ((kd_u64)H.cnt[s] << 32) | s);
}
// pass 2: the best slot of a site nominates its representative event; any OTHER slot of the site with the same
// count makes it a tie (kindel.py:377, :421)
__global__ void __launch_bounds__(KD_BLOCK)
k_ins_site_pick(KdIns ins, KdInsTab H, const kd_u64 *best, uint32_t *win) {
    const kd_u64 s = (kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x;
    if (s >= H.cap || H.key[s] == 0) return;
    const uint32_t rep = H.rep[s];
    const uint32_t site = ins.ev_site[rep];
    const kd_u64 b = best[site];
    if ((uint32_t)(b >> 32) != H.cnt[s]) return;
    atomicMax(&win[site], (uint32_t)b == (uint32_t)s ? rep + 1u : KD_INS_TIE);
}
