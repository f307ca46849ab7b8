// kd_cns.h -- k_cns_*: per-site consensus rules, scan, byte emission.
// Part of the device code of kd_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
#include "kd_common.h"

// ---------------------------------------------------------------------------------------
// Consensus: kindel.py:384-430.  Each thread owns 4 consecutive G-space sites (one 16-byte
// load per channel), a workgroup owns 1024.
// ---------------------------------------------------------------------------------------
#define KD_CNS_PER_THREAD 4
#define KD_CNS_TILE (KD_BLOCK * KD_CNS_PER_THREAD)

struct KdCns {
    const uint32_t *seg_contig;  // [S/64] contig of each 64-site segment
    const uint32_t *ins_win;     // [S] KD_INS_NONE / KD_INS_TIE / 1 + event index of the unique majority
    uint32_t min_depth;
    uint32_t n_patches;
    const kd_u64 *patch_start, *patch_end;  // skip ranges in G-space (kindel.py:393-401)
    kd_u64 g_lo, g_hi;           // emit [g_lo, g_hi)
};

struct KdSite {
    uint32_t ins_len;  // bytes of insertion text emitted before the site's own character
    uint32_t ins_ev;   // event index when ins == 1
    uint32_t depth;    // A+C+G+T
    uint8_t ins;       // 0 none, 1 unique majority string, 2 tie -> 'N'
    uint8_t has_base;  // the site emits its own character
    uint8_t base;      // that character ('A','T','G','C','N')
    uint8_t change;    // 0, 'D', 'N', 'I'
    bool live;         // a real site of a contig inside the emit interval
};

// cbase / L: G-space base and length of the contig that owns site g's 64-site segment (looked up once per thread: its
// 4 consecutive sites share a segment)
__device__ __forceinline__ KdSite kd_site_eval(const KdTabs &T, const KdCns &C, const KdIns &ins, kd_u64 g, kd_u64 cbase, kd_u64 L,
                                               uint32_t a, uint32_t tt, uint32_t gg, uint32_t cc, uint32_t nn,
                                               uint32_t del, uint32_t ins_total, uint32_t ad_next_raw) {
    KdSite s;
    s.ins_len = 0; s.ins_ev = 0; s.depth = 0; s.ins = 0; s.has_base = 0; s.base = 'N'; s.change = 0; s.live = false;
    if (g >= T.sites || g < C.g_lo || g >= C.g_hi) return s;
    const kd_u64 p = g - cbase;
    if (p >= L) return s;  // the len-th slot and the padding emit nothing
    s.live = true;
    const kd_u64 ad = (kd_u64)a + cc + gg + tt;  // kindel.py:404 (no N)
    s.depth = (uint32_t)ad;
    for (uint32_t k = 0; k < C.n_patches; k++)
        if (g >= C.patch_start[k] && g < C.patch_end[k]) return s;  // patched / skipped: no change recorded
    const kd_u64 ad_next = (p + 1 < L) ? (kd_u64)ad_next_raw : 0;  // kindel.py:405-410
    const kd_u64 ind2 = ad < ad_next ? ad : ad_next;               // 2 * indel_threshold_freq, :412
    if (2ULL * del > ad) { s.change = 'D'; return s; }             // :413-414
    if (ad < (kd_u64)C.min_depth) { s.has_base = 1; s.change = 'N'; s.base = 'N'; return s; }  // :415-417
    if (2ULL * ins_total > ind2) {                                 // :419-422
        s.change = 'I';
        const uint32_t wv = C.ins_win[g];
        if (wv == KD_INS_TIE || wv == KD_INS_NONE) { s.ins = 2; s.ins_len = 1; }
        else { s.ins = 1; s.ins_ev = wv - 1u; s.ins_len = ins.ev_len[wv - 1u]; }
    }
    // consensus(weight): first max in A,T,G,C,N order, tie -> 'N'  (kindel.py:369-381, :423-424)
    uint32_t best = a; uint8_t bc = 'A';
    if (tt > best) { best = tt; bc = 'T'; }
    if (gg > best) { best = gg; bc = 'G'; }
    if (cc > best) { best = cc; bc = 'C'; }
    if (nn > best) { best = nn; bc = 'N'; }
    const uint32_t n_at_max = (a == best) + (tt == best) + (gg == best) + (cc == best) + (nn == best);
    s.base = (best == 0 || n_at_max > 1) ? 'N' : bc;
    s.has_base = 1;
    return s;
}

// load the 4 sites of this thread and evaluate them
__device__ __forceinline__ void kd_cns_load_eval(const KdTabs &T, const KdCns &C, const KdIns &ins, kd_u64 g0,
                                                 KdSite out[KD_CNS_PER_THREAD]) {
    uint32_t v[8][KD_CNS_PER_THREAD + 1];
    const int chs[7] = {KDC_A, KDC_T, KDC_G, KDC_C, KDC_N, KDC_DEL, KDC_INS_TOTAL};
    const kd_u64 S = T.sites;
#pragma unroll
    for (int c = 0; c < 7; c++) {
        const uint32_t *row = T.tab + (kd_u64)chs[c] * T.stride;
        if (g0 + KD_CNS_PER_THREAD <= S) {
            const uint4 x = *reinterpret_cast<const uint4 *>(row + g0);
            v[c][0] = x.x; v[c][1] = x.y; v[c][2] = x.z; v[c][3] = x.w;
        } else {
            for (int k = 0; k < KD_CNS_PER_THREAD; k++) v[c][k] = g0 + k < S ? row[g0 + k] : 0;
        }
        v[c][KD_CNS_PER_THREAD] = (c < 4 && g0 + KD_CNS_PER_THREAD < S) ? row[g0 + KD_CNS_PER_THREAD] : 0;
    }
    kd_u64 cbase = 0, L = 0;   // g0 is a multiple of KD_CNS_PER_THREAD = 4: the thread's sites lie in one 64-site segment
    if (g0 < S) { const uint32_t c = C.seg_contig[g0 >> 6]; cbase = T.contig_base[c]; L = T.contig_len[c]; }
    for (int k = 0; k < KD_CNS_PER_THREAD; k++) {
        const uint32_t adn = v[0][k + 1] + v[1][k + 1] + v[2][k + 1] + v[3][k + 1];
        out[k] = kd_site_eval(T, C, ins, g0 + k, cbase, L, v[0][k], v[1][k], v[2][k], v[3][k], v[4][k], v[5][k], v[6][k], adn);
    }
}

// pass 1: bytes emitted per 1024-site tile + per-contig min/max depth
__global__ void __launch_bounds__(KD_BLOCK)
k_cns_count(KdTabs T, KdCns C, KdIns ins, kd_u64 tile_first, kd_u64 *tile_sum, uint32_t *depth_minmax) {
    __shared__ uint32_t s_sum, s_min, s_max;
    const uint32_t t = threadIdx.x;
    if (t == 0) { s_sum = 0; s_min = 0xffffffffu; s_max = 0; }
    __syncthreads();
    const kd_u64 tile0 = (tile_first + blockIdx.x) * KD_CNS_TILE;
    const kd_u64 g0 = tile0 + (kd_u64)t * KD_CNS_PER_THREAD;
    const uint32_t cfirst = tile0 < T.sites ? C.seg_contig[tile0 >> 6] : 0;
    KdSite s[KD_CNS_PER_THREAD];
    kd_cns_load_eval(T, C, ins, g0, s);
    uint32_t sum = 0, mn = 0xffffffffu, mx = 0;
    for (int k = 0; k < KD_CNS_PER_THREAD; k++) {
        sum += s[k].ins_len + s[k].has_base;
        if (s[k].live) {
            const uint32_t c = C.seg_contig[(g0 + k) >> 6];
            if (c == cfirst) { mn = s[k].depth < mn ? s[k].depth : mn; mx = s[k].depth > mx ? s[k].depth : mx; }
            else { atomicMin(&depth_minmax[2 * c], s[k].depth); atomicMax(&depth_minmax[2 * c + 1], s[k].depth); }
        }
    }
    if (sum) atomicAdd(&s_sum, sum);
    if (mn != 0xffffffffu) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); }
    __syncthreads();
    if (t == 0) {
        tile_sum[blockIdx.x] = s_sum;
        if (s_min != 0xffffffffu) { atomicMin(&depth_minmax[2 * cfirst], s_min); atomicMax(&depth_minmax[2 * cfirst + 1], s_max); }
    }
}

// pass 2: exclusive scan of the tile sums (one workgroup), tile_off[n_tiles] = total
__global__ void __launch_bounds__(KD_BLOCK)
k_cns_scan(const kd_u64 *tile_sum, kd_u64 *tile_off, kd_u64 n_tiles) {
    __shared__ kd_u64 s_scan[KD_BLOCK];
    __shared__ kd_u64 s_carry;
    const uint32_t t = threadIdx.x;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (kd_u64 b0 = 0; b0 < n_tiles; b0 += KD_BLOCK) {
        const kd_u64 b = b0 + t;
        const kd_u64 v = b < n_tiles ? tile_sum[b] : 0;
        s_scan[t] = v;
        __syncthreads();
        for (uint32_t d = 1; d < KD_BLOCK; d <<= 1) {
            kd_u64 a = t >= d ? s_scan[t - d] : 0;
            __syncthreads();
            s_scan[t] += a;
            __syncthreads();
        }
        if (b < n_tiles) tile_off[b] = s_carry + s_scan[t] - v;
        __syncthreads();
        if (t == KD_BLOCK - 1) s_carry += s_scan[t];
        __syncthreads();
    }
    if (t == 0) tile_off[n_tiles] = s_carry;
}

// pass 3: recompute, scan inside the tile, write bytes / changes / per-contig start offsets
__global__ void __launch_bounds__(KD_BLOCK)
k_cns_emit(KdTabs T, KdCns C, KdIns ins, kd_u64 tile_first, const kd_u64 *tile_off, uint8_t *out, uint8_t *changes,
           kd_u64 *contig_off, uint32_t n_contigs, kd_u64 *patch_off) {
    __shared__ uint32_t s_scan[KD_BLOCK];
    const uint32_t t = threadIdx.x;
    const kd_u64 tile0 = (tile_first + blockIdx.x) * KD_CNS_TILE;
    const kd_u64 g0 = tile0 + (kd_u64)t * KD_CNS_PER_THREAD;
    if (blockIdx.x == 0 && t == 0) contig_off[n_contigs] = tile_off[gridDim.x];   // total length, next to the per-contig offsets
    KdSite s[KD_CNS_PER_THREAD];
    kd_cns_load_eval(T, C, ins, g0, s);
    uint32_t sum = 0;
    for (int k = 0; k < KD_CNS_PER_THREAD; k++) sum += s[k].ins_len + s[k].has_base;
    s_scan[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < KD_BLOCK; d <<= 1) {
        uint32_t a = t >= d ? s_scan[t - d] : 0;
        __syncthreads();
        s_scan[t] += a;
        __syncthreads();
    }
    kd_u64 o = tile_off[blockIdx.x] + s_scan[t] - sum;
    const char lower[17] = "=acmgrsvtwyhkdbn";
    for (int k = 0; k < KD_CNS_PER_THREAD; k++) {
        const kd_u64 g = g0 + k;
        if (g >= T.sites) break;
        changes[g] = s[k].change;
        // contig c starts at G-site contig_base[c]: record the output offset there
        if ((g & 63) == 0) {
            const uint32_t c = C.seg_contig[g >> 6];
            if (T.contig_base[c] == g) contig_off[c] = o;
        }
        for (uint32_t pk = 0; pk < C.n_patches; pk++) if (C.patch_start[pk] == g) patch_off[pk] = o;
        if (s[k].ins == 1) {
            const uint8_t *p = ins.pool + ins.ev_off[s[k].ins_ev];
            for (uint32_t b = 0; b < s[k].ins_len; b++) out[o + b] = (uint8_t)lower[p[b] & 15];  // .lower(), :421
            o += s[k].ins_len;
        } else if (s[k].ins == 2) {
            out[o++] = 'N';
        }
        if (s[k].has_base) out[o++] = s[k].base;
    }
    (void)n_contigs;
}
