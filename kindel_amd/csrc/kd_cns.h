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
    // the insertion reduction's per-site maxima (kd_ins.h: k_ins_verify_max), shard-local, biased like the tables:
    //   best_a[g] = max over the site's hash slots of (count << 32 | slot),  best_b[g] = max of (count << 32 | ~slot);
    //   0 = no insertion string was reduced on the site.  The slot with the top count is unique iff both name the same slot.
    const kd_u64 *best_a, *best_b;
    const uint32_t *ins_rep;     // [hash capacity] representative event of a slot
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
        // consensus(insertions[pos]) (:420-421): the string with the top count, or a tie -> 'N' (:377).  Several strings
        // share the top count iff the largest and the smallest slot holding it differ.
        const kd_u64 ba = C.best_a[g];
        const uint32_t sa = (uint32_t)ba, sb = ~(uint32_t)C.best_b[g];
        if (ba == 0ULL || sa != sb) { s.ins = 2; s.ins_len = 1; }
        else { const uint32_t ev = C.ins_rep[sa]; s.ins = 1; s.ins_ev = ev; s.ins_len = ins.ev_len[ev]; }
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

// pass 1: bytes emitted per 1024-site tile + the tile's min / max A+C+G+T depth.
// The per-contig depth range is NOT reduced here with atomics: 4883 tiles x 2 device-scope atomics on the two words of one
// contig serialise at ~12 ns each (0.12 of this kernel's 0.14 ms on C3, measured); the tile writes (contig of its first
// site, min, max) and k_cns_scan folds those.  Sites of a second contig inside the tile (contig boundaries) go the atomic way.
struct KdTileMM { uint32_t contig, mn, mx, pad; };

__global__ void __launch_bounds__(KD_BLOCK)
k_cns_count(KdTabs T, KdCns C, KdIns ins, kd_u64 tile_first, kd_u64 *tile_sum, KdTileMM *tile_mm, uint32_t *depth_minmax) {
    __shared__ uint32_t s_w[3][KD_WAVES_PER_BLOCK];
    const uint32_t t = threadIdx.x, lane = t & (KD_WAVE - 1), wave = t / KD_WAVE;
    const kd_u64 tile0 = (tile_first + blockIdx.x) * KD_CNS_TILE;
    const kd_u64 g0 = tile0 + (kd_u64)t * KD_CNS_PER_THREAD;
    const uint32_t cfirst = tile0 < T.sites ? C.seg_contig[tile0 >> 6] : 0;
    KdSite s[KD_CNS_PER_THREAD];
    kd_cns_load_eval(T, C, ins, g0, s);
    // a thread's 4 sites lie in one 64-site segment = one contig; 16 consecutive lanes share the segment
    uint32_t sum = 0, mn = 0xffffffffu, mx = 0;
    for (int k = 0; k < KD_CNS_PER_THREAD; k++) {
        sum += s[k].ins_len + s[k].has_base;
        if (s[k].live) { mn = s[k].depth < mn ? s[k].depth : mn; mx = s[k].depth > mx ? s[k].depth : mx; }
    }
    const uint32_t cseg = g0 < T.sites ? C.seg_contig[g0 >> 6] : cfirst;
    if (kd_ballot(cseg != cfirst) != 0) {
        // the tile crosses into other contigs: their segments are reduced over their 16 lanes and go to the contig's words
        // directly (a few atomics per contig boundary); the lanes of the tile's first contig continue below
        uint32_t smn = mn, smx = mx;
#pragma unroll
        for (uint32_t m = 1; m < 16; m <<= 1) {
            const uint32_t a = kd_shfl_xor(smn, m), b = kd_shfl_xor(smx, m);
            smn = a < smn ? a : smn; smx = b > smx ? b : smx;
        }
        if (cseg != cfirst) {
            if ((lane & 15u) == 0 && smn != 0xffffffffu) { atomicMin(&depth_minmax[2 * cseg], smn); atomicMax(&depth_minmax[2 * cseg + 1], smx); }
            mn = 0xffffffffu; mx = 0;
        }
    }
    // wavefront reductions by shuffles, then the four wavefronts through LDS
    sum = kd_wave_sum(sum); mn = kd_wave_min(mn); mx = kd_wave_max(mx);
    if (lane == 0) { s_w[0][wave] = sum; s_w[1][wave] = mn; s_w[2][wave] = mx; }
    __syncthreads();
    if (t == 0) {
        uint32_t a = 0, b = 0xffffffffu, c = 0;
        for (uint32_t w = 0; w < KD_WAVES_PER_BLOCK; w++) { a += s_w[0][w]; b = s_w[1][w] < b ? s_w[1][w] : b; c = s_w[2][w] > c ? s_w[2][w] : c; }
        tile_sum[blockIdx.x] = a;
        KdTileMM m; m.contig = cfirst; m.mn = b; m.mx = c; m.pad = 0;
        tile_mm[blockIdx.x] = m;
    }
}

// pass 2 (one workgroup): exclusive scan of the tile sums, tile_off[n_tiles] = total; fold the tiles' depth ranges into the
// per-contig ranges (a thread merges its consecutive tiles of one contig before touching the contig's words)
// (256 threads on purpose: every thread ends with an atomicMin / atomicMax on its contig's two words, and 1024 threads are 2048
// same-address atomics -- measured 0.034 instead of 0.022 ms on C3's one contig)
__global__ void __launch_bounds__(KD_BLOCK)
k_cns_scan(const kd_u64 *tile_sum, kd_u64 *tile_off, kd_u64 n_tiles, const KdTileMM *tile_mm, uint32_t *depth_minmax) {
    __shared__ kd_u64 s_wave[KD_WAVES_PER_BLOCK];
    const uint32_t t = threadIdx.x;
    kd_u64 carry = 0;
    // each thread owns a contiguous run of tiles (contiguous -> few contig changes per thread)
    const kd_u64 per = (n_tiles + KD_BLOCK - 1) / KD_BLOCK;
    const kd_u64 b0 = (kd_u64)t * per < n_tiles ? (kd_u64)t * per : n_tiles, b1 = b0 + per < n_tiles ? b0 + per : n_tiles;
    kd_u64 mine = 0;
    uint32_t cur = 0xffffffffu, mn = 0xffffffffu, mx = 0;
    for (kd_u64 b = b0; b < b1; b++) {
        mine += tile_sum[b];
        const KdTileMM m = tile_mm[b];
        if (m.mn == 0xffffffffu) continue;        // no live site of its first contig
        if (m.contig != cur) {
            if (cur != 0xffffffffu) { atomicMin(&depth_minmax[2 * cur], mn); atomicMax(&depth_minmax[2 * cur + 1], mx); }
            cur = m.contig; mn = m.mn; mx = m.mx;
        } else { mn = m.mn < mn ? m.mn : mn; mx = m.mx > mx ? m.mx : mx; }
    }
    if (cur != 0xffffffffu) { atomicMin(&depth_minmax[2 * cur], mn); atomicMax(&depth_minmax[2 * cur + 1], mx); }
    kd_u64 total;
    const kd_u64 incl = kd_block_scan_incl(mine, s_wave, total);
    kd_u64 o = carry + incl - mine;
    for (kd_u64 b = b0; b < b1; b++) { tile_off[b] = o; o += tile_sum[b]; }
    if (t == 0) tile_off[n_tiles] = total;
}

// The per-contig fold of the tiles' depth ranges (a thread merges its consecutive tiles of one contig before touching the
// contig's words): all threads of ONE workgroup.
__device__ __forceinline__ void kd_cns_fold_mm(const KdTileMM *tile_mm, kd_u64 n_tiles, uint32_t *depth_minmax, uint32_t t, uint32_t nt) {
    const kd_u64 per = (n_tiles + nt - 1) / nt;
    const kd_u64 b0 = (kd_u64)t * per < n_tiles ? (kd_u64)t * per : n_tiles, b1 = b0 + per < n_tiles ? b0 + per : n_tiles;
    uint32_t cur = 0xffffffffu, mn = 0xffffffffu, mx = 0;
    for (kd_u64 b = b0; b < b1; b++) {
        const KdTileMM m = tile_mm[b];
        if (m.mn == 0xffffffffu) continue;        // no live site of its first contig
        if (m.contig != cur) {
            if (cur != 0xffffffffu) { atomicMin(&depth_minmax[2 * cur], mn); atomicMax(&depth_minmax[2 * cur + 1], mx); }
            cur = m.contig; mn = m.mn; mx = m.mx;
        } else { mn = m.mn < mn ? m.mn : mn; mx = m.mx > mx ? m.mx : mx; }
    }
    if (cur != 0xffffffffu) { atomicMin(&depth_minmax[2 * cur], mn); atomicMax(&depth_minmax[2 * cur + 1], mx); }
}

// pass 2 (round 4: the last pass): recompute, scan inside the tile, write bytes / changes / per-contig start offsets.
// tile_off == NULL (up to KD_CNS_SELF_SCAN tiles): there is no scan kernel between the passes -- every workgroup sums the byte
// counts of the tiles in front of its own (<= 64 coalesced loads per thread out of the L2) and workgroup 0 folds the tiles'
// depth ranges into the per-contig ranges on the way; k_cns_scan (one workgroup, 19 us on C3 for a prefix sum of 4 883 numbers,
// most of it the dispatch and its dependent round trips) is only launched beyond that, where the quadratic sum would cost more.
#define KD_CNS_SELF_SCAN 16384u
// Round 6: the tile's bytes are STAGED in LDS and leave the workgroup as whole dwords -- to the device buffer and, when the caller's
// output buffer is pinned host memory the device can address (host_out != NULL: kd_step / kd_finish), straight into it as well:
// 64 lanes x 4 bytes = 256 contiguous bytes per store instruction, full lines on the host link, written WHILE the kernel runs.
// The step's closing device-to-host copy of the FASTA (5 MB at C3: 90 us on the critical path behind a 44 us kernel) is gone;
// bytes a thread used to store one by one (4 instructions touching the same 256 bytes) now cost one LDS write each.
// A tile whose bytes do not fit the stage (insertions of thousands of bases) takes the byte-by-byte path, both destinations.
#define KD_CNS_STAGE 2048u      // bytes of LDS staging per workgroup: a tile is 1024 sites + its insertion texts
__global__ void __launch_bounds__(KD_BLOCK)
k_cns_emit(KdTabs T, KdCns C, KdIns ins, kd_u64 tile_first, const kd_u64 *tile_sum, const kd_u64 *tile_off, const KdTileMM *tile_mm,
           uint32_t *depth_minmax, uint8_t *out, uint8_t *changes, kd_u64 *contig_off, uint32_t n_contigs, kd_u64 *patch_off,
           uint8_t *host_out, kd_u64 host_cap) {
    __shared__ kd_u64 s_wave[KD_WAVES_PER_BLOCK];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[KD_CNS_STAGE];
    const uint32_t t = threadIdx.x;
    const kd_u64 tile0 = (tile_first + blockIdx.x) * KD_CNS_TILE;
    const kd_u64 g0 = tile0 + (kd_u64)t * KD_CNS_PER_THREAD;
    KdSite s[KD_CNS_PER_THREAD];
    kd_cns_load_eval(T, C, ins, g0, s);
    kd_u64 base;
    if (tile_off) base = tile_off[blockIdx.x];
    else {
        kd_u64 part = 0;
        for (kd_u64 b = t; b < blockIdx.x; b += KD_BLOCK) part += tile_sum[b];
        kd_u64 tot;
        (void)kd_block_scan_incl(part, s_wave, tot);
        base = tot;
        if (blockIdx.x == 0) kd_cns_fold_mm(tile_mm, gridDim.x, depth_minmax, t, KD_BLOCK);
    }
    uint32_t sum = 0;
    for (int k = 0; k < KD_CNS_PER_THREAD; k++) sum += s[k].ins_len + s[k].has_base;
    kd_u64 tile_total;
    const kd_u64 incl = kd_block_scan_incl((kd_u64)sum, s_wave, tile_total);   // __shfl_up scans, two barriers
    if (blockIdx.x == gridDim.x - 1 && t == 0) contig_off[n_contigs] = base + tile_total;   // total length, next to the per-contig offsets
    const bool staged = tile_total <= KD_CNS_STAGE;      // (uniform over the workgroup)
    kd_u64 o = base + incl - sum;
    const char lower[17] = "=acmgrsvtwyhkdbn";
    // one output byte: into the stage (tile-relative) or, for a tile too long for it, to its final places
#define KD_CNS_PUT(off, v)                                                                         \
    {                                                                                              \
        const kd_u64 off_ = (off); const uint8_t v_ = (uint8_t)(v);                                \
        if (staged) s_stage[(uint32_t)(off_ - base)] = v_;                                         \
        else { out[off_] = v_; if (host_out && off_ < host_cap) host_out[off_] = v_; }             \
    }
    // the thread's four change codes as one store (g0 is a multiple of 4, S a multiple of 1024, the array 4-byte aligned)
    if (g0 < T.sites)
        *reinterpret_cast<uint32_t *>(changes + g0) = (uint32_t)s[0].change | (uint32_t)s[1].change << 8 | (uint32_t)s[2].change << 16 | (uint32_t)s[3].change << 24;
    for (int k = 0; k < KD_CNS_PER_THREAD; k++) {
        const kd_u64 g = g0 + k;
        if (g >= T.sites) break;
        // contig c starts at G-site contig_base[c]: record the output offset there
        if ((g & 63) == 0) {
            const uint32_t c = C.seg_contig[g >> 6];
            if (T.contig_base[c] == g) contig_off[c] = o;
        }
        for (uint32_t pk = 0; pk < C.n_patches; pk++) if (C.patch_start[pk] == g) patch_off[pk] = o;
        if (s[k].ins == 1) {
            const uint8_t *p = ins.pool + ins.ev_off[s[k].ins_ev];
            for (uint32_t b = 0; b < s[k].ins_len; b++) KD_CNS_PUT(o + b, lower[p[b] & 15])  // .lower(), :421
            o += s[k].ins_len;
        } else if (s[k].ins == 2) {
            KD_CNS_PUT(o, 'N')
            o++;
        }
        if (s[k].has_base) { KD_CNS_PUT(o, s[k].base) o++; }
    }
#undef KD_CNS_PUT
    if (!staged) return;      // (uniform)
    __syncthreads();
    // the stage -> [base, base + tile_total) of both destinations: bytes up to the first 4-byte boundary of the DESTINATION and behind the
    // last one singly, whole dwords in between (a dword of the stage sits at any byte offset: four LDS byte reads)
    const uint32_t n = (uint32_t)tile_total;
    const uint32_t head = (uint32_t)((4u - (uint32_t)(base & 3u)) & 3u) < n ? (uint32_t)((4u - (uint32_t)(base & 3u)) & 3u) : n;
    const uint32_t n_dw = (n - head) >> 2, tail0 = head + 4u * n_dw;
    if (t < head) { out[base + t] = s_stage[t]; if (host_out && base + t < host_cap) host_out[base + t] = s_stage[t]; }
    if (t < n - tail0) { const kd_u64 a = base + tail0 + t; out[a] = s_stage[tail0 + t]; if (host_out && a < host_cap) host_out[a] = s_stage[tail0 + t]; }
    for (uint32_t j = t; j < n_dw; j += KD_BLOCK) {
        const uint32_t so = head + 4u * j;
        const uint32_t v = (uint32_t)s_stage[so] | (uint32_t)s_stage[so + 1] << 8 | (uint32_t)s_stage[so + 2] << 16 | (uint32_t)s_stage[so + 3] << 24;
        const kd_u64 a = base + so;
        *reinterpret_cast<uint32_t *>(out + a) = v;
        if (host_out && a + 4 <= host_cap) *reinterpret_cast<uint32_t *>(host_out + a) = v;
        else if (host_out) for (uint32_t b = 0; b < 4 && a + b < host_cap; b++) host_out[a + b] = (uint8_t)(v >> (8 * b));
    }
    (void)n_contigs;
}

// k_exchange_head: header + metadata of a shard's EXCHANGE ROW (kd_engine.h: exchange_queue; include/kindel_hip.h: kd_exchange_row)
//   row[0] = row bytes, row[1] = 0, then contig_off u64[n_contigs + 1] | depth min / max u32[2 n_contigs]
// from the run's metadata block as the device holds it, with what consensus_collect does to the host's copy: a contig whose first
// site lies outside the processed tiles [t_lo, t_hi) was not visited by k_cns_emit -- it has no bytes (offset 0 in front of the
// tiles, the total behind them).  One workgroup, queued behind k_cns_emit: the row is complete without a host round trip of its own.
// A row too small for its metadata gets the 16-byte header only (row bytes > cap says so).
__global__ void __launch_bounds__(KD_BLOCK)
k_exchange_head(const kd_u64 *contig_off, const uint32_t *minmax, const kd_u64 *contig_base, uint32_t n_contigs, kd_u64 t_lo, kd_u64 t_hi,
                kd_u64 fixed, kd_u64 cap, kd_u64 *row) {
    const kd_u64 total = contig_off[n_contigs];
    if (threadIdx.x == 0) { row[0] = fixed + total; row[1] = 0; }
    if (16ULL + ((kd_u64)n_contigs + 1) * 8 + (kd_u64)n_contigs * 8 > cap) return;
    for (uint32_t c = threadIdx.x; c <= n_contigs; c += KD_BLOCK) {
        kd_u64 v = contig_off[c];
        if (c < n_contigs) {
            const kd_u64 b = contig_base[c];
            if (b < t_lo) v = 0;
            else if (b >= t_hi) v = total;
        }
        row[2 + c] = v;
    }
    const kd_u64 *mm = reinterpret_cast<const kd_u64 *>(minmax);     // (the block is u64 contig_off[n + 1] | u32 minmax[2 n]: 8-byte aligned)
    for (uint32_t c = threadIdx.x; c < n_contigs; c += KD_BLOCK) row[2 + n_contigs + 1 + c] = mm[c];
}
