// scripts/exp/kd_cns_one.h -- EXPERIMENT RECORD (round 5), not part of the product: the consensus in ONE pass with a decoupled
// look-back over per-tile states (replacing k_cns_count + k_cns_emit of kindel_amd/csrc/kd_cns.h).  Bit-exact (the whole -m gpu
// suite passed with it: gpurun_out/r05_pytest_gpu_6.log), but SLOWER: C3 (4 883 tiles) k_cns_one 0.474 ms against 0.027 + 0.042 ms
// for the two passes (step 1.98 against 1.53 ms), C2 0.0159 ms, step 0.249 against 0.241 ms (profiles/r05_cns_single_pass_nogo.json).
// Why: the tiles of a consensus run all take the same time, so the ~2 000 resident workgroups publish their own counts at the same
// moment and then wait in groups of 64 for the group in front to know its prefix -- 76 dependent hops of a device-scope store ->
// load round trip each at full size; the look-back pays when tiles finish at staggered times, which these do not.  The two passes
// cost one more launch and read the tables twice (0.13 GB), and are 7 x faster.
// (needs: KDS_CNS_TICKET zeroed by k_ins_flag, kd_st_release / kd_ld_relaxed32 next to kd_ld_acquire, an epoch per run.)
// ONE pass (round 5; rounds 1 - 4: count, [scan,] emit -- the tables read twice, two or three dependent launches): a workgroup
// takes a 1024-site tile from a ticket counter, evaluates its sites once, scans their byte counts inside the tile, learns the
// bytes in front of the tile by a DECOUPLED LOOK-BACK over the tiles before it, and writes bytes / change codes / per-contig
// start offsets.  Tile state word: [epoch:14 | kind:2 | value:48] -- kind 1: the tile's own byte count, kind 2: the byte count
// of all tiles up to and including it.  A tile publishes its own count as soon as it has it, then wavefront 0 reads the 64
// states in front of it at once: the nearest tile that already knows its prefix ends the search, tiles between that one and
// this one contribute their own counts (still counting themselves: wait); nothing known among 64 ready ones: the next 64.
// Tickets are handed out in dispatch order, so a tile only ever waits for workgroups that are resident or done.  The epoch (a
// host counter, per consensus run) tells this run's states from the last run's: no memset between runs.
// The per-contig depth range: a tile folds its own sites and touches the contig's two words only when it IMPROVES them -- a
// relaxed device-scope load first (min / max only move one way: a stale value costs a redundant atomic, never a result).
// Rounds 2 - 4 avoided 4 883 x 2 same-address atomics (0.12 ms on C3) with a per-tile record folded by a scan; of 4 883 tiles
// a few dozen improve a running minimum or maximum.
#define KD_CNS_ST_OWN 1ULL
#define KD_CNS_ST_PFX 2ULL
#define KD_CNS_ST_VAL ((1ULL << 48) - 1ULL)
#define KD_CNS_EPOCHS 0x3fffu
__device__ __forceinline__ kd_u64 kd_cns_state(uint32_t epoch, kd_u64 kind, kd_u64 v) { return (kd_u64)epoch << 50 | kind << 48 | (v & KD_CNS_ST_VAL); }

__global__ void __launch_bounds__(KD_BLOCK)
k_cns_one(KdTabs T, KdCns C, KdIns ins, kd_u64 tile_first, kd_u64 *tile_state, uint32_t epoch, kd_u64 *status,
          uint32_t *depth_minmax, uint8_t *out, uint8_t *changes, kd_u64 *contig_off, uint32_t n_contigs, kd_u64 *patch_off) {
    __shared__ kd_u64 s_wave[KD_WAVES_PER_BLOCK];
    __shared__ uint32_t s_w[2][KD_WAVES_PER_BLOCK];
    __shared__ kd_u64 s_base;
    __shared__ uint32_t s_tile;
    const uint32_t t = threadIdx.x, lane = t & (KD_WAVE - 1), wave = t / KD_WAVE;
    if (t == 0) s_tile = (uint32_t)atomicAdd(&status[KDS_CNS_TICKET], 1ULL);
    __syncthreads();
    const uint32_t tile = s_tile;
    const kd_u64 tile0 = (tile_first + tile) * KD_CNS_TILE;
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
    mn = kd_wave_min(mn); mx = kd_wave_max(mx);
    if (lane == 0) { s_w[0][wave] = mn; s_w[1][wave] = mx; }
    kd_u64 tile_total;
    const kd_u64 incl = kd_block_scan_incl((kd_u64)sum, s_wave, tile_total);   // (two barriers: s_w is written)
    if (wave == 0) {
        if (lane == 0) {     // the tile's own count first: tiles behind this one can go on with it
            kd_st_release(&tile_state[tile], kd_cns_state(epoch, tile ? KD_CNS_ST_OWN : KD_CNS_ST_PFX, tile_total));
            uint32_t a = 0xffffffffu, b = 0;
            for (uint32_t w = 0; w < KD_WAVES_PER_BLOCK; w++) { a = s_w[0][w] < a ? s_w[0][w] : a; b = s_w[1][w] > b ? s_w[1][w] : b; }
            if (a != 0xffffffffu) {      // (live sites of the tile's first contig)
                if (a < kd_ld_relaxed32(&depth_minmax[2 * cfirst])) atomicMin(&depth_minmax[2 * cfirst], a);
                if (b > kd_ld_relaxed32(&depth_minmax[2 * cfirst + 1])) atomicMax(&depth_minmax[2 * cfirst + 1], b);
            }
        }
        kd_u64 excl = 0;
        if (tile) {
            int64_t pos = (int64_t)tile - 1;     // lane l looks at tile pos - l
            for (;;) {
                const int64_t j = pos - (int64_t)lane;
                kd_u64 st = kd_cns_state(epoch, KD_CNS_ST_PFX, 0);       // (in front of tile 0: nothing, known)
                if (j >= 0) st = kd_ld_acquire(&tile_state[j]);
                const bool ready = (uint32_t)(st >> 50) == epoch;
                const unsigned long long m_ready = kd_ballot(ready), m_pfx = kd_ballot(ready && ((st >> 48) & 3ULL) == KD_CNS_ST_PFX);
                const uint32_t first = m_pfx ? (uint32_t)__builtin_ctzll(m_pfx) : KD_WAVE;      // nearest tile that knows its prefix
                const unsigned long long want = first >= KD_WAVE - 1u ? ~0ULL : (2ULL << first) - 1ULL;      // lanes 0 .. first (all 64 when none knows)
                if ((m_ready & want) != want) { kd_spin_pause(); continue; }
                kd_u64 v = (lane <= first || first == KD_WAVE) ? (st & KD_CNS_ST_VAL) : 0ULL;
#pragma unroll
                for (uint32_t m = 1; m < KD_WAVE; m <<= 1) v += kd_shfl64(v, lane ^ m);
                excl += v;
                if (first != KD_WAVE) break;
                pos -= KD_WAVE;
            }
            if (lane == 0)
                kd_st_release(&tile_state[tile], kd_cns_state(epoch, KD_CNS_ST_PFX, excl + tile_total));
        }
        if (lane == 0) s_base = excl;
    }
    __syncthreads();
    const kd_u64 base = s_base;
    if (tile + 1u == gridDim.x && t == 0) contig_off[n_contigs] = base + tile_total;   // total length, next to the per-contig offsets
    kd_u64 o = base + incl - sum;
    const char lower[17] = "=acmgrsvtwyhkdbn";
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
            for (uint32_t b = 0; b < s[k].ins_len; b++) out[o + b] = (uint8_t)lower[p[b] & 15];  // .lower(), :421
            o += s[k].ins_len;
        } else if (s[k].ins == 2) {
            out[o++] = 'N';
        }
        if (s[k].has_base) out[o++] = s[k].base;
    }
}

