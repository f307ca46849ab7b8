// kd_engine.h -- host-side orchestration of the kernels in kd_kernels.h.
//
// KdEngine<Rt> owns the device tables and sequences the launches for one GPU / one stream.
// `Rt` is the runtime policy: kindel_hip.hip instantiates it with the HIP runtime (the
// product); tests/emu/emu_lib.cpp instantiates it with the CPU kernel emulator (test
// infrastructure for kernel logic, never shipped).
//
// Reference correspondence: push_batch = the record loop of parse_records
// (/root/reference/kindel/kindel.py:40-81), finalize = the insertions dicts (:38,55-58) +
// consensus(insertions[pos]) (:420), consensus_run = consensus_sequence (:384-430).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/kindel_hip.h"
#include "kd_kernels.h"

template <class Rt>
struct KdEngine {
    Rt rt;
    std::string err;
    uint32_t n_contigs = 0;
    std::vector<uint32_t> clen;
    std::vector<uint64_t> cbase;
    uint64_t S = 0;  // G-space sites, multiple of 1024 (consensus tile)
    uint64_t g_lo = 0, g_hi = 0;  // emit interval [g_lo, g_hi)
    int mode = KD_MODE_AUTO;
    uint32_t W = 0, slice_cfg = 0;     // sites per LDS window; 0 = the kernel's own default (kd_set_tuning overrides both kernels)
    static constexpr uint32_t W_LANE = 448;   // k_window (lane per read): 19 ch x (448 + 192 reach + 16 halo) x 2 B = 25 KB of LDS histogram per workgroup
    static constexpr uint32_t W_COOP = 640;   // k_window_coop (16 lanes per read): 33 rows x (320 + 32) dwords = 46 KB + 29 KB of staged tile, two workgroups per CU
    bool coop_mode() const { return mode == KD_MODE_COOP; }   // (measured slower than k_window: DESIGN.md section 3; never the default)
    uint32_t window_sites(bool coop) const { return W ? W : coop ? W_COOP : W_LANE; }

    uint32_t *d_tab = nullptr, *d_clen = nullptr, *d_seg = nullptr;
    // the tables are allocated for the shard only: sites [alloc_lo, alloc_hi) (tile aligned) + slack, `pitch` dwords per
    // channel.  Lazily (first push / first read-out), so that kd_create + kd_set_shard never allocate the whole G-space.
    uint64_t alloc_lo = 0, alloc_hi = 0, pitch = 0;
    bool batch_status_clean = false;   // the per-batch status words are all zero (k_reset ran, no batch since)
    bool tables_ready = false;
    kd_u64 *d_cbase = nullptr, *d_status = nullptr;
    kd_u64 *d_first_idx = nullptr, *d_err_first = nullptr;   // per contig: first record / first failing read (global indices)
    uint32_t *d_err_code = nullptr;
    std::vector<kd_u64> h_status = std::vector<kd_u64>(KDS_COUNT, 0);

    // grow-only device buffers
    struct Buf {
        void *p = nullptr;
        size_t cap = 0;
    };
    Buf b_longorder;      // k_long_order: the long reads longest first
    bool long_ordered = false;
    Buf b_rinfo, b_cold, b_irreg, b_long, b_winlo, b_winhi, b_itemoff, b_itemwin, b_readev, b_readpool, b_order, b_bincnt, b_binoff, b_rows, b_rowinfo, b_rowoff, b_longacc, b_coldcnt, b_coldev, b_coldpool;
    Buf b_stage[9];
    Buf b_gi_file, b_gi_blocks, b_gi_out, b_gi_bstat, b_gi_start, b_gi_cnt, b_gi_tot, b_gi_recat;   // device-side ingest (kd_ingest.h)
    Buf b_gi_tok, b_gi_ntok, b_gi_work;   // the two-pass inflate's token lists, their lengths, and the groups' work counters (kd_gpu_inflate2.h)
    static constexpr uint32_t GI_MAX_GROUPS = 64;
    Buf b_sortrows, b_sortseg;   // unsorted input: per-workgroup bin counts / starts and their segment totals (kd_plan.h)
    Buf b_smallcig;   // a batch with fewer than 4 CIGAR words: its padded copy
    Buf b_srec;   // an unsorted batch's regular reads in window order: KdSortRec[] (k_sort_scatter_reads)
    Buf b_ev_site, b_ev_len, b_ev_off, b_pool;
    uint64_t ev_cap = 0, pool_cap = 0;
    Buf b_hkey, b_hcnt, b_hrep, b_evslot, b_best, b_best2, b_flag;   // b_best / b_best2: the per-site maxima best_a / best_b (kd_ins.h); b_flag: k_ins_flag's site bytes
    bool ins_site_flags = false;      // this reduction's site test was made by k_ins_flag (else per event by k_ins_insert)
    Buf b_bound, b_hot;   // k_window's work queue: k_prep's boundary table, the hot-window list (kd_window.h: KdWq)
    Buf b_patch;          // consensus_run with CDR patches: patch_off | patch_start | patch_end
    uint64_t hash_cap = 0;
    // what the last insertion reduction left behind (k_ins_cleanup undoes it before the event buffers are reused)
    uint64_t ins_dirty_ev = 0, ins_dirty_bias = 0;   // (bias: alloc_lo when the reduction ran -- best_a[] / best_b[] are shard-local)
    KdInsTab ins_dirty_tab;
    Buf b_cns, b_changes, b_tilesum, b_tilemm, b_tileoff;

    uint64_t reads_pushed = 0;
    uint64_t last_windowed = 0, last_nwin = 0;   // (kd_batch_info: the last batch took the window path; windows it planned)
    bool finalized = false, have_cns = false;
    uint64_t n_ev_final = 0, pool_final = 0;
    // host copies of the last consensus run
    std::vector<kd_u64> h_coff;
    std::vector<uint32_t> h_minmax;
    std::vector<kd_u64> h_pstart, h_poff;
    // host copy of the insertion table (kd_get_insertions)
    struct InsKey { uint32_t site, count, len; uint64_t off; uint32_t rep; };
    std::vector<InsKey> h_inskeys;
    std::vector<uint8_t> h_insbytes;
    bool have_inskeys = false;

    int fail(int code, const std::string &m) { err = m; return code; }
    int hipfail(const char *what) { return fail(KD_E_HIP, std::string(what) + ": " + rt.err()); }

    // (the macro below passes the buffer's name on: what KD_GUARD's allocation table and fault report call it)
    int ensure_(const char *tag, Buf &b, size_t bytes, bool keep = false, size_t keep_bytes = 0) {
        if (bytes <= b.cap) return KD_OK;
        size_t ncap = std::max(bytes, b.cap + b.cap / 2);
        ncap = (ncap + 255) & ~size_t(255);
        if (rt.exact_sizes()) ncap = (bytes + 15) & ~size_t(15);      // KD_GUARD: no head-room -- an access past what was asked for must fault
        void *np = rt.alloc(ncap, tag);
        if (!np) return fail(KD_E_NOMEM, "device allocation of " + std::to_string(ncap) + " bytes failed: " + rt.err());
        if (keep && b.p && keep_bytes) {
            if (rt.d2d(np, b.p, keep_bytes)) return hipfail("d2d");
            if (rt.sync()) return hipfail("sync");
        }
        if (b.p) rt.free(b.p);
        b.p = np; b.cap = ncap;
        return KD_OK;
    }
#define ensure(b, ...) ensure_(#b, b, __VA_ARGS__)
    void release(Buf &b) { if (b.p) rt.free(b.p); b.p = nullptr; b.cap = 0; }

    KdTabs tabs() const {
        KdTabs T;
        T.tab = d_tab - alloc_lo; T.stride = pitch; T.sites = S; T.contig_len = d_clen; T.contig_base = d_cbase;
        T.g_lo = g_lo; T.g_hi = g_hi;  // commit includes the halo site g_hi
        T.first_idx = d_first_idx; T.err_first = d_err_first; T.err_code = d_err_code;
        return T;
    }
    KdIns insdesc() const {
        KdIns I;
        I.ev_site = (uint32_t *)b_ev_site.p; I.ev_len = (uint32_t *)b_ev_len.p; I.ev_off = (kd_u64 *)b_ev_off.p;
        I.pool = (uint8_t *)b_pool.p; I.ev_cap = ev_cap; I.pool_cap = pool_cap;
        I.read_ev = (uint32_t *)b_readev.p; I.read_pool = (kd_u64 *)b_readpool.p;
        return I;
    }

    int create(int device, uint32_t n, const uint32_t *lens, void *stream) {
        if (!n || !lens) return fail(KD_E_ARG, "kd_create: no contigs");
        if (rt.init(device, stream)) return hipfail("kd_create: device init");
        if (const char *e = getenv("KD_COLD_TAIL")) knob_cold_tail = atoi(e) != 0;
        if (const char *e = getenv("KD_INS_SITE_FLAGS")) knob_ins_site_flags = atoi(e) != 0;
        if (const char *e = getenv("KD_ZERO_COPY")) knob_zero_copy = atoi(e) != 0;
        if (const char *e = getenv("KD_INFLATE")) knob_inflate = atoi(e) == 2 ? 2 : 1;
        n_contigs = n;
        clen.assign(lens, lens + n);
        cbase.resize(n);
        uint64_t g = 0;
        for (uint32_t c = 0; c < n; c++) {
            cbase[c] = g;
            g += ((uint64_t)lens[c] + 1 + 63) & ~uint64_t(63);  // len+1 slots (kindel.py:36-39), padded to 64
        }
        S = (g + KD_CNS_TILE - 1) / KD_CNS_TILE * KD_CNS_TILE;
        if (S >= 0xfffffff0ULL) return fail(KD_E_ARG, "kd_create: more than 2^32 reference sites");
        g_lo = 0; g_hi = S;
        d_clen = (uint32_t *)rt.alloc((size_t)n * 4, "d_clen");
        d_cbase = (kd_u64 *)rt.alloc((size_t)n * 8, "d_cbase");
        d_seg = (uint32_t *)rt.alloc((size_t)(S / 64) * 4, "d_seg");
        // the status words and, right behind them, the consensus run's metadata block (per-contig output offsets and depth ranges,
        // meta_coff() / meta_mm()): what a step hands back to the host is ONE copy (round 5; it was two, 6 us apart)
        d_status = (kd_u64 *)rt.alloc(KDS_COUNT * 8 + meta_bytes(), "d_status");
        d_first_idx = (kd_u64 *)rt.alloc((size_t)n * 8, "d_first_idx");
        d_err_first = (kd_u64 *)rt.alloc((size_t)n * 8, "d_err_first");
        d_err_code = (uint32_t *)rt.alloc((size_t)n * 4, "d_err_code");
        if (!d_clen || !d_cbase || !d_seg || !d_status || !d_first_idx || !d_err_first || !d_err_code)
            return fail(KD_E_NOMEM, std::string("kd_create: device allocation failed: ") + rt.err());
        std::vector<uint32_t> seg(S / 64, n - 1);
        for (uint32_t c = 0; c < n; c++) {
            uint64_t e = c + 1 < n ? cbase[c + 1] : S;
            for (uint64_t s = cbase[c] / 64; s < e / 64; s++) seg[s] = c;
        }
        if (rt.h2d(d_clen, clen.data(), (size_t)n * 4) || rt.h2d(d_cbase, cbase.data(), (size_t)n * 8) ||
            rt.h2d(d_seg, seg.data(), seg.size() * 4))
            return hipfail("kd_create: upload");
        if (rt.sync()) return hipfail("kd_create: sync");
        return reset();
    }

    void destroy() {
        Buf *all[] = {&b_rinfo, &b_cold, &b_irreg, &b_long, &b_winlo, &b_winhi, &b_itemoff, &b_itemwin, &b_readev, &b_readpool, &b_order, &b_bincnt, &b_binoff, &b_rows, &b_rowinfo, &b_rowoff, &b_longacc, &b_coldcnt, &b_coldev, &b_coldpool, &b_ev_site, &b_ev_len,
                      &b_ev_off, &b_pool, &b_hkey, &b_hcnt, &b_hrep, &b_evslot, &b_best, &b_best2, &b_flag, &b_bound, &b_hot, &b_patch, &b_cns, &b_changes,
                      &b_tilesum, &b_tilemm, &b_tileoff};
        for (Buf *b : all) release(*b);
        for (Buf &b : b_stage) release(b);
        for (Buf &b : b_gin) release(b);
        release(b_srec); release(b_smallcig); release(b_sortrows); release(b_sortseg); release(b_longorder);
        for (Buf *g : {&b_gi_file, &b_gi_blocks, &b_gi_out, &b_gi_bstat, &b_gi_start, &b_gi_cnt, &b_gi_tot, &b_gi_recat, &b_gi_tok, &b_gi_ntok, &b_gi_work}) release(*g);
        if (d_tab) rt.free(d_tab);
        if (d_clen) rt.free(d_clen);
        if (d_cbase) rt.free(d_cbase);
        if (d_seg) rt.free(d_seg);
        if (d_status) rt.free(d_status);
        if (d_first_idx) rt.free(d_first_idx);
        if (d_err_first) rt.free(d_err_first);
        if (d_err_code) rt.free(d_err_code);
        d_tab = d_clen = d_seg = d_err_code = nullptr; d_cbase = d_status = d_first_idx = d_err_first = nullptr;
        rt.shutdown();
    }

    // 64-site aligned cover of the shard's commit range [g_lo, g_hi] (g_hi = halo site)
    void shard_cover(uint64_t &lo, uint64_t &hi) const {
        lo = g_lo & ~uint64_t(63);
        hi = std::min<uint64_t>(S, (g_hi + 1 + 63) & ~uint64_t(63));
    }

    // (re)allocate the tables for the current shard and zero them
    int prepare_tables() {
        const uint64_t lo = g_lo / KD_CNS_TILE * KD_CNS_TILE;
        const uint64_t hi = std::min<uint64_t>(S, (g_hi + 1 + KD_CNS_TILE - 1) / KD_CNS_TILE * KD_CNS_TILE);
        if (!d_tab || lo != alloc_lo || hi != alloc_hi) {
            if (d_tab) { if (rt.sync()) return hipfail("sync"); rt.free(d_tab); d_tab = nullptr; }
            alloc_lo = lo; alloc_hi = hi; pitch = (hi - lo) + 64;   // slack: the consensus reads one site past its last tile
            d_tab = (uint32_t *)rt.alloc((size_t)KDC_NCH * pitch * 4, "d_tab");
            if (!d_tab)
                return fail(KD_E_NOMEM, "device allocation of the tables failed (" + std::to_string((size_t)KDC_NCH * pitch * 4) + " bytes): " + rt.err());
        }
        if (rt.memset(d_tab, 0, (size_t)KDC_NCH * pitch * 4)) return hipfail("memset tables");
        tables_ready = true;
        return KD_OK;
    }

    int reset() {
        tables_ready = false;   // zeroed (and, after kd_set_shard, re-allocated) by the next push or read-out
        std::fill(h_status.begin(), h_status.end(), 0);
        h_status[KDS_ERR_READ] = ~0ULL;
        // status words and the per-contig first-record / first-error state in ONE launch (it was a copy and three memsets) --
        // the launch that undoes the last insertion reduction (before its event buffers are overwritten), when there was one
        const uint64_t n_reset = std::max<uint64_t>(n_contigs, KDS_COUNT);
        if (ins_dirty_ev) {
            KdIns I = insdesc();
            const uint64_t grid = std::max<uint64_t>((ins_dirty_ev + KD_INS_CHUNK - 1) / KD_INS_CHUNK, (n_reset + KD_BLOCK - 1) / KD_BLOCK);
            if (rt.launch("k_ins_cleanup", k_ins_cleanup, (unsigned)grid, KD_BLOCK, 0, I, ins_dirty_tab,
                          (kd_u64)ins_dirty_ev, (kd_u64 *)b_best.p - ins_dirty_bias, (kd_u64 *)b_best2.p - ins_dirty_bias,
                          d_status, d_first_idx, d_err_first, d_err_code, n_contigs))
                return hipfail("k_ins_cleanup");
            ins_dirty_ev = 0;
        } else if (rt.launch("k_reset", k_reset, (unsigned)((n_reset + KD_BLOCK - 1) / KD_BLOCK), KD_BLOCK, 0,
                             d_status, d_first_idx, d_err_first, d_err_code, n_contigs))
            return hipfail("k_reset");
        batch_status_clean = true;
        reads_pushed = 0; finalized = false; have_cns = false; have_inskeys = false;
        errors_pending = false;
        return KD_OK;
    }

    int set_shard(uint64_t lo, uint64_t hi) {
        if (lo > hi || hi > S) return fail(KD_E_ARG, "kd_set_shard: bad interval");
        if (reads_pushed) return fail(KD_E_ARG, "kd_set_shard: call before the first batch (or after kd_reset)");
        g_lo = lo; g_hi = hi;
        return reset();
    }

    // hash table, best_a[] and best_b[] back to all-zero: one thread per four events of the last reduction
    int ins_cleanup() {
        if (!ins_dirty_ev) return KD_OK;
        KdIns I = insdesc();
        if (rt.launch("k_ins_cleanup", k_ins_cleanup, (unsigned)((ins_dirty_ev + KD_INS_CHUNK - 1) / KD_INS_CHUNK), KD_BLOCK, 0, I, ins_dirty_tab,
                      (kd_u64)ins_dirty_ev, (kd_u64 *)b_best.p - ins_dirty_bias, (kd_u64 *)b_best2.p - ins_dirty_bias,
                      (kd_u64 *)nullptr, (kd_u64 *)nullptr, (kd_u64 *)nullptr, (uint32_t *)nullptr, 0u))
            return hipfail("k_ins_cleanup");
        ins_dirty_ev = 0;
        return KD_OK;
    }

    int fetch_status() {
        if (rt.d2h_small(h_status.data(), d_status, KDS_COUNT * 8)) return hipfail("status d2h");
        return KD_OK;
    }

    // ---- pileup ----
    Buf b_gin[9];      // KD_GUARD: the caller's device batch, copied into fenced buffers of exactly the promised sizes
    int push_device(const kd_batch &B_in) {
        kd_batch B = B_in;
        const uint64_t n = B.n_reads;
        if (!defer_errors) errors_pending = false;      // (a kd_step that failed half-way leaves nothing for a later kd_finish to classify)
        if (!n) return KD_OK;
        if (rt.exact_sizes()) {
            // include/kindel_hip.h: n entries per read array, cigar_words words, seq4_bytes + 16 readable bytes of packed bases --
            // a kernel that reads more than that faults on the copy's fence
            const void *src[9] = {B.contig, B.pos0, B.flag, B.seq_off, B.seq_len, B.cig_off, B.n_cig, B.seq4, B.cigar};
            const size_t bytes[9] = {n * 4, n * 4, n * 4, n * 8, n * 4, n * 8, n * 4, (size_t)B.seq4_bytes + 16, (size_t)B.cigar_words * 4};
            if (rt.sync()) return hipfail("push: sync");      // (an earlier batch's kernels may still read the last copies)
            for (int k = 0; k < 9; k++) {
                release(b_gin[k]);                          // exact size every time: a smaller batch must not inherit a larger one's room
                int rcg = ensure(b_gin[k], std::max<size_t>(bytes[k], 16));
                if (rcg) return rcg;
                const size_t have = k == 7 ? (size_t)B.seq4_bytes : bytes[k];      // (the 16 bytes behind the packed bases: readable, not meaningful)
                if ((k == 7 && rt.memset((uint8_t *)b_gin[k].p + have, 0, 16)) || (have && rt.d2d(b_gin[k].p, src[k], have))) return hipfail("push: d2d");
            }
            B.contig = (const uint32_t *)b_gin[0].p; B.pos0 = (const int32_t *)b_gin[1].p; B.flag = (const uint32_t *)b_gin[2].p;
            B.seq_off = (const uint64_t *)b_gin[3].p; B.seq_len = (const uint32_t *)b_gin[4].p; B.cig_off = (const uint64_t *)b_gin[5].p;
            B.n_cig = (const uint32_t *)b_gin[6].p; B.seq4 = (const uint8_t *)b_gin[7].p; B.cigar = (const uint32_t *)b_gin[8].p;
        }
        if (n >= 0xffffffffULL) return fail(KD_E_ARG, "kd_push_batch: more than 2^32-1 reads in one batch");
        if (reinterpret_cast<uintptr_t>(B.seq4) & 15u) return fail(KD_E_ARG, "kd_push_batch_device: seq4 must be 16-byte aligned");
        int rc;
        // reads per lane of k_prep (one wavefront per workgroup): 64 keep its per-wavefront atomics few (one per counter and 4096
        // reads), but a small batch (a shard of a strong-scaling run, a deep small genome) then launches fewer wavefronts than the
        // chip has slots: halve until ~4 wavefronts per CU are there
        uint32_t prep_per = KD_PREP_PER_THREAD;
        if (const char *e = getenv("KD_PREP_PER")) prep_per = (uint32_t)std::min(KD_PREP_PER_THREAD, std::max(KD_PREP_UNROLL, atoi(e) / KD_PREP_UNROLL * KD_PREP_UNROLL));   // (knob: measurement)
        else {
            // (round 6, scripts/exp/prep_per_sweep.sh on the shards of C3's strong decomposition -- k_prep ms at 8 / 16 / 32 / 64 reads per lane:
            //  16.7 M reads 0.464 / 0.283 / 0.227 / 0.216, 8.3 M 0.254 / 0.166 / 0.136 / 0.175, 4.2 M 0.147 / 0.101 / 0.102 / 0.141, 2.1 M
            //  0.096 / 0.085 / 0.089 / 0.125: a lane's chain of dependent loads is what a launch waits for, so down to 16 reads per lane
            //  the launch should hold ~14 wavefronts per CU; below that only a batch too small for 4 per CU goes on halving)
            while (prep_per > 16u && n / ((uint64_t)KD_PREP_BLOCK * prep_per) < (uint64_t)14 * rt.n_cus()) prep_per /= 2;
            while (prep_per > KD_PREP_UNROLL && n / ((uint64_t)KD_PREP_BLOCK * prep_per) < (uint64_t)4 * rt.n_cus()) prep_per /= 2;
        }
        const uint32_t prep_chunk = KD_PREP_BLOCK * prep_per, cold_region = KD_WAVE * prep_per;
        const unsigned prep_grid = (unsigned)((n + prep_chunk - 1) / prep_chunk);
        const unsigned prep_regions = prep_grid;     // one region of compact cold-read records per wavefront of k_prep
        if ((rc = ins_cleanup())) return rc;      // the last reduction's events are about to be joined by new ones
        if ((rc = ensure(b_rinfo, n * sizeof(KdRInfo))) || (rc = ensure(b_cold, (size_t)prep_regions * cold_region * sizeof(KdColdRec))) || (rc = ensure(b_coldcnt, (size_t)prep_regions * 4)) ||
            (rc = ensure(b_coldev, (size_t)prep_regions * 8)) || (rc = ensure(b_coldpool, (size_t)prep_regions * 8)) ||
            (rc = ensure(b_irreg, n * 4)) || (rc = ensure(b_long, n * 4)) || (rc = ensure(b_readev, n * 4)) ||
            (rc = ensure(b_readpool, n * 8)))
            return rc;
        KdReads R;
        R.n = n; R.base_index = reads_pushed;
        R.contig = B.contig; R.pos0 = B.pos0; R.flag = B.flag; R.seq_off = (const kd_u64 *)B.seq_off;
        R.seq_len = B.seq_len; R.cig_off = (const kd_u64 *)B.cig_off; R.n_cig = B.n_cig; R.seq4 = B.seq4;
        R.cigar = B.cigar; R.n_cigar = B.cigar_words; R.osh = 0;
        if (R.n_cigar < 4) {   // k_prep loads a read's first four CIGAR words in one go: a batch with fewer gets a padded copy
            if ((rc = ensure(b_smallcig, 16)) || rt.memset(b_smallcig.p, 0, 16) || rt.d2d(b_smallcig.p, B.cigar, (size_t)B.cigar_words * 4))
                return rc ? rc : hipfail("push: CIGAR copy");
            R.cigar = (const uint32_t *)b_smallcig.p; R.n_cigar = 4;
        }
        KdTabs T = tabs();
        KdRInfo *rinfo = (KdRInfo *)b_rinfo.p;
        KdColdRec *cold = (KdColdRec *)b_cold.p; uint32_t *irreg = (uint32_t *)b_irreg.p, *lng = (uint32_t *)b_long.p;
        // per-batch status words are contiguous: KDS_B_INS_OPS .. KDS_TOTAL_ITEMS
        if (!batch_status_clean && rt.memset(d_status + KDS_B_INS_OPS, 0, (size_t)(KDS_TOTAL_ITEMS - KDS_B_INS_OPS + 1) * 8))
            return hipfail("push: memset status");      // (the first batch after kd_reset finds them zeroed by k_reset)
        batch_status_clean = false;
        const uint64_t ev_before = h_status[KDS_N_EV], pool_before = h_status[KDS_POOL];  // as of the last fetch
        // k_window's boundary table (first read at or behind every 64th site of G-space; kd_window.h: KdWq) is written by k_prep
        const bool self_planned = mode != KD_MODE_GLOBAL && mode != KD_MODE_STRIP;
        const uint32_t nb = (uint32_t)(S / 64);
        if (self_planned && (rc = ensure(b_bound, ((size_t)nb + 1) * 4))) return rc;
        if (rt.launch("k_prep", k_prep, prep_grid, KD_PREP_BLOCK, 0, R, T, rinfo, cold, (uint32_t *)b_coldcnt.p, (kd_u64 *)b_coldev.p, (kd_u64 *)b_coldpool.p, irreg, lng, (uint32_t *)b_readev.p,
                      (kd_u64 *)b_readpool.p, d_status, prep_per, self_planned ? (uint32_t *)b_bound.p : (uint32_t *)nullptr, nb))
            return hipfail("k_prep");
        // k_prep touches no table: the first batch's table zeroing is queued BEHIND it and behind the status copy, so that
        // the host's wait for the copy (a round trip of ~30 us) passes while the memset runs.  (Round 5 measured the zeroing NEXT
        // to k_prep instead, on a side stream joined in front of k_window: C3 1.593 ms against 1.598 with the zeroing in front of
        // k_prep on the main stream and 1.52 - 1.56 this way -- the two kernels share the memory system, nothing is gained: dropped.)
        if (rt.d2h_small_begin(d_status, KDS_COUNT * 8)) return hipfail("status d2h");
        if (!tables_ready) {
            if ((rc = prepare_tables())) return rc;
            T = tabs();      // (the tables may just have been allocated: k_prep only used the contig geometry of T)
        }
        if (rt.d2h_small_end(h_status.data(), KDS_COUNT * 8)) return hipfail("status d2h");
        const uint64_t n_long = h_status[KDS_B_N_LONG];
        const uint64_t n_reg_short = h_status[KDS_B_N_REG];   // (k_long_reduce adds the regular long reads: their rows are the second pass)
        if (n_long) {
            // long-CIGAR reads (kd_long.h): validated one workgroup each, slots / rows handed out, then expanded into rows
            if ((rc = ensure(b_rowinfo, (size_t)n_long * sizeof(KdRInfo))) || (rc = ensure(b_rowoff, (size_t)n_long * 8)) ||
                (rc = ensure(b_longacc, (size_t)n_long * sizeof(KdLongAcc))))
                return rc;
            // k_prep_long and k_long_expand start the longest reads first (k_long_order)
            long_ordered = n_long > 1 && n_long <= KD_LONG_ORDER_MAX;
            if (long_ordered) {
                if ((rc = ensure(b_longorder, (size_t)n_long * 8))) return rc;      // (the order | the reads' length classes)
                if (rt.launch("k_long_order", k_long_order, 1u, KD_LONG_ORDER_BLOCK, 0, R, (const uint32_t *)lng, (uint32_t)n_long, (uint32_t *)b_longorder.p))
                    return hipfail("k_long_order");
            }
            const unsigned long_grid = (unsigned)((n_long + KD_LONG_WAVES - 1) / KD_LONG_WAVES);   // a wavefront per long read, a workgroup per wavefront
            if (rt.launch("k_prep_long", k_prep_long, long_grid, KD_LONG_BLOCK, 0, R, T, rinfo,
                          (const uint32_t *)lng, (uint32_t)n_long, (KdLongAcc *)b_longacc.p, long_ordered ? (const uint32_t *)b_longorder.p : (const uint32_t *)nullptr))
                return hipfail("k_prep_long");
            if (rt.launch("k_long_reduce", k_long_reduce, (unsigned)((n_long + KD_BLOCK - 1) / KD_BLOCK), KD_BLOCK, 0, (const KdLongAcc *)b_longacc.p,
                          (const uint32_t *)lng, (uint32_t)n_long, (const KdRInfo *)rinfo, irreg, (uint32_t *)b_readev.p, (kd_u64 *)b_readpool.p,
                          (KdRInfo *)b_rowinfo.p, (kd_u64 *)b_rowoff.p, d_status))
                return hipfail("k_long_reduce");
            if ((rc = fetch_status())) return rc;
            if ((rc = ensure(b_rows, (size_t)h_status[KDS_B_ROW_DWORDS] * 4 + 64))) return rc;   // (+ 64: the walk loads 16-byte chunks)
        }
        // size the insertion event buffers from the exact counts of this batch
        // k_prep / k_prep_long have already reserved this batch's slots in KDS_N_EV / KDS_POOL
        const uint64_t need_ev = h_status[KDS_N_EV], need_pool = h_status[KDS_POOL];
        if (need_ev > ev_cap) {
            uint64_t ncap = std::max<uint64_t>(need_ev, ev_cap + ev_cap / 2);
            const uint64_t used = ev_before;
            if ((rc = ensure(b_ev_site, ncap * 4, true, used * 4)) || (rc = ensure(b_ev_len, ncap * 4, true, used * 4)) ||
                (rc = ensure(b_ev_off, ncap * 8, true, used * 8)))
                return rc;
            ev_cap = ncap;
        }
        if (need_pool > pool_cap) {
            uint64_t ncap = std::max<uint64_t>(need_pool, pool_cap + pool_cap / 2);
            if ((rc = ensure(b_pool, ncap, true, pool_before))) return rc;
            pool_cap = ncap;
        }
        KdIns I = insdesc();
        if (n_long && mode != KD_MODE_GLOBAL &&
            rt.launch("k_long_expand", k_long_expand, (unsigned)((n_long + KD_LONG_WAVES - 1) / KD_LONG_WAVES), KD_LONG_BLOCK, 0, R, T, I,
                      (const KdRInfo *)rinfo, (const uint32_t *)lng, (uint32_t)n_long, (const KdLongAcc *)b_longacc.p, (const kd_u64 *)b_rowoff.p, (uint8_t *)b_rows.p, d_status,
                      long_ordered ? (const uint32_t *)b_longorder.p : (const uint32_t *)nullptr))
            return hipfail("k_long_expand");
        const uint64_t n_reg = h_status[KDS_B_N_REG], n_cold = h_status[KDS_B_N_COLD], n_irreg = h_status[KDS_B_N_IRREG];
        const bool windowed = (mode != KD_MODE_GLOBAL) && n_reg > 0;
        const bool sorted_input = h_status[KDS_B_UNSORTED] == 0;
        last_windowed = windowed ? 1 : 0; last_nwin = 0;
        if (windowed) {
            // windows intersecting the shard's commit range only; the planning arrays are shared by the passes (first pass by
            // k_window_coop / k_window / k_strip, long-read segments by k_window), each with its own window size
            const uint64_t g_end = std::min<uint64_t>(S, g_hi + 1);
            auto windows_of = [&](uint32_t Wx, uint32_t &first) { first = (uint32_t)(g_lo / Wx); return (uint32_t)((g_end + Wx - 1) / Wx) - first; };
            const bool coop = coop_mode();
            bool cold_fused = false;      // the cold records' workgroups rode in k_window's launch (kd_window.h: KdColdTail)
            const bool cold_tail_on = knob_cold_tail;
            const uint32_t W_first = window_sites(coop), W_seg = window_sites(false);
            uint32_t ws0, dummy0;
            const uint32_t ns_win = windows_of(KD_STRIP, ws0);
            const size_t nw_max = std::max<size_t>(std::max(windows_of(W_first, dummy0), windows_of(W_seg, dummy0)), ns_win);
            if ((rc = ensure(b_winlo, nw_max * 8)) || (rc = ensure(b_winhi, nw_max * 8)) || (rc = ensure(b_itemoff, (nw_max + 1) * 8)))
                return rc;
            kd_u64 *wl = (kd_u64 *)b_winlo.p, *wh = (kd_u64 *)b_winhi.p, *io = (kd_u64 *)b_itemoff.p;
            // One pass over `ne` entries described by `info`: the batch's reads (rows == false; k_window_coop when `use_coop`),
            // then the ROWS of its long reads (k_long_expand; one entry per long read, in the order of the long list: bucket-
            // sorted by window through a permutation).
            auto window_pass = [&](const KdRInfo *info, uint64_t ne, bool in_order, bool rows, uint32_t span_slot,
                                   uint32_t W, bool want_coop) -> int {
                int rc2;
                // 16 lanes per read, row-major histogram (kd_coop.h) -- unless a hand-picked window is too wide for its 33 rows
                const bool use_coop = want_coop && KD_COOP_LDS_BYTES(KD_COOP_PITCH(W)) <= (size_t)160 * 1024 - 1024;
                // the histogram reaches H sites past the window: an entry is tallied whole by the window it starts in (kd_window.h:
                // OWNERSHIP); H = the longest footprint of this pass's entries, up to 256 sites (longer ones leave a remainder)
                // (a row is thousands of sites long: every window tallies its own part of it, H = 0)
                uint32_t H = (use_coop || rows) ? 0u : (uint32_t)std::min<uint64_t>(256, (h_status[span_slot] + 63) & ~uint64_t(63));
                while (H && KD_WINDOW_LDS_BYTES((W + H + 2 * KD_HALO) / 2) > (size_t)160 * 1024 - 1024) H -= 64;   // (a hand-picked window near the LDS limit)
                while (!use_coop && W > 64 && KD_WINDOW_LDS_BYTES((W + H + 2 * KD_HALO) / 2) > (size_t)160 * 1024 - 1024) W -= 64;   // (... or beyond it)
                auto grid_of = [&](uint32_t Wx) {
                    const size_t l = KD_WINDOW_LDS_BYTES((Wx + H + 2 * KD_HALO) / 2);
                    return std::max(1u, (unsigned)rt.n_cus() * (unsigned)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024 - 512) / (l + 64))));
                };
                uint32_t w0;
                // the default window keeps window + reach at 640 sites (448 + 192 for 150-base reads, 384 + 256 for 250-base ones, 512 + 128
                // for 100-base ones): the LDS footprint that lets five workgroups share a CU
                if (!this->W && !use_coop && !rows && H) W = 640u - H;
                if (!this->W && !use_coop && !rows) {
                    // A SMALL shard (1/8 of C3 on one of eight GPUs: 1395 windows of 448 sites on 1280 resident workgroups): a few windows
                    // left over for a second, thin round cost a window's whole latency.  A wider window that puts every window into
                    // the FIRST round is taken when there is one (measured on that shard: k_window 0.21 -> 0.18 ms, step -7 %).
                    uint32_t d0;
                    const uint32_t n0 = windows_of(W, d0), g0 = grid_of(W);
                    if (n0 > g0 && n0 < 2 * g0)
                        for (uint32_t Wx = W + 64; Wx <= W + 256; Wx += 64)
                            if (windows_of(Wx, d0) <= grid_of(Wx)) { W = Wx; break; }
                }
                const uint32_t n_win = windows_of(W, w0);
                const uint32_t Wh = (W + H + 2 * KD_HALO) / 2;   // dwords per channel row
                const size_t win_lds = KD_WINDOW_LDS_BYTES(Wh);
                const unsigned win_grid = grid_of(W);
                uint32_t slice = slice_cfg, static_cut = 0;
                if (!slice && use_coop) slice = (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(256, ne / 4096));   // (planned queue: a few thousand items)
                // A work item = one zeroing + one flush of the window's histogram, whatever it tallies in between: every slice
                // a window is cut into repeats both (measured, 1/8 of C3: 3 - 4 slices per window 1.00 ms, one 0.19 ms).  So: a
                // window is ONE item unless it holds several times the average (a hot spot: cut, so that helpers can share it)
                // -- except when there are several resident workgroups per window (a deep small genome: C2 is 23 windows of
                // 29 000 reads): then every workgroup takes an equal part of its window (the STATIC queue of kd_window.h: parts of
                // 256 reads or more, tallied in pieces of `slice`: the u16 counters' limit, i.e. one piece unless a test asks).
                const uint64_t avg = rows ? 0 : ne / std::max<uint32_t>(n_win, 1u) + 1;     // (rows: an entry is a candidate of many windows; their depth is what matters: 4096)
                const uint64_t cut = win_grid / std::max<uint32_t>(n_win, 1u);     // workgroups per window (rounded down: 1.5 is not worth a second flush)
                if (!rows && !use_coop && cut >= 2) static_cut = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(cut, avg / 256));
                if (!slice) {
                    if (rows) slice = 4096u;
                    else if (cut <= 1) slice = (uint32_t)std::min<uint64_t>(32768, std::max<uint64_t>(1024, 4 * avg));
                    else slice = 32768u;
                }
                slice = std::min<uint32_t>(slice, 32768u);   // u16 LDS counters: an item may not tally more reads than that
                const uint32_t *order = nullptr;
                const KdRInfo *walk_info = info;      // what k_window walks: the entries themselves, or their window-sorted copy
                KdReads walk_R = R;
                KdWq Q;       // k_window plans for itself (kd_window.h): the boundary table its ranges come from
                Q.bound32 = (const uint32_t *)b_bound.p; Q.bound64 = nullptr; Q.gran = 64u; Q.reps = 1u; Q.nb = (uint32_t)(S / 64);
                {   // k_prep wrote the table from the granule of the batch's first read to the one behind its last (kd_prep.h): clamp to that
                    const uint64_t hi1 = h_status[KDS_B_BOUND_HI1];
                    if (hi1 && hi1 - 1 < Q.nb) Q.nb = (uint32_t)(hi1 - 1);
                    Q.jlo = (uint32_t)std::min<uint64_t>(h_status[KDS_B_BOUND_LO], Q.nb);
                }
                Q.n_win = n_win; Q.span_slot = span_slot; Q.hot = nullptr; Q.cut = static_cut >= 2 ? static_cut : 0u;
                if (in_order) {
                    if (use_coop && rt.launch("k_plan_ranges", k_plan_ranges, (n_win + KD_BLOCK - 1) / KD_BLOCK, KD_BLOCK, 0, info, (kd_u64)ne, w0,
                                  n_win, W, slice, wl, wh, io, (const kd_u64 *)d_status, span_slot == (uint32_t)KDS_B_MAXSPAN ? H : 0u))
                        return hipfail("k_plan_ranges");
                } else if (!rows) {
                    // an UNSORTED batch of reads: counting sort by window, the regular reads' footprints / offsets scattered
                    // physically into window order (k_sort_scatter_reads); k_window then walks them like a sorted batch
                    const uint32_t n_bins = (uint32_t)((S + W - 1) / W);
                    const bool lds_bins = (size_t)n_bins * 4 <= (size_t)156 * 1024 && ne < (1ull << 32) && !getenv("KD_SORT_GLOBAL");   // (knob: measurement)
                    uint32_t reps = lds_bins ? 1u : 2u;   // (global counters, C3 shuffled on MI355X: 1: 4.40 ms, 2: 4.24, 4: 4.44, 8: 4.54 per step)
                    if (const char *e = getenv("KD_SORT_REPS")) reps = lds_bins ? 1u : (uint32_t)std::max(1, atoi(e));
                    const size_t n_cnt = ((size_t)n_bins + 1) * reps;
                    if ((rc2 = ensure(b_bincnt, n_cnt * 4)) || (rc2 = ensure(b_binoff, (n_cnt + 1) * 8)) || (rc2 = ensure(b_srec, ne * sizeof(KdSortRec))))
                        return rc2;
                    uint32_t *bc = (uint32_t *)b_bincnt.p;
                    kd_u64 *bo = (kd_u64 *)b_binoff.p;
                    if (rt.memset(bc, 0, n_cnt * 4)) return hipfail("k_sort_*");
                    if (lds_bins) {
                        // bin counters private to a workgroup in LDS (kd_plan.h): two workgroups of 1024 lanes per CU, one chunk each;
                        // no global atomics: count rows -> column scan -> bin offsets -> scatter
                        uint64_t want_wgs = (uint64_t)rt.n_cus() * ((size_t)n_bins * 4 <= (size_t)72 * 1024 ? 2 : 1);
                        if (const char *e = getenv("KD_SORT_WGS")) want_wgs = (uint64_t)std::max(1, atoi(e));   // (knob: tests, measurement)
                        const unsigned gr = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want_wgs, (ne + KD_SORT_BLOCK - 1) / KD_SORT_BLOCK));
                        const kd_u64 chunk = (((kd_u64)ne + gr - 1) / gr + KD_SORT_BLOCK - 1) / KD_SORT_BLOCK * KD_SORT_BLOCK;
                        const unsigned n_seg = (gr + KD_SORT_SEG - 1) / KD_SORT_SEG;
                        if ((rc2 = ensure(b_sortrows, (size_t)gr * n_bins * 4)) || (rc2 = ensure(b_sortseg, (size_t)n_seg * n_bins * 4))) return rc2;
                        uint32_t *rows = (uint32_t *)b_sortrows.p, *segt = (uint32_t *)b_sortseg.p;
                        if (rt.launch("k_sort_count", k_sort_count_lds, gr, KD_SORT_BLOCK, (size_t)n_bins * 4, info, (kd_u64)ne, chunk, W, n_bins, rows) ||
                            rt.launch("k_sort_colscan", k_sort_colscan, (unsigned)(((uint64_t)n_seg * n_bins + KD_BLOCK - 1) / KD_BLOCK), KD_BLOCK, 0, rows, gr, n_bins, segt) ||
                            rt.launch("k_sort_colscan2", k_sort_colscan2, (n_bins + KD_BLOCK - 1) / KD_BLOCK, KD_BLOCK, 0, segt, n_seg, n_bins, bc) ||
                            rt.launch("k_sort_scan", k_sort_scan, 1u, KD_SCAN_WIDE, 0, bc, bo, (uint32_t)n_cnt, d_status) ||
                            rt.launch("k_sort_scatter", k_sort_scatter_lds, gr, KD_SORT_BLOCK, (size_t)n_bins * 4, info, R, chunk, W, n_bins, (const uint32_t *)rows,
                                      (const uint32_t *)segt, (const kd_u64 *)bo, (KdSortRec *)b_srec.p))
                            return hipfail("k_sort_*");
                    } else {
                        const unsigned gr = (unsigned)((ne + KD_BLOCK - 1) / KD_BLOCK);
                        if (rt.launch("k_sort_count", k_sort_count<1>, gr, KD_BLOCK, 0, info, (kd_u64)ne, W, bc, reps) ||
                            rt.launch("k_sort_scan", k_sort_scan, 1u, KD_SCAN_WIDE, 0, bc, bo, (uint32_t)n_cnt, d_status) ||
                            rt.launch("k_sort_scatter", k_sort_scatter_reads, gr, KD_BLOCK, 0, info, R, W, bc, (const kd_u64 *)bo, reps,
                                      (KdSortRec *)b_srec.p))
                            return hipfail("k_sort_*");
                    }
                    if (use_coop && rt.launch("k_plan_ranges_sorted", k_plan_ranges_sorted, (n_win + KD_BLOCK - 1) / KD_BLOCK, KD_BLOCK, 0,
                                  (const kd_u64 *)bo, n_bins, w0, n_win, W, slice, wl, wh, io, (const kd_u64 *)d_status, span_slot, reps))
                        return hipfail("k_sort_*");
                    Q.bound32 = nullptr; Q.bound64 = (const kd_u64 *)bo; Q.gran = W; Q.reps = reps; Q.nb = n_bins; Q.jlo = 0;
                    walk_info = (const KdRInfo *)b_srec.p;
                    walk_R.seq_off = (const kd_u64 *)b_srec.p + 2; walk_R.cig_off = (const kd_u64 *)b_srec.p + 3; walk_R.n_cig = nullptr;
                    walk_R.osh = 1;
                } else {
                    // the ROWS of long reads: counting sort by window -> permutation `order`
                    const uint32_t n_bins = (uint32_t)((S + W - 1) / W);
                    if ((rc2 = ensure(b_order, ne * 4)) || (rc2 = ensure(b_bincnt, ((size_t)n_bins + 1) * 4)) ||
                        (rc2 = ensure(b_binoff, ((size_t)n_bins + 1) * 8)))
                        return rc2;
                    uint32_t *ord = (uint32_t *)b_order.p, *bc = (uint32_t *)b_bincnt.p;
                    kd_u64 *bo = (kd_u64 *)b_binoff.p;
                    const unsigned gr = (unsigned)((ne + (uint64_t)KD_BLOCK - 1) / (uint64_t)KD_BLOCK);
                    if (rt.memset(bc, 0, ((size_t)n_bins + 1) * 4) ||
                        rt.launch("k_sort_count", k_sort_count<1>, gr, KD_BLOCK, 0, info, (kd_u64)ne, W, bc, 1u) ||
                        rt.launch("k_sort_scan", k_sort_scan, 1u, KD_SCAN_WIDE, 0, bc, bo, n_bins, d_status) ||
                        rt.launch("k_sort_scatter", k_sort_scatter<1>, gr, KD_BLOCK, 0, info, (kd_u64)ne, W, bc, (const kd_u64 *)bo, ord))
                        return hipfail("k_sort_*");
                    Q.bound32 = nullptr; Q.bound64 = (const kd_u64 *)bo; Q.gran = W; Q.reps = 1u; Q.nb = n_bins; Q.jlo = 0;
                    order = ord;
                }
                if (!use_coop) {
                    // k_window: self-planned queue -- nothing but the list of windows with more than one slice to allocate
                    if ((rc2 = ensure(b_hot, (size_t)n_win * sizeof(KdHot)))) return rc2;
                    Q.hot = (KdHot *)b_hot.p;
                    const size_t lds = win_lds;
                    const unsigned grid = win_grid;
                    if (rows) {
                        walk_R.seq4 = (const uint8_t *)b_rows.p; walk_R.seq_off = (const kd_u64 *)b_rowoff.p;
                        walk_R.cig_off = nullptr; walk_R.n_cig = nullptr; walk_R.osh = 0;
                    }
                    last_nwin += n_win;
                    // the first pass carries the cold records' workgroups behind its persistent ones (kd_window.h: KdColdTail) -- when
                    // the reads it walks are the batch's own arrays (an unsorted batch's window-ordered copy has another layout)
                    KdColdTail tail;
                    tail.rec = nullptr; tail.cnt = nullptr; tail.evbase = tail.poolbase = nullptr; tail.ins = I;
                    tail.region_slots = 0; tail.n_regions = 0; tail.first_block = grid;
                    if (!rows && n_cold && walk_R.osh == 0 && cold_tail_on) {
                        tail.rec = (const KdColdRec *)cold; tail.cnt = (const uint32_t *)b_coldcnt.p; tail.evbase = (const kd_u64 *)b_coldev.p;
                        tail.poolbase = (const kd_u64 *)b_coldpool.p; tail.region_slots = cold_region; tail.n_regions = prep_regions;
                        cold_fused = true;
                    }
                    if (rows ? rt.launch("k_window_rows", k_window<true>, grid, KD_BLOCK, lds, walk_R, walk_info, order, T, Q, w0, W, H, Wh, slice, d_status, tail)
                             : rt.launch("k_window", k_window<false>, grid + tail.n_regions, KD_BLOCK, lds, walk_R, walk_info, order, T, Q, w0, W, H, Wh, slice, d_status, tail))
                        return hipfail("k_window");
                    return KD_OK;
                }
                // k_window_coop (planned queue): work item -> window table; an item is (window, slice of its candidates): at most one per window plus
                // one per `slice` (entry, window) candidate pairs, and an entry is a candidate of the windows its
                // footprint (<= max span + max lead, both known from k_prep / k_prep_long) can touch
                const uint64_t reach = h_status[span_slot] / W + h_status[KDS_B_MAXLEAD] / W + 4;   // whole-bin ranges included
                const uint64_t items_cap = (uint64_t)n_win + (ne * reach) / slice + 1;
                if ((rc2 = ensure(b_itemwin, items_cap * 4))) return rc2;
                uint32_t *iw = (uint32_t *)b_itemwin.p;
                // (up to 2^16 windows the scan's workgroup also writes the item -> window table; beyond, a kernel of its own)
                const bool fused_items = n_win <= 65536u;
                if (rt.launch("k_plan_scan", k_plan_scan, 1u, KD_SCAN_WIDE, 0, io, n_win, fused_items ? iw : (uint32_t *)nullptr, (kd_u64)items_cap, d_status) ||
                    (!fused_items && rt.launch("k_plan_items", k_plan_items, (n_win + KD_BLOCK - 1) / KD_BLOCK, KD_BLOCK, 0, (const kd_u64 *)io,
                                               n_win, iw, (kd_u64)items_cap, d_status)))
                    return hipfail("k_plan_scan");
                {
                    const uint32_t P = KD_COOP_PITCH(W);
                    const size_t lds = KD_COOP_LDS_BYTES(P);
                    const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024 - 512) / (lds + 64)));
                    const unsigned grid = std::max(1u, (unsigned)rt.n_cus() * per_cu);
                    if (rt.launch("k_window", k_window_coop, grid, KD_BLOCK, lds, walk_R, walk_info, order, T, (const kd_u64 *)wl, (const kd_u64 *)wh,
                                  (const kd_u64 *)io, (const uint32_t *)iw, (kd_u64)items_cap, w0, W, P, slice, d_status))
                        return hipfail("k_window_coop");
                    return KD_OK;
                }
            };
            // First pass over the batch's short regular reads.  Default: k_window (LDS histograms).  KD_MODE_STRIP: k_strip, the
            // site-major kernel (wavefront = strip of 64 sites, counters in registers) -- bit-identical, measured slower on
            // MI355X, kept as an independent implementation.  Planning is the same machinery with a
            // "window" of one strip.
            auto strip_pass = [&](const KdRInfo *info, uint64_t ne, bool in_order) -> int {
                int rc2;
                const uint32_t Ws = KD_STRIP;
                kd_u64 *swl = wl, *swh = wh, *sio = io;
                const uint64_t reach = h_status[KDS_B_MAXSPAN] / Ws + h_status[KDS_B_MAXLEAD] / Ws + 4;
                uint32_t slice = slice_cfg;
                if (!slice) {   // enough (strip, slice) items to keep every resident wavefront busy several times over
                    const uint64_t pairs = ne * (h_status[KDS_B_MAXSPAN] / Ws + 2), want = (uint64_t)rt.n_cus() * 4 * KD_STRIP_WGS * 8;
                    slice = (uint32_t)std::min<uint64_t>(32768, std::max<uint64_t>(256, (pairs / want + 63) / 64 * 64));
                }
                slice = std::min<uint32_t>(slice, 32768u);   // item-relative candidate indices are kept as u16 in LDS
                const uint32_t *order = nullptr;
                if (in_order) {
                    if (rt.launch("k_plan_ranges", k_plan_ranges, (ns_win + KD_BLOCK - 1) / KD_BLOCK, KD_BLOCK, 0, info, (kd_u64)ne, ws0,
                                  ns_win, Ws, slice, swl, swh, sio, (const kd_u64 *)d_status, 0u))
                        return hipfail("k_plan_ranges");
                } else {
                    const uint32_t n_bins = (uint32_t)((S + Ws - 1) / Ws);
                    if ((rc2 = ensure(b_order, ne * 4)) || (rc2 = ensure(b_bincnt, ((size_t)n_bins + 1) * 4)) ||
                        (rc2 = ensure(b_binoff, ((size_t)n_bins + 1) * 8)))
                        return rc2;
                    uint32_t *ord = (uint32_t *)b_order.p, *bc = (uint32_t *)b_bincnt.p;
                    kd_u64 *bo = (kd_u64 *)b_binoff.p;
                    const unsigned gr = (unsigned)((ne + KD_BLOCK - 1) / KD_BLOCK);
                    if (rt.memset(bc, 0, ((size_t)n_bins + 1) * 4) ||
                        rt.launch("k_sort_count", k_sort_count<1>, gr, KD_BLOCK, 0, info, (kd_u64)ne, Ws, bc, 1u) ||
                        rt.launch("k_sort_scan", k_sort_scan, 1u, KD_SCAN_WIDE, 0, bc, bo, n_bins, d_status) ||
                        rt.launch("k_sort_scatter", k_sort_scatter<1>, gr, KD_BLOCK, 0, info, (kd_u64)ne, Ws, bc, (const kd_u64 *)bo, ord) ||
                        rt.launch("k_plan_ranges_sorted", k_plan_ranges_sorted, (ns_win + KD_BLOCK - 1) / KD_BLOCK, KD_BLOCK, 0,
                                  (const kd_u64 *)bo, n_bins, ws0, ns_win, Ws, slice, swl, swh, sio, (const kd_u64 *)d_status,
                                  (uint32_t)KDS_B_MAXSPAN, 1u))
                        return hipfail("k_sort_*");
                    order = ord;
                }
                const uint64_t items_cap = (uint64_t)ns_win + (ne * reach) / slice + 1;
                if ((rc2 = ensure(b_itemwin, items_cap * 4))) return rc2;
                uint32_t *siw = (uint32_t *)b_itemwin.p;
                if (rt.launch("k_plan_scan", k_plan_scan, 1u, KD_SCAN_WIDE, 0, sio, ns_win, (uint32_t *)nullptr, (kd_u64)items_cap, d_status) ||
                    rt.launch("k_plan_items", k_plan_items, (ns_win + KD_BLOCK - 1) / KD_BLOCK, KD_BLOCK, 0, (const kd_u64 *)sio, ns_win, siw,
                              (kd_u64)items_cap, d_status))
                    return hipfail("k_plan_scan");
                const unsigned grid = std::max(1u, (unsigned)rt.n_cus() * (unsigned)KD_STRIP_WGS);   // resident workgroups: LDS and registers
                if (rt.launch("k_strip", k_strip, grid, KD_BLOCK, 0, R, info, order, T, (const kd_u64 *)swl, (const kd_u64 *)swh,
                              (const kd_u64 *)sio, (const uint32_t *)siw, (kd_u64)items_cap, ws0, slice, d_status))
                    return hipfail("k_strip");
                return KD_OK;
            };
            if (!n_reg_short) rc = KD_OK;      // (a batch of long reads only: nothing for the first pass)
            else if (mode == KD_MODE_STRIP) rc = strip_pass((const KdRInfo *)rinfo, n, sorted_input);
            else rc = window_pass((const KdRInfo *)rinfo, n, sorted_input, false, (uint32_t)KDS_B_MAXSPAN, W_first, coop);
            if (rc) return rc;
            if (n_long && h_status[KDS_B_MAXSEGSPAN] &&
                (rc = window_pass((const KdRInfo *)b_rowinfo.p, n_long, false, true, (uint32_t)KDS_B_MAXSEGSPAN, W_seg, false)))
                return rc;
            if (n_irreg &&
                rt.launch("k_pileup_wave_irreg", k_pileup_wave<true, true>,
                          (unsigned)((n_irreg + KD_WAVES_PER_BLOCK - 1) / KD_WAVES_PER_BLOCK), KD_BLOCK, 0, R, T, I,
                          (const uint32_t *)irreg, (kd_u64)n_irreg, (const KdRInfo *)rinfo, d_status))
                return hipfail("k_pileup_wave_irreg");
            // the batch's last kernel: the clip counters / insertion events of the clipped and inserted regular reads, and -- its
            // last workgroup -- the error classification (kd_errors.h: leaves at once when nothing was flagged)
            if (n_cold && !cold_fused) {
                if (rt.launch("k_cold_lane", k_cold_lane, prep_regions + 1u, KD_BLOCK, 0, R, T, I, (const KdColdRec *)cold,
                              (const uint32_t *)b_coldcnt.p, (const kd_u64 *)b_coldev.p, (const kd_u64 *)b_coldpool.p, cold_region, d_status,
                              (const KdRInfo *)rinfo, n_contigs))
                    return hipfail("k_cold_lane");
            } else if (defer_errors) {      // (kd_step: finish() launches it if the status words ask for it)
                errors_pending = true; err_R = R; err_windowed = 1u;
            } else if (rt.launch("k_errors", k_errors, 1u, KD_BLOCK, 0, R, T, (const KdRInfo *)rinfo, n_contigs, d_status, 1u))
                return hipfail("k_errors");
        } else {
            // (k_pileup_wave looks a read's insertion slots up by read index: spell the regular reads' out)
            if (n_cold && rt.launch("k_cold_slots", k_cold_slots, prep_regions, KD_BLOCK, 0, (const KdColdRec *)cold, (const uint32_t *)b_coldcnt.p,
                                    (const kd_u64 *)b_coldev.p, (const kd_u64 *)b_coldpool.p, cold_region, (uint32_t *)b_readev.p,
                                    (kd_u64 *)b_readpool.p))
                return hipfail("k_cold_slots");
            if (rt.launch("k_pileup_wave_all", k_pileup_wave<true, true>,
                          (unsigned)((n + KD_WAVES_PER_BLOCK - 1) / KD_WAVES_PER_BLOCK), KD_BLOCK, 0, R, T, I,
                          (const uint32_t *)nullptr, (kd_u64)n, (const KdRInfo *)rinfo, d_status) ||
                rt.launch("k_errors", k_errors, 1u, KD_BLOCK, 0, R, T, (const KdRInfo *)rinfo, n_contigs, d_status, 0u))
                return hipfail("k_pileup_wave_all");
        }
        reads_pushed += n;
        finalized = false; have_cns = false; have_inskeys = false;
        return KD_OK;
    }

    int push_host(const kd_batch &B) {
        const uint64_t n = B.n_reads;
        if (!n) return KD_OK;
        const void *src[9] = {B.contig, B.pos0, B.flag, B.seq_off, B.seq_len, B.cig_off, B.n_cig, B.seq4, B.cigar};
        const size_t bytes[9] = {n * 4, n * 4, n * 4, n * 8, n * 4, n * 8, n * 4, (size_t)B.seq4_bytes + 64,
                                 (size_t)B.cigar_words * 4 + 8};
        const size_t copy[9] = {n * 4, n * 4, n * 4, n * 8, n * 4, n * 8, n * 4, (size_t)B.seq4_bytes,
                                (size_t)B.cigar_words * 4};
        int rc;
        // the previous batch's kernels read the staging buffers: drain before overwriting
        if (rt.sync()) return hipfail("push: sync");
        for (int k = 0; k < 9; k++) {
            if ((rc = ensure(b_stage[k], bytes[k]))) return rc;
            if (copy[k] && rt.h2d(b_stage[k].p, src[k], copy[k])) return hipfail("push: h2d");
        }
        kd_batch D = B;
        D.contig = (const uint32_t *)b_stage[0].p; D.pos0 = (const int32_t *)b_stage[1].p;
        D.flag = (const uint32_t *)b_stage[2].p; D.seq_off = (const uint64_t *)b_stage[3].p;
        D.seq_len = (const uint32_t *)b_stage[4].p; D.cig_off = (const uint64_t *)b_stage[5].p;
        D.n_cig = (const uint32_t *)b_stage[6].p; D.seq4 = (const uint8_t *)b_stage[7].p;
        D.cigar = (const uint32_t *)b_stage[8].p;
        return push_device(D);
    }

    // ---- device-side ingest (kd_ingest.h): the FILE's bytes -> inflated stream -> kd_batch arrays in HBM -> push_device ----
    // file / blocks: host memory (the mapped file, the BGZF block table of kd_bgzf_plan_open); total_out: length of the inflated
    // stream, hdr_end: offset of its first record.  KD_E_UNSUPPORTED: the host decoder must read this file.
    int ingest_bam(const uint8_t *file, uint64_t file_bytes, const void *blocks, uint32_t n_blocks, uint64_t total_out, uint64_t hdr_end,
                   uint64_t *stats) {
        typedef std::chrono::steady_clock clk;
        const clk::time_point t0 = clk::now();
        auto us = [](clk::time_point a, clk::time_point b) { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        int rc;
        if (stats) for (int k = 0; k < 8; k++) stats[k] = 0;
        if (!n_blocks || hdr_end >= total_out) return KD_OK;        // a header and nothing else
        if (rt.sync()) return hipfail("ingest: sync");              // (the previous batch's kernels read the staging buffers)
        {   // the file, its inflated stream and the batch are all resident at once on this path: a file too big for that is the
            // streamed host decoder's (chunks of 64 MiB)
            const uint64_t have = b_gi_file.cap + b_gi_out.cap + b_gi_tok.cap, need = file_bytes + 2 * total_out + (knob_inflate == 2 ? 4 * (uint64_t)gi2_tok_off(total_out, n_blocks) : 0);
            const uint64_t free_now = rt.free_bytes();
            if (need > have && need - have > free_now / 10 * 8)
                return fail(KD_E_UNSUPPORTED, "the file (" + std::to_string(file_bytes >> 20) + " MiB, " + std::to_string(total_out >> 20) +
                                                  " MiB inflated) does not fit the GPU's free memory at once: the streamed host decoder reads it");
        }
        // KD_INFLATE=2 (round 6): the two-pass inflate (kd_gpu_inflate2.h: a LANE per block records the matches, a wavefront per block
        // resolves them); 1: the one-pass kernel of rounds 3 - 5 (a wavefront per block).  The token lists: 4 / 3 of the inflated size
        const bool two_pass = knob_inflate == 2;
        const size_t tok_words = two_pass ? (size_t)gi2_tok_off(total_out, n_blocks) + 16 : 0;
        if (two_pass && ((rc = ensure(b_gi_tok, tok_words * 4)) || (rc = ensure(b_gi_ntok, (size_t)n_blocks * 4)) || (rc = ensure(b_gi_work, (size_t)GI_MAX_GROUPS * 4)))) return rc;
        if ((rc = ensure(b_gi_file, file_bytes + 64)) || (rc = ensure(b_gi_blocks, (size_t)n_blocks * sizeof(GiBlock))) ||
            (rc = ensure(b_gi_out, total_out + 64)) || (rc = ensure(b_gi_bstat, (size_t)n_blocks * 4)) || (rc = ensure(b_gi_start, (size_t)n_blocks * 8)) ||
            (rc = ensure(b_gi_cnt, (size_t)n_blocks * 8 * 3)) || (rc = ensure(b_gi_tot, 64)))
            return rc;
        if (rt.memset((uint8_t *)b_gi_file.p + file_bytes, 0, 64) ||
            rt.h2d(b_gi_blocks.p, blocks, (size_t)n_blocks * sizeof(GiBlock)) || rt.memset(b_gi_tot.p, 0, 64))
            return hipfail("ingest: h2d");
        const GiBlock *d_blocks = (const GiBlock *)b_gi_blocks.p;
        uint32_t *bstat = (uint32_t *)b_gi_bstat.p;
        // the file goes up in pieces (rt.upload: pinned double buffer, its own copy stream); the blocks whose bytes have arrived are
        // inflated while the next piece is on its way
        {
            const GiBlock *hb = (const GiBlock *)blocks;
            uint32_t next_block = 0, group = 0;
            if (two_pass && (rt.memset(b_gi_work.p, 0, (size_t)GI_MAX_GROUPS * 4) || rt.side_fork())) return hipfail("ingest: side streams");
            auto after = [&](size_t there) -> int {
                uint32_t e = next_block;
                while (e < n_blocks && hb[e].in_off + hb[e].in_len + 8 <= there) e++;     // (+ 8: the block's trailer; the kernel's last dword load ends inside it)
                if (there >= file_bytes) e = n_blocks;
                // one-pass: one wavefront works ~20 ms on a block and launches on one stream run one after the other: a launch of fewer
                // blocks than two rounds of the chip's slots (26 wavefronts per CU) leaves most of it idle for that long.
                // two-pass: a LANE works that long on its block; the groups go to side streams in turn, so that the groups whose bytes have
                // arrived share the chip (a group alone cannot fill it: 16 384 blocks are one wavefront per CU) -- smaller groups, and the
                // last ones as many as the group counters allow
                const uint32_t min_group = two_pass ? std::max<uint32_t>(32u * (uint32_t)rt.n_cus(), (n_blocks + GI_MAX_GROUPS - 2u) / (GI_MAX_GROUPS - 1u))
                                                    : 52u * (uint32_t)rt.n_cus();
                if (e == next_block || (e < n_blocks && e - next_block < min_group && !getenv("KD_UPLOAD_CHUNK"))) return 0;
                if (two_pass && group + 1u >= GI_MAX_GROUPS && e < n_blocks) return 0;      // (tests with tiny upload pieces: the last counter takes the rest)
                const uint32_t cnt = e - next_block;
                int bad;
                if (two_pass) {
                    const int sd = (int)(group % (uint32_t)Rt::N_SIDE);
                    const unsigned wgs = (unsigned)std::min<uint32_t>((cnt + GI2_WG - 1u) / GI2_WG, (uint32_t)rt.n_cus());      // what can be resident (a workgroup = a CU's LDS); the rest through the counter
                    bad = rt.side_after_upload(sd) ||
                          rt.launch_side(sd, "k_inflate_tokens", k_inflate_tokens, wgs, GI2_WG, (size_t)GI2_LDS_BYTES, (const uint8_t *)b_gi_file.p, d_blocks + next_block, cnt,
                                         (uint8_t *)b_gi_out.p, (uint32_t *)b_gi_tok.p, (uint32_t *)b_gi_ntok.p + next_block, bstat + next_block, next_block,
                                         (uint32_t *)b_gi_work.p + group) ||
                          rt.launch_side(sd, "k_inflate_resolve", k_inflate_resolve, cnt, KD_WAVE, 0, d_blocks + next_block, cnt, (uint8_t *)b_gi_out.p,
                                         (const uint32_t *)b_gi_tok.p, (const uint32_t *)b_gi_ntok.p + next_block, (const uint32_t *)(bstat + next_block), next_block);
                    group++;
                } else
                    bad = rt.launch("k_gpu_inflate", k_gpu_inflate, cnt, KD_WAVE, 0, (const uint8_t *)b_gi_file.p, d_blocks + next_block, cnt,
                                    (uint8_t *)b_gi_out.p, bstat + next_block);
                next_block = e;
                return bad;
            };
            if (rt.upload(b_gi_file.p, file, file_bytes, after)) return hipfail("ingest: upload / k_gpu_inflate");
            if (two_pass && rt.side_join()) return hipfail("ingest: side streams");
        }
        const bool trace = getenv("KD_INGEST_TRACE") != nullptr;
        const clk::time_point t_up = clk::now();
        if (trace) { (void)rt.sync(); fprintf(stderr, "kd ingest: upload loop %.1f ms (host: copies into the pinned pieces), inflate done %.1f ms after start\n", us(t0, t_up) / 1e3, us(t0, clk::now()) / 1e3); }
        kd_u64 *start = (kd_u64 *)b_gi_start.p, *c_rec = (kd_u64 *)b_gi_cnt.p, *c_seq = c_rec + n_blocks, *c_cig = c_seq + n_blocks;
        kd_u64 *tot = (kd_u64 *)b_gi_tot.p;          // [0..2] kept records / packed-base bytes / CIGAR words, [3] records seen, [4] status, [5] diagnosis, [6] blocks with a wrong CRC
        KdBam Bm;
        Bm.d = (const uint8_t *)b_gi_out.p; Bm.n = total_out; Bm.hdr_end = hdr_end; Bm.blocks = d_blocks; Bm.n_blocks = n_blocks; Bm.n_ref = n_contigs;
        const unsigned gb = (n_blocks + KD_BLOCK - 1) / KD_BLOCK;
        // every block's CRC-32 against its trailer, as the host reader and htslib check it (KD_BGZF_NO_CRC=1: measurement)
        if (!getenv("KD_BGZF_NO_CRC") &&
            rt.launch("k_bgzf_crc", k_bgzf_crc, std::min<unsigned>(n_blocks, 16u * (unsigned)rt.n_cus()), KD_WAVE, 0, (const uint8_t *)b_gi_file.p, d_blocks, n_blocks,
                      (const uint8_t *)b_gi_out.p, (uint32_t *)(tot + 6)))
            return hipfail("k_bgzf_crc");
        if (rt.launch("k_bam_starts", k_bam_starts, n_blocks, KD_WAVE, 0, Bm, start) ||
            rt.launch("k_bam_count", k_bam_count, gb, KD_BLOCK, 0, Bm, (const kd_u64 *)start, (const uint32_t *)bstat, c_rec, c_seq, c_cig, tot + 3, (uint32_t *)(tot + 4)) ||
            rt.launch("k_bam_scan", k_bam_scan, 1u, KD_SCAN_WIDE, 0, c_rec, c_seq, c_cig, n_blocks, tot))
            return hipfail("ingest kernels");
        kd_u64 h_tot[8];
        if (rt.d2h_small(h_tot, tot, 64)) return hipfail("ingest: totals d2h");
        const uint32_t st = (uint32_t)h_tot[4];
        if (!(st & KD_INGEST_INFLATE) && (uint32_t)h_tot[6])
            return fail(KD_E_IO, "BGZF block CRC-32 mismatch in " + std::to_string((uint32_t)h_tot[6]) + " block(s) (device-side ingest)");
        if (st & (KD_INGEST_INFLATE | KD_INGEST_RECORD)) return fail(KD_E_IO, st & KD_INGEST_INFLATE ? "BGZF inflate failed (device-side ingest)" : "malformed or truncated BAM record (device-side ingest)");
        if (st & (KD_INGEST_CHAIN | KD_INGEST_HOST))
            return fail(KD_E_UNSUPPORTED, st & KD_INGEST_HOST ? "a CIGAR in a CG:B,I tag: the host decoder reads this file"
                                                               : "the device-side record walk could not verify its starts (the part from BGZF block " + std::to_string((h_tot[5] >> 32) - 1) +
                                                                     " did not end on the start guessed in block " + std::to_string((h_tot[5] & 0xffffffffu) - 1) + "): the host decoder reads this file");
        const uint64_t n = h_tot[0], seq_bytes = h_tot[1], cig_words = h_tot[2];
        if (stats) { stats[0] = h_tot[3]; stats[1] = n; stats[2] = total_out; stats[3] = n_blocks; }
        if (!n) return KD_OK;
        const size_t bytes[9] = {n * 4, n * 4, n * 4, n * 8, n * 4, n * 8, n * 4, (size_t)seq_bytes + 64, (size_t)cig_words * 4 + 8};
        for (int k = 0; k < 9; k++) if ((rc = ensure(b_stage[k], bytes[k]))) return rc;
        if ((rc = ensure(b_gi_recat, n * 16))) return rc;
        KdBamOut O;
        O.contig = (uint32_t *)b_stage[0].p; O.pos0 = (int32_t *)b_stage[1].p; O.flag = (uint32_t *)b_stage[2].p; O.seq_off = (kd_u64 *)b_stage[3].p;
        O.seq_len = (uint32_t *)b_stage[4].p; O.cig_off = (kd_u64 *)b_stage[5].p; O.n_cig = (uint32_t *)b_stage[6].p;
        O.seq4 = (uint8_t *)b_stage[7].p; O.cigar = (uint32_t *)b_stage[8].p; O.rec_at = (kd_u64 *)b_gi_recat.p; O.cig_at = (kd_u64 *)b_gi_recat.p + n;
        if (rt.memset((uint8_t *)b_stage[7].p + seq_bytes, 0, 64) ||
            rt.launch("k_bam_fields", k_bam_fields, gb, KD_BLOCK, 0, Bm, (const kd_u64 *)start, (const kd_u64 *)c_rec, (const kd_u64 *)c_seq, (const kd_u64 *)c_cig, O) ||
            rt.launch("k_bam_payload", k_bam_payload, (unsigned)((n + KD_WAVE - 1) / KD_WAVE), KD_WAVE, 0, Bm, (kd_u64)n, O, (kd_u64)seq_bytes, (kd_u64)cig_words))
            return hipfail("ingest kernels");
        kd_batch D;
        memset(&D, 0, sizeof D);
        D.n_reads = n; D.seq4_bytes = seq_bytes; D.cigar_words = cig_words;
        D.contig = O.contig; D.pos0 = O.pos0; D.flag = O.flag; D.seq_off = (const uint64_t *)O.seq_off; D.seq_len = O.seq_len;
        D.cig_off = (const uint64_t *)O.cig_off; D.n_cig = O.n_cig; D.seq4 = O.seq4; D.cigar = O.cigar;
        const clk::time_point t1 = clk::now();
        if (stats && !rt.sync()) stats[5] = us(t0, clk::now());      // (measurement: the batch exists)
        rc = push_device(D);
        if (stats) stats[6] = us(t1, clk::now());
        return rc;
    }

    // ---- finalize: insertion multiset -> per-site maxima (the winner / tie is read off them by the consensus kernels) ----
    // Two halves, so that kd_step can queue the consensus behind the reduction and read everything back in ONE round trip:
    // finalize_launch() queues the kernels, finalize_check() looks at the status words the caller has fetched since.
    bool fin_launched = false;      // the hash reduction ran (there were insertion events)
    // kd_step: the batch's error classification (k_errors: one workgroup that leaves after two loads unless a kernel flagged something)
    // is not queued behind the record loop but launched by finish() WHEN the step's status words say a read was flagged -- the words
    // it looks at are the ones finish() reads back anyway, and the batch's arrays are the caller's until kd_step returns.  One dispatch
    // and one dependent gap less on every step (round 6: 3.5 + ~5 us; a 1/8 shard of C3 is 0.32 ms); a flagged step pays one more
    // round trip.  Only kd_step: between kd_push_batch calls the classification must see a batch before the next one overwrites rinfo.
    bool defer_errors = false, errors_pending = false;
    KdReads err_R; uint32_t err_windowed = 0;
    bool cns_meta_fresh = false;    // k_ins_flag has just left the consensus run's per-contig words initialised
    KdInsTab fin_H;
    kd_u64 *meta_coff() const { return d_status + KDS_COUNT; }
    uint32_t *meta_mm() const { return reinterpret_cast<uint32_t *>(d_status + KDS_COUNT + n_contigs + 1); }
    size_t meta_bytes() const { return ((size_t)n_contigs + 1) * 8 + (size_t)n_contigs * 8; }

    int reduce_insertions(int attempt) {
        const uint64_t n_ev = h_status[KDS_N_EV];
        KdIns I = insdesc();
        KdInsTab &H = fin_H;
        H.seed = 0x9e3779b97f4a7c15ULL * (uint64_t)(attempt + 1);
        // KD_TEST_INS_COLLIDE=1 (tests): two possible keys in the first attempt, so that different insertions share a key, the
        // byte-for-byte verification fails and the re-seeded second attempt has to put everything right
        const bool collide = getenv("KD_TEST_INS_COLLIDE") != nullptr;
        H.key_mask = (collide && attempt == 0) ? 0x2ULL : ~0ULL;
        // grids are sized by the upper bound n_ev; the kernels stop at the device-side count of selected events
        const unsigned ge = (unsigned)((n_ev + KD_BLOCK - 1) / KD_BLOCK), g4 = (unsigned)((n_ev + KD_INS_CHUNK - 1) / KD_INS_CHUNK);
        // (attempt 0: k_ins_insert zeroes the collision counter and marks the events itself)
        if (attempt && rt.memset(d_status + KDS_INS_COLLISION, 0, 8)) return hipfail("finalize: memset status");
        if (rt.launch("k_ins_insert", k_ins_insert, ge, KD_BLOCK, 0, I, H, (kd_u64)n_ev, tabs(), attempt ? 0u : 1u, d_status, meta_coff(), meta_mm(), n_contigs,
                      ins_site_flags ? (const uint8_t *)b_flag.p - alloc_lo : (const uint8_t *)nullptr))
            return hipfail("k_ins_insert");
        if (rt.launch("k_ins_verify_max", k_ins_verify_max, g4, KD_BLOCK, 0, I, H, (kd_u64)n_ev, (kd_u64 *)b_best.p - alloc_lo,
                      (kd_u64 *)b_best2.p - alloc_lo, d_status))
            return hipfail("k_ins_verify_max");
        ins_dirty_ev = n_ev; ins_dirty_tab = H; ins_dirty_bias = alloc_lo;
        return KD_OK;
    }

    int finalize_launch() {
        int rc;
        if (!tables_ready && (rc = prepare_tables())) return rc;   // nothing was pushed: all-zero tables
        if ((rc = ins_cleanup())) return rc;
        // best_a[] / best_b[] / flag[] (and changes[], kd_consensus_run) are site-indexed and SHARD-LOCAL like the tables: allocated
        // for the sites [alloc_lo, alloc_hi) only, the pointers handed to the kernels biased by -alloc_lo so that they keep
        // indexing with global sites (8 ranks x one C3-sized interval each: 18 B/site x 5 M sites per rank, not x 40 M).
        // best_a[] / best_b[] are all-zero between reductions (k_ins_cleanup): zeroed once, when (re)allocated.
        const size_t n_local = (size_t)(alloc_hi - alloc_lo) + 64;
        if (b_best.cap < n_local * 8 || b_best2.cap < n_local * 8) {
            if ((rc = ensure(b_best, n_local * 8)) || (rc = ensure(b_best2, n_local * 8))) return rc;
            if (rt.memset(b_best.p, 0, b_best.cap) || rt.memset(b_best2.p, 0, b_best2.cap)) return hipfail("finalize: memset best");
        }
        // the event counts are exact as of the last push (k_prep reserves the slots, push_device reads them back)
        const uint64_t n_ev = h_status[KDS_N_EV];
        fin_launched = false;
        KdInsTab &H = fin_H;
        H = KdInsTab();
        KdTabs T = tabs();
        if (n_ev) {
            uint64_t cap = 1024;
            while (cap < 2 * n_ev) cap <<= 1;
            if (cap >= 0xfffffff0ULL) return fail(KD_E_NOMEM, "too many insertion events");
            if (cap * 8 > b_hkey.cap) {   // a new (larger) table: zero it once
                if ((rc = ensure(b_hkey, cap * 8)) || (rc = ensure(b_hcnt, cap * 4)) || (rc = ensure(b_hrep, cap * 4))) return rc;
                if (rt.memset(b_hkey.p, 0, b_hkey.cap) || rt.memset(b_hcnt.p, 0, b_hcnt.cap)) return hipfail("finalize: memset hash");
            }
            // the site test: per event in k_ins_insert (measured: +11 us per million events), or once per site by k_ins_flag (a dependent
            // launch, ~10 us, + 4.4 us per million sites) where the events are many for the sites -- long reads: C5 has 10 M events on 1 M sites
            ins_site_flags = knob_ins_site_flags >= 0 ? knob_ins_site_flags != 0 : n_ev > 1000000ULL + (uint64_t)(alloc_hi - alloc_lo) * 2 / 5;
            if ((rc = ensure(b_evslot, n_ev * 4)) || (ins_site_flags && (rc = ensure(b_flag, n_local)))) return rc;
            hash_cap = cap;
            H.key = (kd_u64 *)b_hkey.p; H.cnt = (uint32_t *)b_hcnt.p; H.rep = (uint32_t *)b_hrep.p;
            H.ev_slot = (uint32_t *)b_evslot.p; H.cap = cap; H.sites = S;
        }
        // the words the reduction's verification and the consensus run start from: left by k_ins_insert on its way (round 5: it also
        // decides per event whether its site can emit an insertion at all -- rounds 2 - 4 flagged every site of the shard first); a
        // batch without insertion events has no k_ins_insert: k_ins_flag's minimal grid writes them
        if (!n_ev) ins_site_flags = false;
        const uint64_t flag_threads = ins_site_flags ? (alloc_hi - alloc_lo) / 4 : std::min<uint64_t>((uint64_t)n_contigs + 1, 65536);
        if ((!n_ev || ins_site_flags) &&
            rt.launch("k_ins_flag", k_ins_flag, (unsigned)((flag_threads + KD_BLOCK - 1) / KD_BLOCK), KD_BLOCK, 0, T, (kd_u64)alloc_lo, (kd_u64)alloc_hi,
                      ins_site_flags ? (uint8_t *)b_flag.p - alloc_lo : (uint8_t *)nullptr, d_status, meta_coff(), meta_mm(), n_contigs))
            return hipfail("k_ins_flag");
        cns_meta_fresh = true;
        if (n_ev) {
            if ((rc = reduce_insertions(0))) return rc;
            fin_launched = true;
        }
        return KD_OK;
    }

    // h_status: the status words as fetched BEHIND finalize_launch's kernels
    int finalize_check(uint64_t *err_read, bool *redone = nullptr) {
        int rc;
        if (h_status[KDS_INTERNAL]) return fail(KD_E_INTERNAL, "insertion event buffers overran (internal error)");
        KD_PHASE_REPORT(h_status)
        if (h_status[KDS_ERR_READ] != ~0ULL) {
            // the reference raises for the first failing read of the earliest-appearing contig (kindel.py:150-151)
            std::vector<kd_u64> fi(n_contigs), ef(n_contigs);
            std::vector<uint32_t> ec(n_contigs);
            if (rt.d2h(fi.data(), d_first_idx, (size_t)n_contigs * 8) || rt.d2h(ef.data(), d_err_first, (size_t)n_contigs * 8) ||
                rt.d2h(ec.data(), d_err_code, (size_t)n_contigs * 4))
                return hipfail("finalize: error state d2h");
            uint32_t cs = n_contigs;
            for (uint32_t c = 0; c < n_contigs; c++)
                if (ef[c] != ~0ULL && (cs == n_contigs || fi[c] < fi[cs])) cs = c;
            const uint64_t rd_idx = cs < n_contigs ? ef[cs] : h_status[KDS_ERR_READ];
            const uint64_t code = cs < n_contigs ? ec[cs] : 0;
            if (err_read) *err_read = rd_idx;
            const std::string at = " (read " + std::to_string(rd_idx) + ")";
            if (code == 1) return fail(KD_E_BASE, "base outside A,C,G,T,N in an aligned or clipped segment" + at);
            if (code == 2) return fail(KD_E_RANGE, "list index out of range: alignment runs off the reference" + at);
            if (code == 3) return fail(KD_E_CIGAR, "mapped read with CIGAR '*'" + at);
            return fail(KD_E_INTERNAL, "read flagged by a kernel but no reference exception reproduced" + at);
        }
        n_ev_final = h_status[KDS_N_EV]; pool_final = h_status[KDS_POOL];
        if (fin_launched) {
            for (int attempt = 1; h_status[KDS_INS_COLLISION] != 0; attempt++) {   // a 64-bit hash collision (never seen): re-seed
                if (attempt >= 8) return fail(KD_E_INTERNAL, "insertion hash: repeated 64-bit collisions");
                if (redone) *redone = true;
                if ((rc = ins_cleanup()) || (rc = reduce_insertions(attempt)) || (rc = fetch_status())) return rc;
            }
        }
        finalized = true; have_cns = false; have_inskeys = false;
        return KD_OK;
    }

    int finalize(uint64_t *err_read) {
        int rc;
        if ((rc = finalize_launch())) return rc;
        // ONE status read-back: deferred reference exceptions, buffer overruns and the hash verification
        if ((rc = fetch_status())) return rc;
        return finalize_check(err_read);
    }

    int get_stats(uint64_t out[4]) {
        int rc;
        if ((rc = fetch_status())) return rc;
        out[0] = h_status[KDS_ST_READS]; out[1] = h_status[KDS_ST_ALIGNED]; out[2] = h_status[KDS_ST_WALKED];
        out[3] = h_status[KDS_ST_INS];
        return KD_OK;
    }

    int get_tables(uint32_t contig, uint32_t n_ch, const uint32_t *channels, uint32_t *out) {
        if (contig >= n_contigs) return fail(KD_E_ARG, "kd_get_tables: bad contig");
        const size_t L1 = (size_t)clen[contig] + 1;
        int rc;
        if (!tables_ready && (rc = prepare_tables())) return rc;
        // the part of the contig inside this context's allocation; sites of other shards read as zero
        const uint64_t c0 = cbase[contig], c1 = c0 + L1;
        const uint64_t a = std::max(c0, alloc_lo), b = std::min(c1, alloc_hi);
        for (uint32_t k = 0; k < n_ch; k++) {
            if (channels[k] >= KDC_NCH) return fail(KD_E_ARG, "kd_get_tables: bad channel");
            uint32_t *dst = out + (size_t)k * L1;
            if (a > c0 || b < c1 || a >= b) memset(dst, 0, L1 * 4);
            if (a < b && rt.d2h(dst + (a - c0), d_tab + (size_t)channels[k] * pitch + (a - alloc_lo), (size_t)(b - a) * 4))
                return hipfail("kd_get_tables: d2h");
        }
        return KD_OK;
    }

    // The insertions dicts for the API (kd_get_insertions): the device reduction only looks at the events on sites where an
    // insertion can be emitted, so the full multiset {(site, string): count} is built here, on demand, from the event list.
    int load_inskeys() {
        if (have_inskeys) return KD_OK;
        h_inskeys.clear(); h_insbytes.clear();
        if (n_ev_final) {
            std::vector<kd_u64> off(n_ev_final);
            std::vector<uint32_t> site(n_ev_final), len(n_ev_final);
            std::vector<uint8_t> pool(pool_final + 1);
            if (rt.d2h(site.data(), b_ev_site.p, n_ev_final * 4) || rt.d2h(len.data(), b_ev_len.p, n_ev_final * 4) ||
                rt.d2h(off.data(), b_ev_off.p, n_ev_final * 8) || (pool_final && rt.d2h(pool.data(), b_pool.p, pool_final)))
                return hipfail("kd_get_insertions: d2h");
            // (hash, event) pairs sorted: equal keys become neighbours; equal hashes are split by comparing the bytes
            std::vector<std::pair<uint64_t, uint32_t>> hv;
            hv.reserve(n_ev_final);
            for (uint64_t e = 0; e < n_ev_final; e++) {
                if (site[e] == KD_EV_DROPPED || site[e] >= S || off[e] + len[e] > pool_final) continue;
                uint64_t h = ((uint64_t)site[e] << 32 | len[e]) * 0x9e3779b97f4a7c15ULL;
                for (uint32_t b = 0; b < len[e]; b++) h = (h ^ pool[off[e] + b]) * 0x100000001b3ULL;
                hv.emplace_back(h, (uint32_t)e);
            }
            std::sort(hv.begin(), hv.end());
            auto same = [&](uint32_t a, uint32_t b) {
                return site[a] == site[b] && len[a] == len[b] && (len[a] == 0 || memcmp(&pool[off[a]], &pool[off[b]], len[a]) == 0);
            };
            for (size_t i = 0; i < hv.size();) {
                size_t j = i;
                while (j < hv.size() && hv[j].first == hv[i].first) j++;
                std::vector<char> done(j - i, 0);
                for (size_t a = i; a < j; a++) {
                    if (done[a - i]) continue;
                    uint32_t cnt = 0;
                    for (size_t b = a; b < j; b++)
                        if (!done[b - i] && same(hv[a].second, hv[b].second)) { done[b - i] = 1; cnt++; }
                    const uint32_t r = hv[a].second;   // smallest event index of the key (pairs are sorted)
                    h_inskeys.push_back({site[r], cnt, len[r], off[r], r});
                }
                i = j;
            }
            std::sort(h_inskeys.begin(), h_inskeys.end(), [](const InsKey &a, const InsKey &b) {
                return a.site != b.site ? a.site < b.site : a.rep < b.rep;
            });
            static const char N2C[17] = "=ACMGRSVTWYHKDBN";
            for (auto &k : h_inskeys) {
                const uint64_t o = h_insbytes.size();
                for (uint32_t b = 0; b < k.len; b++) h_insbytes.push_back((uint8_t)N2C[pool[k.off + b] & 15]);
                k.off = o;
            }
        }
        have_inskeys = true;
        return KD_OK;
    }

    int get_insertions(uint32_t contig, uint64_t *n_keys, uint64_t *n_bytes, uint32_t *site, uint32_t *count,
                       uint32_t *len, uint64_t *off, uint8_t *bytes) {
        if (contig >= n_contigs) return fail(KD_E_ARG, "kd_get_insertions: bad contig");
        if (!finalized) return fail(KD_E_ARG, "kd_get_insertions: call kd_finalize first");
        int rc;
        if ((rc = load_inskeys())) return rc;
        const uint64_t lo = cbase[contig], hi = cbase[contig] + clen[contig];  // slots 0..len inclusive
        uint64_t nk = 0, nb = 0;
        for (const auto &k : h_inskeys)
            if (k.site >= lo && k.site <= hi) {
                if (site) {
                    site[nk] = (uint32_t)(k.site - lo); count[nk] = k.count; len[nk] = k.len; off[nk] = nb;
                    if (k.len) memcpy(bytes + nb, h_insbytes.data() + k.off, k.len);
                }
                nk++; nb += k.len;
            }
        if (n_keys) *n_keys = nk;
        if (n_bytes) *n_bytes = nb;
        return KD_OK;
    }

    // ---- consensus ----
    // consensus_launch() queues the kernels; consensus_collect() takes what they left in the run's metadata block
    //   u64 contig_off[n_contigs + 1] | u32 minmax[2 n_contigs]        (behind the status words; initialised on the device by k_ins_flag)
    //   u64 patch_off[np] | patch_start[np] | patch_end[np]            (b_patch; only with CDR patches: --realign)
    // once the caller has copied it back (consensus_run: its own copy; kd_step: together with the status words and the FASTA).
    uint64_t cns_tile_first = 0, cns_tiles = 0, cns_cap = 0;
    uint32_t cns_patches = 0;
    // host_out (may be NULL): the caller's output buffer as the DEVICE addresses it (pinned host memory, Rt::device_view): k_cns_emit
    // then stores the consensus bytes there as well, as it emits them (kd_cns.h) -- no device-to-host copy behind the kernel
    int consensus_launch(uint32_t min_depth, uint32_t n_patches, const uint64_t *ps, const uint64_t *pe, uint8_t *host_out = nullptr, uint64_t host_cap = 0) {
        int rc;
        const uint64_t tile_first = g_lo / KD_CNS_TILE;
        const uint64_t n_tiles = std::max<uint64_t>(1, (std::min<uint64_t>(S, g_hi) + KD_CNS_TILE - 1) / KD_CNS_TILE - tile_first);
        const uint64_t cap = n_tiles * KD_CNS_TILE + h_status[KDS_POOL] + 64;
        cns_tile_first = tile_first; cns_tiles = n_tiles; cns_cap = cap; cns_patches = n_patches;
        if ((rc = ensure(b_changes, (size_t)(alloc_hi - alloc_lo) + 64))) return rc;   // shard-local (read back through copy_changes)
        const bool self_scan = n_tiles <= KD_CNS_SELF_SCAN;
        if ((rc = ensure(b_cns, cap)) || (rc = ensure(b_tilesum, n_tiles * 8)) || (rc = ensure(b_tilemm, n_tiles * sizeof(KdTileMM))) ||
            (!self_scan && (rc = ensure(b_tileoff, (n_tiles + 1) * 8))))
            return rc;
        KdTabs T = tabs();
        KdIns I = insdesc();
        if (!cns_meta_fresh &&      // a second run on the same tables (another min_depth, CDR patches): the per-contig words again
            rt.launch("k_ins_flag", k_ins_flag, (unsigned)((std::min<uint64_t>((uint64_t)n_contigs + 1, 65536) + KD_BLOCK - 1) / KD_BLOCK), KD_BLOCK, 0, T,
                      (kd_u64)alloc_lo, (kd_u64)alloc_hi, (uint8_t *)nullptr, d_status, meta_coff(), meta_mm(), n_contigs))
            return hipfail("k_ins_flag");
        cns_meta_fresh = false;
        kd_u64 *d_poff = nullptr, *d_ps = nullptr, *d_pe = nullptr;
        if (n_patches) {
            const size_t np = n_patches;
            if ((rc = ensure(b_patch, 3 * np * 8))) return rc;
            std::vector<uint64_t> up(3 * np, ~0ULL);
            for (uint32_t k = 0; k < n_patches; k++) { up[np + k] = ps[k]; up[2 * np + k] = pe[k]; }
            if (rt.h2d(b_patch.p, up.data(), 3 * np * 8) || rt.sync()) return hipfail("consensus: h2d");   // (`up` is a local)
            d_poff = (kd_u64 *)b_patch.p; d_ps = d_poff + np; d_pe = d_ps + np;
        }
        KdCns C;
        C.seg_contig = d_seg; C.min_depth = min_depth; C.n_patches = n_patches;
        C.best_a = (const kd_u64 *)b_best.p - alloc_lo; C.best_b = (const kd_u64 *)b_best2.p - alloc_lo; C.ins_rep = (const uint32_t *)b_hrep.p;
        C.patch_start = d_ps; C.patch_end = d_pe;
        C.g_lo = g_lo; C.g_hi = g_hi;
        if (rt.launch("k_cns_count", k_cns_count, (unsigned)n_tiles, KD_BLOCK, 0, T, C, I, (kd_u64)tile_first, (kd_u64 *)b_tilesum.p,
                      (KdTileMM *)b_tilemm.p, meta_mm()))
            return hipfail("k_cns_count");
        if (!self_scan && rt.launch("k_cns_scan", k_cns_scan, 1u, KD_BLOCK, 0, (const kd_u64 *)b_tilesum.p, (kd_u64 *)b_tileoff.p,
                                    (kd_u64)n_tiles, (const KdTileMM *)b_tilemm.p, meta_mm()))
            return hipfail("k_cns_scan");
        if (rt.launch("k_cns_emit", k_cns_emit, (unsigned)n_tiles, KD_BLOCK, 0, T, C, I, (kd_u64)tile_first, (const kd_u64 *)b_tilesum.p,
                      self_scan ? (const kd_u64 *)nullptr : (const kd_u64 *)b_tileoff.p, (const KdTileMM *)b_tilemm.p, meta_mm(),
                      (uint8_t *)b_cns.p, (uint8_t *)b_changes.p - alloc_lo, meta_coff(), n_contigs, d_poff, host_out, (kd_u64)host_cap))
            return hipfail("k_cns_emit");
        h_pstart.assign(ps, ps + n_patches);
        h_poff.assign(n_patches, ~0ULL);
        return KD_OK;
    }
    // meta: the metadata block as copied back (meta_bytes() of it)
    int consensus_collect(const void *meta) {
        const uint64_t *m = (const uint64_t *)meta;
        h_coff.assign(m, m + n_contigs + 1);
        const uint32_t *mm = reinterpret_cast<const uint32_t *>(m + n_contigs + 1);
        h_minmax.assign(mm, mm + 2 * (size_t)n_contigs);
        if (h_coff[n_contigs] > cns_cap) return fail(KD_E_INTERNAL, "consensus longer than its buffer");
        // contigs whose first site lies outside the processed tiles were not visited by k_cns_emit
        for (uint32_t c = 0; c < n_contigs; c++) {
            if (cbase[c] < cns_tile_first * KD_CNS_TILE) h_coff[c] = 0;
            else if (cbase[c] >= (cns_tile_first + cns_tiles) * KD_CNS_TILE) h_coff[c] = h_coff[n_contigs];
        }
        have_cns = true;
        return KD_OK;
    }
    int consensus_run(uint32_t min_depth, uint32_t n_patches, const uint64_t *ps, const uint64_t *pe) {
        if (!finalized) return fail(KD_E_ARG, "kd_consensus_run: call kd_finalize first");
        int rc;
        if ((rc = consensus_launch(min_depth, n_patches, ps, pe))) return rc;
        std::vector<uint64_t> down(meta_bytes() / 8, 0);
        if (rt.d2h(down.data(), meta_coff(), meta_bytes())) return hipfail("consensus: d2h");
        if (n_patches && rt.d2h(h_poff.data(), b_patch.p, (size_t)n_patches * 8)) return hipfail("consensus: d2h");
        return consensus_collect(down.data());
    }

    // change codes of G-space sites [g0, g0 + n) into dst: the part inside this context's emit interval from the (shard-local)
    // device array, zero elsewhere
    int copy_changes(uint8_t *dst, uint64_t g0, uint64_t n) {
        memset(dst, 0, (size_t)n);
        const uint64_t a = std::max(g0, g_lo), b = std::min(g0 + n, std::min<uint64_t>(g_hi, S));
        if (a < b && rt.d2h(dst + (a - g0), (uint8_t *)b_changes.p + (a - alloc_lo), (size_t)(b - a))) return 1;
        return 0;
    }

    int consensus_fetch(uint32_t contig, uint8_t *seq_out, uint64_t cap, uint64_t *len_out, uint8_t *changes,
                        uint32_t *depth_minmax, uint64_t *patch_off) {
        if (!have_cns) return fail(KD_E_ARG, "kd_consensus_fetch: call kd_consensus_run first");
        if (contig >= n_contigs) return fail(KD_E_ARG, "kd_consensus_fetch: bad contig");
        const uint64_t o0 = h_coff[contig], o1 = h_coff[contig + 1];
        if (len_out) *len_out = o1 - o0;
        if (seq_out) {
            if (o1 - o0 > cap) return fail(KD_E_ARG, "kd_consensus_fetch: buffer too small");
            if (o1 > o0 && rt.d2h(seq_out, (uint8_t *)b_cns.p + o0, o1 - o0)) return hipfail("consensus fetch: d2h");
        }
        if (changes && clen[contig] && copy_changes(changes, cbase[contig], clen[contig])) return hipfail("consensus fetch: d2h changes");
        if (depth_minmax) { depth_minmax[0] = h_minmax[2 * contig]; depth_minmax[1] = h_minmax[2 * contig + 1]; }
        if (patch_off)
            for (size_t k = 0; k < h_pstart.size(); k++) {
                const bool mine = h_pstart[k] >= cbase[contig] && h_pstart[k] < cbase[contig] + clen[contig];
                patch_off[k] = (mine && h_poff[k] != ~0ULL) ? h_poff[k] - o0 : ~0ULL;
            }
        return KD_OK;
    }

    int consensus_fetch_all(uint8_t *seq_out, uint64_t cap, uint64_t *len_out, uint64_t *contig_off, uint8_t *changes) {
        if (!have_cns) return fail(KD_E_ARG, "kd_consensus_fetch_all: call kd_consensus_run first");
        const uint64_t o0 = h_coff[0], o1 = h_coff[n_contigs];
        if (len_out) *len_out = o1 - o0;
        if (contig_off)
            for (uint32_t c = 0; c <= n_contigs; c++) contig_off[c] = h_coff[c] - o0;
        if (seq_out) {
            if (o1 - o0 > cap) return fail(KD_E_ARG, "kd_consensus_fetch_all: buffer too small");
            if (o1 > o0 && rt.d2h(seq_out, (uint8_t *)b_cns.p + o0, o1 - o0)) return hipfail("consensus fetch: d2h");
        }
        if (changes && S && copy_changes(changes, 0, S)) return hipfail("consensus fetch: d2h changes");
        return KD_OK;
    }

    // ---- the exchange row (multi-GPU, kindel_amd/shard.py: the ONE all-gather that stitches the FASTA) ----
    //   u64 row bytes | u64 0 | contig_off u64[n_contigs + 1] | depth min / max u32[2 n_contigs] | change codes of [g_lo, g_hi) | consensus bytes
    // written into a caller's DEVICE buffer by the context itself: two device-to-device copies and a one-workgroup kernel for the header
    // and the run's metadata, queued behind k_cns_emit (by kd_finish / kd_step when a row is registered: kd_set_exchange) -- the row is
    // complete when the step's last read-back returns, no host round trip of its own.  (Before: the Python side built the row with eight
    // torch operations, two pageable uploads and a blocking .item() per step: +0.10 ms on a 1/8 shard's 0.38 ms step, measured;
    // scripts/exp/exchange_ab.py.)  A row that does not fit gets its header only: row bytes > cap tells every rank so after the gather.
    uint8_t *exch_row = nullptr;
    uint64_t exch_cap = 0, exch_cns_queued = 0;
    uint64_t exch_sites() const { return std::min<uint64_t>(S, g_hi) - std::min<uint64_t>(S, g_lo); }
    uint64_t exch_fixed() const { return 16 + meta_bytes() + exch_sites(); }
    // queued behind k_cns_emit: the change codes, the first cns_bytes of the consensus (its length is not known yet: a guess) and
    // k_exchange_head (header + metadata, from the device's copy of the run's metadata block)
    int exchange_queue(uint8_t *row, uint64_t cap, uint64_t cns_bytes) {
        exch_cns_queued = 0;
        if (cap < 16 || (reinterpret_cast<uintptr_t>(row) & 7u)) return fail(KD_E_ARG, "exchange row: at least 16 bytes, 8-byte aligned");
        const uint64_t fixed = exch_fixed(), sites = exch_sites();
        if (fixed <= cap) {
            if (sites && rt.d2d(row + (fixed - sites), (const uint8_t *)b_changes.p + (std::min<uint64_t>(S, g_lo) - alloc_lo), (size_t)sites)) return hipfail("exchange: d2d");
            const uint64_t n = std::min<uint64_t>(cns_bytes, cap - fixed);
            if (n && rt.d2d(row + fixed, b_cns.p, (size_t)n)) return hipfail("exchange: d2d");
            exch_cns_queued = n;
        }
        if (rt.launch("k_exchange_head", k_exchange_head, 1u, KD_BLOCK, 0, (const kd_u64 *)meta_coff(), (const uint32_t *)meta_mm(), (const kd_u64 *)d_cbase, n_contigs,
                      (kd_u64)(cns_tile_first * KD_CNS_TILE), (kd_u64)((cns_tile_first + cns_tiles) * KD_CNS_TILE), (kd_u64)fixed, (kd_u64)cap, (kd_u64 *)row))
            return hipfail("k_exchange_head");
        return KD_OK;
    }
    // after consensus_collect: the consensus bytes the guess did not cover (net insertions beyond it: rare); waits for the row
    int exchange_rest(uint8_t *row, uint64_t cap, uint64_t *row_bytes) {
        const uint64_t total = h_coff[n_contigs], fixed = exch_fixed(), need = fixed + total;
        if (need <= cap && total > exch_cns_queued &&
            rt.d2d(row + fixed + exch_cns_queued, (const uint8_t *)b_cns.p + exch_cns_queued, (size_t)(total - exch_cns_queued)))
            return hipfail("exchange: d2d");
        if (rt.sync()) return hipfail("exchange: sync");
        if (row_bytes) *row_bytes = need;
        return KD_OK;
    }
    // on demand, after any consensus run (kd_exchange_row)
    int exchange_row(uint8_t *row, uint64_t cap, uint64_t *row_bytes) {
        if (!have_cns) return fail(KD_E_ARG, "kd_exchange_row: call kd_consensus_run first");
        if (!row) return fail(KD_E_ARG, "kd_exchange_row: no row");
        int rc;
        if ((rc = exchange_queue(row, cap, h_coff[n_contigs]))) return rc;
        return exchange_rest(row, cap, row_bytes);
    }

    // kd_finish: everything behind the pushes -- insertion reduction, consensus, read-out -- queued back to back and collected in
    // ONE host round trip: the status words (deferred reference exceptions, hash verification), the run's metadata and the
    // consensus bytes -- as many as a consensus without net insertions has; the rare rest in a second copy.  (kd_finalize +
    // kd_consensus_run + kd_consensus_fetch_all are three blocking read-backs and two more for the bytes.)
    int finish(uint32_t min_depth, uint8_t *seq_out, uint64_t cap, uint64_t *len_out, uint64_t *contig_off) {
        int rc;
        // a pinned seq_out the device can address: the consensus kernel writes the bytes there itself (round 6; the copy behind the
        // kernel was 90 us of C3's step).  Pageable memory (a numpy array of the API path) keeps the copy.
        uint8_t *zc = (seq_out && knob_zero_copy && !(reinterpret_cast<uintptr_t>(seq_out) & 3u)) ? (uint8_t *)rt.device_view(seq_out) : nullptr;
        if ((rc = finalize_launch()) || (rc = consensus_launch(min_depth, 0, nullptr, nullptr, zc, cap))) return rc;
        const size_t mb = meta_bytes();
        const uint64_t shard_sites = std::min<uint64_t>(S, g_hi) - std::min<uint64_t>(S, g_lo);
        uint64_t guess = (seq_out && !zc) ? std::min<uint64_t>(std::min<uint64_t>(cap, cns_cap), shard_sites + 4096) : 0;
        uint8_t *st = (uint8_t *)rt.stage(KDS_COUNT * 8 + mb);
        if (!st) return hipfail("kd_finish: pinned staging");
        if (exch_row && (rc = exchange_queue(exch_row, exch_cap, std::min<uint64_t>(cns_cap, shard_sites + 4096)))) return rc;
        if (rt.d2h_async(st, d_status, KDS_COUNT * 8 + mb) ||      // (status words | metadata: adjacent in device memory)
            (guess && rt.d2h_async(seq_out, b_cns.p, guess)) || rt.sync())
            return hipfail("kd_finish: d2h");
        memcpy(h_status.data(), st, KDS_COUNT * 8);
        const void *meta = st + KDS_COUNT * 8;
        if (errors_pending) {       // kd_step left the batch's error classification to this moment (defer_errors)
            errors_pending = false;
            if ((err_windowed && h_status[KDS_BAD_BASE] != 0) || h_status[KDS_ERR_READ] != ~0ULL) {
                if (rt.launch("k_errors", k_errors, 1u, KD_BLOCK, 0, err_R, tabs(), (const KdRInfo *)b_rinfo.p, n_contigs, d_status, err_windowed))
                    return hipfail("k_errors");
                if ((rc = fetch_status())) return rc;
            }
        }
        bool redone = false;
        if ((rc = finalize_check(nullptr, &redone))) return rc;
        if (redone) {     // a hash collision was repaired (never seen outside the tests): the consensus once more, read back on its own
            if ((rc = consensus_run(min_depth, 0, nullptr, nullptr))) return rc;
            guess = 0; zc = nullptr;
            if (exch_row && ((rc = exchange_queue(exch_row, exch_cap, h_coff[n_contigs])) || (rc = exchange_rest(exch_row, exch_cap, nullptr)))) return rc;     // (the row once more)
        } else if ((rc = consensus_collect(meta))) return rc;
        else if (exch_row && h_coff[n_contigs] > exch_cns_queued && (rc = exchange_rest(exch_row, exch_cap, nullptr))) return rc;
        const uint64_t o0 = h_coff[0], o1 = h_coff[n_contigs];
        if (len_out) *len_out = o1 - o0;
        if (contig_off) for (uint32_t c = 0; c <= n_contigs; c++) contig_off[c] = h_coff[c] - o0;
        if (seq_out) {
            if (o1 - o0 > cap) return fail(KD_E_ARG, "kd_finish: buffer too small");
            if (!zc && o1 > guess && rt.d2h((uint8_t *)seq_out + guess, (uint8_t *)b_cns.p + guess, o1 - guess)) return hipfail("consensus fetch: d2h");
        }
        return KD_OK;
    }

    int knob_inflate = 2;             // KD_INFLATE=2 (default since round 6: the two-pass inflate, kd_gpu_inflate2.h) / 1 (the one-pass kernel of rounds 3 - 5): which GPU inflate the device-side ingest uses
    bool knob_zero_copy = true;       // KD_ZERO_COPY=0 (tests, measurement): kd_step / kd_finish copy the FASTA behind the consensus kernel even into pinned memory
    int knob_ins_site_flags = -1;     // KD_INS_SITE_FLAGS=0 / 1 (tests, measurement): the insertion reduction's site test per event / once per site, whatever the counts
    bool knob_cold_tail = true;       // the cold records' workgroups ride in k_window's launch (kd_window.h: KdColdTail) instead of k_cold_lane's own: the memory-bound
                                      // cold work fills the launch's tail (round 5: step -1.6 % on C3, -2.4 % on C4, bit-exact; default since round 6).  KD_COLD_TAIL=0:
                                      // k_cold_lane as a launch of its own (tests keep both branches covered)

    // One step over a device-resident batch (kd_step): reset, record loop, insertion reduction and consensus queued back to back;
    // the host waits ONCE behind k_prep (the counts that size buffers and choose kernels) and ONCE at the end (finish()).  (Round 3:
    // five blocking read-backs per step.  Rounds 3 - 5 could also capture a repeated step as a hipGraph and replay it; with two host
    // round trips left per step the replay measured nothing -- C3 1.573 vs 1.576 ms -- and it was the one path that faulted on
    // hardware: removed in round 6)
    int step(const kd_batch &B, uint32_t min_depth, uint8_t *seq_out, uint64_t cap, uint64_t *len_out, uint64_t *contig_off) {
        int rc;
        defer_errors = true; errors_pending = false;
        rc = reset();
        if (!rc) rc = push_device(B);
        defer_errors = false;
        if (rc) { errors_pending = false; return rc; }
        return finish(min_depth, seq_out, cap, len_out, contig_off);
    }
};
