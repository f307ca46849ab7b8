// kd_plan.h -- k_plan_*, k_sort_*: windows -> candidate ranges -> work items; bucket sort by window.
// Part of the device code of kd_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
#include "kd_common.h"

// ---------------------------------------------------------------------------------------
// Windowed path: k_plan + k_window
// ---------------------------------------------------------------------------------------

// first index in [0,n) with rinfo[idx].gstart >= key
__device__ __forceinline__ kd_u64 kd_lower_bound(const KdRInfo *rinfo, kd_u64 n, kd_u64 key) {
    kd_u64 lo = 0, hi = n;
    while (lo < hi) {
        const kd_u64 mid = (lo + hi) >> 1;
        if ((kd_u64)rinfo[mid].gstart < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// k_plan_ranges: one thread per window w of W sites.  The candidate reads are those whose G-start
// lies in [w*W - maxspan, (w+1)*W) -- a contiguous index range because the batch is sorted -- cut
// into slices of `slice` reads (one work item each).
__global__ void __launch_bounds__(KD_BLOCK)
k_plan_ranges(const KdRInfo *rinfo, kd_u64 n_reads, uint32_t w0, uint32_t n_win, uint32_t W, uint32_t slice,
              kd_u64 *win_lo, kd_u64 *win_hi, kd_u64 *item_off, const kd_u64 *status, uint32_t H) {
    const uint32_t w = blockIdx.x * KD_BLOCK + threadIdx.x;   // local index; the window is w0 + w (shard-local planning)
    if (w >= n_win) return;
    // H (k_window: OWNERSHIP): an entry that starts in front of the window has something left for it only if it is longer than
    // H -- except for the first window of the plan, which takes such entries from its first site
    kd_u64 maxspan = status[KDS_B_MAXSPAN];
    if (w) maxspan = maxspan > H ? maxspan - H : 0;
    const kd_u64 wlo = (kd_u64)(w0 + w) * W, whi = wlo + W;
    const kd_u64 lo = kd_lower_bound(rinfo, n_reads, wlo > maxspan ? wlo - maxspan : 0);
    const kd_u64 hi = kd_lower_bound(rinfo, n_reads, whi + status[KDS_B_MAXLEAD]);  // leading clips reach back
    win_lo[w] = lo; win_hi[w] = hi;
    item_off[w] = (hi - lo + slice - 1) / slice;  // item count; k_plan_scan turns it into an offset
}

// ---- unsorted batches: bucket the regular reads by window (counting sort), so that the candidate reads of
// a window are again a contiguous range -- of the permutation `order` instead of the batch itself.
// Each thread takes RUN CONSECUTIVE entries and merges neighbours that fall into the same bin into one atomic.
// RUN = KD_SORT_RUN for the segments of long reads: they arrive in reference order, thousands per bin, and one
// atomic per entry would serialise on a handful of addresses.  RUN = 1 for the reads of an unsorted batch
// (nothing to merge; coalesced one-entry-per-lane access).
#define KD_SORT_RUN 16
// `reps` > 1: every bin has `reps` counters, a workgroup uses counter (blockIdx mod reps): the counters of a 5 Mbp genome are
// 31 KB = 488 cache lines, and 1.7 x 10^7 device-scope atomics on 488 lines took 0.6 - 0.9 ms per pass (unsorted C3), which is
// the atomic unit's rate per line, not the memory's.  Entries of one bin stay contiguous (its replicas are neighbours).
template <int RUN>
__global__ void __launch_bounds__(KD_BLOCK)
k_sort_count(const KdRInfo *rinfo, kd_u64 n_reads, uint32_t W, uint32_t *bin_cnt, uint32_t reps) {
    const kd_u64 i0 = ((kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x) * RUN;
    const uint32_t rep = blockIdx.x % reps;
    uint32_t cur = 0xffffffffu, run = 0;
    for (kd_u64 i = i0; i < i0 + RUN && i < n_reads; i++) {
        const KdRInfo ri = rinfo[i];
        if ((ri.span_cls & 3u) != KD_CLS_REG) continue;
        const uint32_t b = ri.gstart / W;
        if (b != cur) {
            if (run) atomicAdd(&bin_cnt[(kd_u64)cur * reps + rep], run);
            cur = b; run = 0;
        }
        run++;
    }
    if (run) atomicAdd(&bin_cnt[(kd_u64)cur * reps + rep], run);
}
// one workgroup: bin_off = exclusive scan of bin_cnt (n_bins + 1 entries), bin_cnt is reset to 0 (it becomes
// the fill cursor of k_sort_scatter).  Every thread scans a contiguous run of bins, the run totals are scanned over the
// workgroup with __shfl_up (kd_block_scan_incl): two barriers whatever n_bins is.
__global__ void __launch_bounds__(KD_SCAN_WIDE)
k_sort_scan(uint32_t *bin_cnt, kd_u64 *bin_off, uint32_t n_bins, kd_u64 *status) {
    __shared__ kd_u64 s_wave[KD_SCAN_WIDE / KD_WAVE];
    const uint32_t t = threadIdx.x;
    // (a bucket-sorted pass follows: k_window's work queue starts empty -- a batch's second pass finds the first one's counts)
    if (t == 0) { status[KDS_WQ_TICKET] = 0; status[KDS_WQ_PUB] = 0; status[KDS_WQ_HOT] = 0; status[KDS_WQ_LEFT] = 0; }
    const uint32_t per = (n_bins + KD_SCAN_WIDE - 1) / KD_SCAN_WIDE;
    const uint32_t b0 = t * per < n_bins ? t * per : n_bins, b1 = b0 + per < n_bins ? b0 + per : n_bins;
    kd_u64 mine = 0;
    for (uint32_t b = b0; b < b1; b++) mine += bin_cnt[b];
    kd_u64 total;
    kd_u64 o = kd_block_scan_incl_wide(mine, s_wave, total) - mine;
    for (uint32_t b = b0; b < b1; b++) { const kd_u64 v = bin_cnt[b]; bin_off[b] = o; bin_cnt[b] = 0; o += v; }
    if (t == 0) bin_off[n_bins] = total;
}
// The READS of an unsorted batch scattered PHYSICALLY into window order (round 3): footprint record, offsets and CIGAR word
// count of every regular read land at its sorted position, so that k_window walks the bucket-sorted batch like a sorted one --
// coalesced footprint / offset loads, no permutation to chase (it read rinfo[order[j]], seq_off[order[j]], ... before: 2.1 x the
// sorted kernel's time).  The packed bases and CIGAR words stay where they are (their offsets travel).
__global__ void __launch_bounds__(KD_BLOCK)
k_sort_scatter_reads(const KdRInfo *rinfo, KdReads rd, uint32_t W, uint32_t *bin_fill, const kd_u64 *bin_off, uint32_t reps,
                     KdSortRec *rec) {
    const kd_u64 i = (kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x;
    if (i >= rd.n) return;
    const KdRInfo ri = rinfo[i];
    if ((ri.span_cls & 3u) != KD_CLS_REG) return;
    const kd_u64 slot = (kd_u64)(ri.gstart / W) * reps + blockIdx.x % reps;
    const kd_u64 at = bin_off[slot] + atomicAdd(&bin_fill[slot], 1u);
    KdSortRec r; r.ri = ri; r.seq_off = rd.seq_off[i]; r.cig_off = rd.cig_off[i];
    rec[at] = r;          // one 32-byte sector per read
}

// The same two passes with the bin counters PRIVATE to a workgroup in LDS (round 3; an unsorted batch whose bin table fits:
// n_bins * 4 bytes of LDS, ~ 17 Mbp of reference at the default window) and NO global atomics.  A workgroup of KD_SORT_BLOCK
// threads takes one contiguous chunk of the batch; its reads hit LDS counters (random addresses cost an LDS atomic, not a
// device-scope one) and the workgroup leaves its counts as ONE ROW of a (workgroups x bins) matrix -- plain coalesced stores.
// The slot range of (workgroup g, bin b) starts at  bin_off[b] + sum of rows[g' < g][b]:  a column scan in two small kernels
// (k_sort_colscan: KD_SORT_SEG rows per thread, all loads in flight together; k_sort_colscan2: the segment totals of a bin),
// then k_sort_scatter_lds reads its row of starts and hands the slots out from LDS: the reads of one chunk that fall into one
// window land next to each other.  (Until the k_prep experiments of round 3 the workgroups reserved their ranges with one
// RETURNING atomic per bin and workgroup on a table of adjacent words: 5.7 x 10^6 atomics, 16 000 per 128-byte line, which
// serialise per line -- profiles/r03_kprep_experiments.json.)
#define KD_SORT_BLOCK 1024
#define KD_SORT_SEG 32
__global__ void __launch_bounds__(KD_SORT_BLOCK)
k_sort_count_lds(const KdRInfo *rinfo, kd_u64 n_reads, kd_u64 chunk, uint32_t W, uint32_t n_bins, uint32_t *rows) {
    KD_DYN_SHARED(uint32_t, s_bins);
    for (uint32_t b = threadIdx.x; b < n_bins; b += KD_SORT_BLOCK) s_bins[b] = 0;
    __syncthreads();
    const kd_u64 c0 = (kd_u64)blockIdx.x * chunk, c1 = c0 + chunk < n_reads ? c0 + chunk : n_reads;
    for (kd_u64 i = c0 + threadIdx.x; i < c1; i += KD_SORT_BLOCK) {
        const KdRInfo ri = rinfo[i];
        if ((ri.span_cls & 3u) == KD_CLS_REG) atomicAdd(&s_bins[ri.gstart / W], 1u);
    }
    __syncthreads();
    uint32_t *row = rows + (kd_u64)blockIdx.x * n_bins;
    for (uint32_t b = threadIdx.x; b < n_bins; b += KD_SORT_BLOCK) row[b] = s_bins[b];
}
// thread (segment s, bin b): rows[s * SEG + k][b], k < SEG, -> their exclusive prefix sums in place, seg_tot[s][b] = their sum
__global__ void __launch_bounds__(KD_BLOCK)
k_sort_colscan(uint32_t *rows, uint32_t n_rows, uint32_t n_bins, uint32_t *seg_tot) {
    const kd_u64 x = (kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x;
    const uint32_t n_seg = (n_rows + KD_SORT_SEG - 1) / KD_SORT_SEG;
    if (x >= (kd_u64)n_seg * n_bins) return;
    const uint32_t s = (uint32_t)(x / n_bins), b = (uint32_t)(x % n_bins);
    const uint32_t r0 = s * KD_SORT_SEG;
    uint32_t v[KD_SORT_SEG];
#pragma unroll
    for (uint32_t k = 0; k < KD_SORT_SEG; k++) v[k] = r0 + k < n_rows ? rows[(kd_u64)(r0 + k) * n_bins + b] : 0u;
    uint32_t run = 0;
#pragma unroll
    for (uint32_t k = 0; k < KD_SORT_SEG; k++) {
        if (r0 + k < n_rows) rows[(kd_u64)(r0 + k) * n_bins + b] = run;
        run += v[k];
    }
    seg_tot[(kd_u64)s * n_bins + b] = run;
}
// thread per bin: the segment totals of the bin -> their exclusive prefix sums in place, bin_cnt[b] = the bin's reads
__global__ void __launch_bounds__(KD_BLOCK)
k_sort_colscan2(uint32_t *seg_tot, uint32_t n_seg, uint32_t n_bins, uint32_t *bin_cnt) {
    const uint32_t b = blockIdx.x * KD_BLOCK + threadIdx.x;
    if (b >= n_bins) return;
    uint32_t run = 0;
    for (uint32_t s = 0; s < n_seg; s++) { const uint32_t v = seg_tot[(kd_u64)s * n_bins + b]; seg_tot[(kd_u64)s * n_bins + b] = run; run += v; }
    bin_cnt[b] = run;
}
__global__ void __launch_bounds__(KD_SORT_BLOCK)
k_sort_scatter_lds(const KdRInfo *rinfo, KdReads rd, kd_u64 chunk, uint32_t W, uint32_t n_bins, const uint32_t *rows,
                   const uint32_t *seg_tot, const kd_u64 *bin_off, KdSortRec *rec) {
    KD_DYN_SHARED(uint32_t, s_bins);
    // the first slot of every bin for this workgroup's chunk (sorted positions are < n_reads < 2^32)
    const uint32_t *row = rows + (kd_u64)blockIdx.x * n_bins, *seg = seg_tot + (kd_u64)(blockIdx.x / KD_SORT_SEG) * n_bins;
    for (uint32_t b = threadIdx.x; b < n_bins; b += KD_SORT_BLOCK) s_bins[b] = (uint32_t)bin_off[b] + seg[b] + row[b];
    __syncthreads();
    const kd_u64 c0 = (kd_u64)blockIdx.x * chunk, c1 = c0 + chunk < rd.n ? c0 + chunk : rd.n;
    for (kd_u64 i = c0 + threadIdx.x; i < c1; i += KD_SORT_BLOCK) {
        const KdRInfo ri = rinfo[i];
        if ((ri.span_cls & 3u) != KD_CLS_REG) continue;
        const uint32_t at = atomicAdd(&s_bins[ri.gstart / W], 1u);
        KdSortRec r; r.ri = ri; r.seq_off = rd.seq_off[i]; r.cig_off = rd.cig_off[i];
        rec[at] = r;          // one 32-byte sector per read
    }
}

template <int RUN>
__global__ void __launch_bounds__(KD_BLOCK)
k_sort_scatter(const KdRInfo *rinfo, kd_u64 n_reads, uint32_t W, uint32_t *bin_fill, const kd_u64 *bin_off,
               uint32_t *order) {
    const kd_u64 i0 = ((kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x) * RUN;
    const kd_u64 i1 = i0 + RUN < n_reads ? i0 + RUN : n_reads;
    // maximal runs of consecutive regular entries of one bin: one reservation, consecutive slots
    kd_u64 i = i0;
    while (i < i1) {
        const KdRInfo ri = rinfo[i];
        if ((ri.span_cls & 3u) != KD_CLS_REG) { i++; continue; }
        const uint32_t b = ri.gstart / W;
        kd_u64 j = i + 1;
        uint32_t m = 1;
        for (; j < i1; j++) {
            const KdRInfo rj = rinfo[j];
            if ((rj.span_cls & 3u) != KD_CLS_REG) continue;   // skipped entries do not break a run
            if (rj.gstart / W != b) break;
            m++;
        }
        kd_u64 at = bin_off[b] + atomicAdd(&bin_fill[b], m);
        order[at++] = (uint32_t)i;
        for (kd_u64 x = i + 1; x < j; x++)   // (i, j): entries of bin b and skipped ones
            if ((rinfo[x].span_cls & 3u) == KD_CLS_REG) order[at++] = (uint32_t)x;
        i = j;
    }
}
// (Round 6 measured the three passes as ONE workgroup with its counters in LDS for the 17 699 rows of C5 -- one launch instead of a
// memset and three: 0.025 ms against 3 x 0.0065, a single workgroup's chain of round trips is longer than two more launches; dropped.)
// candidate range of window w0 + w in `order`: whole bins covering [wlo - maxspan, whi + maxlead)
__global__ void __launch_bounds__(KD_BLOCK)
k_plan_ranges_sorted(const kd_u64 *bin_off, uint32_t n_bins, uint32_t w0, uint32_t n_win, uint32_t W, uint32_t slice,
                     kd_u64 *win_lo, kd_u64 *win_hi, kd_u64 *item_off, const kd_u64 *status, uint32_t span_slot, uint32_t reps) {
    const uint32_t w = blockIdx.x * KD_BLOCK + threadIdx.x;
    if (w >= n_win) return;
    const kd_u64 wlo = (kd_u64)(w0 + w) * W, whi = wlo + W;
    const kd_u64 maxspan = status[span_slot], maxlead = status[KDS_B_MAXLEAD];   // span_slot: KDS_B_MAXSPAN / KDS_B_MAXSEGSPAN
    const kd_u64 blo = (wlo > maxspan ? wlo - maxspan : 0) / W;
    kd_u64 bhi = (whi + maxlead + W - 1) / W;   // exclusive
    if (bhi > n_bins) bhi = n_bins;
    const kd_u64 lo = bin_off[(blo < n_bins ? blo : n_bins) * reps], hi = bin_off[bhi * reps];   // (bin b = counters b * reps ..)
    win_lo[w] = lo; win_hi[w] = hi;
    item_off[w] = (hi - lo + slice - 1) / slice;
}

// k_plan_scan: one workgroup, in-place exclusive scan of the per-window item counts (contiguous run per thread, run totals
// scanned with __shfl_up), and the work item -> window table on the way (k_window then needs one load, not a binary search
// over item_off, to find the window of the item it dequeued).
__global__ void __launch_bounds__(KD_SCAN_WIDE)
k_plan_scan(kd_u64 *item_off, uint32_t n_win, uint32_t *item_win, kd_u64 cap, kd_u64 *status) {
    __shared__ kd_u64 s_wave[KD_SCAN_WIDE / KD_WAVE];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n_win + KD_SCAN_WIDE - 1) / KD_SCAN_WIDE;
    const uint32_t w0 = t * per < n_win ? t * per : n_win, w1 = w0 + per < n_win ? w0 + per : n_win;
    kd_u64 mine = 0;
    for (uint32_t w = w0; w < w1; w++) mine += item_off[w];
    kd_u64 total;
    kd_u64 o = kd_block_scan_incl_wide(mine, s_wave, total) - mine;
    for (uint32_t w = w0; w < w1; w++) {
        const kd_u64 v = item_off[w];
        item_off[w] = o;
        if (item_win) {   // (NULL: many more windows than one workgroup should serve -- k_plan_items does it, k_strip's planning)
            for (kd_u64 it = o; it < o + v; it++) {
                if (it < cap) item_win[it] = w;
                else status[KDS_INTERNAL] = 1;
            }
        }
        o += v;
    }
    if (t == 0) { item_off[n_win] = total; status[KDS_TOTAL_ITEMS] = total; status[KDS_NEXT_ITEM] = 0; }
    if (t < 8) status[KDS_QUEUE0 + t * KDS_STRIDE] = 0;   // k_strip's work queues
}

// k_plan_items: work item -> window table with one thread per window (after a k_plan_scan that was given no table to fill)
__global__ void __launch_bounds__(KD_BLOCK)
k_plan_items(const kd_u64 *item_off, uint32_t n_win, uint32_t *item_win, kd_u64 cap, kd_u64 *status) {
    const uint32_t w = blockIdx.x * KD_BLOCK + threadIdx.x;
    if (w >= n_win) return;
    for (kd_u64 it = item_off[w]; it < item_off[w + 1]; it++) {
        if (it < cap) item_win[it] = w;
        else status[KDS_INTERNAL] = 1;
    }
}

// k_reset (kd_reset): the status words and the per-contig first-record / first-error state in one launch.
__global__ void __launch_bounds__(KD_BLOCK)
k_reset(kd_u64 *status, kd_u64 *first_idx, kd_u64 *err_first, uint32_t *err_code, uint32_t n_contigs) {
    const uint32_t i = blockIdx.x * KD_BLOCK + threadIdx.x;
    if (i < KDS_COUNT) status[i] = i == KDS_ERR_READ ? ~0ULL : 0ULL;
    if (i < n_contigs) { first_idx[i] = ~0ULL; err_first[i] = ~0ULL; err_code[i] = 0u; }
}
