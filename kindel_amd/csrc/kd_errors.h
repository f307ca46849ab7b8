// kd_errors.h -- k_errors: error CLASSIFICATION of a batch (which read, which reference exception); writes no table.
// Part of the device code of kd_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
#include "kd_common.h"

// kd_diagnose_contig (k_errors): one thread per contig re-walks the contig's first failing read of this batch serially, in the reference's
// own statement order, to decide WHICH exception the reference raises (KeyError vs IndexError vs RuntimeError).
// Error classification only -- it writes no table.  (A contig's first failing read is final once the batch that
// contains it has been pushed: later batches only hold larger read indices.)
__device__ __forceinline__ void kd_diagnose_contig(const KdReads &rd, const KdTabs &T, uint32_t cdx) {
    const kd_u64 gidx = T.err_first[cdx];
    if (gidx == ~0ULL || gidx < rd.base_index || gidx >= rd.base_index + rd.n) return;
    const kd_u64 i = gidx - rd.base_index;
    const int64_t sl = rd.seq_len[i];
    const uint32_t nc = rd.n_cig[i];
    const int64_t L = T.contig_len[rd.contig[i]];
    const uint8_t *seq = rd.seq4 + rd.seq_off[i];
    const uint32_t *cg = rd.cigar + rd.cig_off[i];
    kd_u64 code = 8;  // KD_E_INTERNAL magnitude: flagged but no exception reproduced
    if (nc == 0) { T.err_code[cdx] = 3u; return; }
    int64_t r = rd.pos0[i], q = 0;
    for (uint32_t k = 0; k < nc && code == 8; k++) {
        const int64_t len = cg[k] >> 4;
        const uint32_t op = cg[k] & 15u;
        if (op == 0 || op == 7 || op == 8) {
            for (int64_t j = 0; j < len; j++) {
                if (q >= sl) { code = 2; break; }
                int64_t idx = r < 0 ? r + L : r;
                if (idx < 0 || idx >= L) { code = 2; break; }
                if (kd_chan(kd_nib(seq, q)) == 7u) { code = 1; break; }
                r++; q++;
            }
        } else if (op == 1) {
            int64_t idx = r < 0 ? r + L + 1 : r;
            if (idx < 0 || idx > L) { code = 2; break; }
            q += len;
        } else if (op == 2) {
            for (int64_t j = 0; j < len; j++) {
                int64_t idx = r + j < 0 ? r + j + L + 1 : r + j;
                if (idx < 0 || idx > L) { code = 2; break; }
            }
            r += len;
        } else if (op == 4) {
            if (k == 0) {
                int64_t idx = r < 0 ? r + L + 1 : r;
                if (idx < 0 || idx > L) { code = 2; break; }
                for (int64_t j = 0; j < len; j++) {
                    if (j >= sl) { code = 2; break; }
                    const int64_t rel = r - len + j;
                    if (rel >= 0) {
                        if (rel >= L) { code = 2; break; }
                        if (kd_chan(kd_nib(seq, j)) == 7u) { code = 1; break; }
                    }
                }
                q += len;
            } else {
                int64_t idx = r - 1 < 0 ? r - 1 + L + 1 : r - 1;
                if (idx < 0 || idx > L) { code = 2; break; }
                for (int64_t j = 0; j < len; j++) {
                    if (q >= sl) { code = 2; break; }
                    if (r < L) {
                        int64_t wi = r < 0 ? r + L : r;
                        if (wi < 0 || wi >= L) { code = 2; break; }
                        if (kd_chan(kd_nib(seq, q)) == 7u) { code = 1; break; }
                        r++; q++;
                    }
                }
            }
        }
    }
    T.err_code[cdx] = (uint32_t)code;
}

// Rare path: k_window saw a base outside A,C,G,T,N.  One workgroup walks the regular reads of the
// batch and records the first offender (atomicMin of the read index), for kd_diagnose_contig to classify.
__device__ __forceinline__ void kd_find_bad_base(const KdReads &rd, const KdTabs &T, const KdRInfo *rinfo, kd_u64 *status) {
    for (kd_u64 i = threadIdx.x; i < rd.n; i += KD_BLOCK) {
        const uint32_t cls_i = rinfo[i].span_cls & 3u;
        if (cls_i != KD_CLS_REG && cls_i != KD_CLS_LONG) continue;   // regular reads, short and long
        if (rd.base_index + i >= T.err_first[rd.contig[i]]) continue;   // the contig already has an earlier failing read
        const uint8_t *seq = rd.seq4 + rd.seq_off[i];
        const uint32_t *cg = rd.cigar + rd.cig_off[i];
        const uint32_t nc = rd.n_cig[i];
        const int64_t L = T.contig_len[rd.contig[i]];
        int64_t q = 0, r = rd.pos0[i];
        bool found = false;
        for (uint32_t k = 0; k < nc && !found; k++) {
            const int64_t len = cg[k] >> 4;
            const uint32_t op = cg[k] & 15u;
            int64_t x0 = 0, x1 = 0;  // query bases the reference looks up in a weight dict
            if (op == 0 || op == 7 || op == 8) { x0 = q; x1 = q + len; q += len; r += len; }
            else if (op == 1) q += len;
            else if (op == 2) r += len;
            else if (op == 4) {
                if (k == 0) { x0 = r < len ? len - r : 0; x1 = len; q += len; }
                else { const int64_t n_adv = r < L ? (len < L - r ? len : L - r) : 0; x0 = q; x1 = q + n_adv; k = nc; }
            }
            for (int64_t x = x0; x < x1; x++)
                if (kd_chan(kd_nib(seq, x)) == 7u) { found = true; break; }
        }
        if (found) kd_flag_error(T, status, rd.contig[i], rd.base_index + i);
    }
}

// k_errors -- the error CLASSIFICATION of a batch, one workgroup, rare path: which read a bad base belongs to
// (kd_find_bad_base, when a window work item saw one), then per contig which exception the reference raises for the contig's
// first failing read (kd_diagnose_contig).  It leaves after two loads when nothing was flagged, which is why it rides as the
// LAST WORKGROUP of the batch's last kernel (k_cold_lane) instead of being two dispatches of its own (round 3: k_find_bad_base
// + k_diagnose, 3.7 us each on every step); a batch without clipped / inserted reads launches it alone.
__device__ __forceinline__ void kd_errors(const KdReads &rd, const KdTabs &T, const KdRInfo *rinfo, uint32_t n_contigs, kd_u64 *status,
                                          bool windowed) {
    const bool bad = windowed && status[KDS_BAD_BASE] != 0;
    if (!bad && status[KDS_ERR_READ] == ~0ULL) return;          // (uniform: every thread reads the same words)
    if (bad) kd_find_bad_base(rd, T, rinfo, status);
    __threadfence();
    __syncthreads();
    if (*(volatile kd_u64 *)&status[KDS_ERR_READ] == ~0ULL) return;
    for (uint32_t c = threadIdx.x; c < n_contigs; c += KD_BLOCK) kd_diagnose_contig(rd, T, c);
}
__global__ void __launch_bounds__(KD_BLOCK)
k_errors(KdReads rd, KdTabs T, const KdRInfo *rinfo, uint32_t n_contigs, kd_u64 *status, uint32_t windowed) {
    kd_errors(rd, T, rinfo, n_contigs, status, windowed != 0);
}
