// kd_long.h -- long-CIGAR reads (> KD_PREP_MAX_OPS words: long-read aligners, thousands of ops per read):
// k_prep_long (validate, footprint, counts), k_long_reduce (slots), k_long_expand (rows, insertion events, clips).
// Part of the device code of kd_kernels.h (included from there, in order; not a stand-alone header).
#pragma once
#include "kd_common.h"

// A long read's CIGAR is an op every few bases (ONT: ~7), so anything that walks it op by op with one lane per read -- or
// per run of ops -- spends its time decoding, diverging and masking (round 2's segment pass: 86 lane-instructions per base).
// Round 3 turns the problem around: a regular long read is EXPANDED once into a ROW, one 4-bit symbol per reference site
// of its footprint (base / deleted / nothing, + "an insertion precedes this site"), and k_window<ROWS> tallies rows as plain
// runs: 8 sites per dword at compile-time counter offsets, every lane live, no CIGAR in sight.  The kernels here are
// TILE loops: 256 consecutive ops per step, thread = op (coalesced CIGAR loads), a workgroup scan of the reference / query
// advances gives every op its start coordinates.

// Inclusive scan of TWO 64-bit values per thread over the workgroup (the two share their barriers); s_wave: [2 * KD_WAVES_PER_BLOCK]
__device__ __forceinline__ void kd_block_scan_incl2(kd_u64 &a, kd_u64 &b, kd_u64 *s_wave, kd_u64 &tot_a, kd_u64 &tot_b) {
    const uint32_t lane = threadIdx.x & (KD_WAVE - 1), wave = threadIdx.x / KD_WAVE;
#pragma unroll
    for (uint32_t d = 1; d < KD_WAVE; d <<= 1) {
        const kd_u64 ta = kd_shfl_up64(a, d), tb = kd_shfl_up64(b, d);
        if (lane >= d) { a += ta; b += tb; }
    }
    if (lane == KD_WAVE - 1) { s_wave[wave] = a; s_wave[KD_WAVES_PER_BLOCK + wave] = b; }
    __syncthreads();
    kd_u64 oa = 0, ob = 0, xa = 0, xb = 0;
#pragma unroll
    for (uint32_t w = 0; w < KD_WAVES_PER_BLOCK; w++) {
        const kd_u64 va = s_wave[w], vb = s_wave[KD_WAVES_PER_BLOCK + w];
        if (w < wave) { oa += va; ob += vb; }
        xa += va; xb += vb;
    }
    tot_a = xa; tot_b = xb;
    a += oa; b += ob;
    __syncthreads();   // s_wave may be reused by the caller
}

// One WAVEFRONT per long read, tiles of 64 ops (lane = op): the scans are DPP adds inside the wavefront, the running
// coordinates are carried in registers -- no barrier anywhere, four reads per workgroup.  (A workgroup per read with 256-op
// tiles and __shfl_up scans across the workgroup was 0.37 ms for the validation pass alone on the long-read bench: 48
// ds_bpermute per wavefront and tile, two barriers.)
// An op of 2^23 bases or more makes the read irregular (k_pileup_wave walks it exactly): a tile's advances then fit 32 bits.
#define KD_LONG_MAX_OP (1u << 23)
#define KD_LONG_SEQ_LDS 4096u   // bytes of query bases k_long_expand copies into LDS per wavefront and tile (8192 bases)

// reference / query advance of one CIGAR word as the scans see it (a non-first S moves nothing here: for a regular read it
// is the last op that touches r, and its own reach is kept apart as the trailing clip)
struct KdAdv { uint32_t r, q; };
__device__ __forceinline__ KdAdv kd_op_advance(uint32_t w, uint32_t k) {
    const uint32_t op = w & 15u;
    const uint32_t lens = w < (KD_LONG_MAX_OP << 4) ? w >> 4 : 0u;      // the length when it is below KD_LONG_MAX_OP
    const bool m = ((0x181u >> op) & 1u) != 0;                          // M, =, X
    KdAdv a;      // (bitwise on purpose: selects, no branches)
    a.r = (m | (op == 2)) ? lens : 0u;
    a.q = (m | (op == 1) | ((op == 4) & (k == 0))) ? lens : 0u;
    return a;
}

// k_long_order: the long reads LONGEST FIRST (round 5) -- order[x] = the x-th read k_prep_long and k_long_expand start.  Their
// wavefronts each take one read, 2 to 30 kilobases on a long-read run, and a launch ends with its last read: in list order a
// 30-kilobase read that starts late is the kernel's tail; longest first the long ones start at once and the short ones fill the
// gaps (longest-processing-time-first list scheduling).  One workgroup: a counting sort over 128 length classes (CIGAR words, two
// mantissa bits per power of two), descending.
#define KD_LONG_ORDER_BLOCK 1024
#define KD_LONG_ORDER_PER 16u
#define KD_LONG_ORDER_MAX (1u << 20)     // (more long reads than this: list order -- a tail of one read no longer shows)
__global__ void __launch_bounds__(KD_LONG_ORDER_BLOCK)
k_long_order(KdReads rd, const uint32_t *long_list, uint32_t n_long, uint32_t *order) {
    __shared__ uint32_t s_cnt[128];
    const uint32_t t = threadIdx.x;
    if (t < 128) s_cnt[t] = 0;
    __syncthreads();
    auto cls = [](uint32_t v) -> uint32_t {
        if (v < 4u) return 127u - v;
        const uint32_t lg = 31u - (uint32_t)__builtin_clz(v);
        return 127u - ((lg << 2) | ((v >> (lg - 2u)) & 3u));      // 0 = the longest class
    };
    // pass 1: a read's class from its CIGAR word count (two dependent loads per read), kept in the second half of `order`
    // (2 x n_long words) for pass 2.  (round 6) KD_LONG_ORDER_PER reads per thread and turn, their loads all in flight together:
    // the kernel is ONE workgroup, so its time is the number of dependent memory round trips it makes -- four reads per turn were
    // ten round trips on C5's 17 699 long reads, and the 128 classes were scanned by one thread (128 dependent LDS
    // read-modify-writes): 20 us together.
    uint32_t *cl = order + n_long;
    for (uint32_t b0 = t; b0 < n_long; b0 += KD_LONG_ORDER_PER * KD_LONG_ORDER_BLOCK) {
        uint32_t i_[KD_LONG_ORDER_PER], n_[KD_LONG_ORDER_PER];
#pragma unroll
        for (uint32_t u = 0; u < KD_LONG_ORDER_PER; u++) { const uint32_t b = b0 + u * KD_LONG_ORDER_BLOCK; i_[u] = b < n_long ? long_list[b] : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < KD_LONG_ORDER_PER; u++) { const uint32_t b = b0 + u * KD_LONG_ORDER_BLOCK; n_[u] = b < n_long ? rd.n_cig[i_[u]] : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < KD_LONG_ORDER_PER; u++) {
            const uint32_t b = b0 + u * KD_LONG_ORDER_BLOCK;
            if (b < n_long) { const uint32_t c = cls(n_[u]); cl[b] = c; atomicAdd(&s_cnt[c], 1u); }
        }
        // (one atomic per class a wavefront holds -- ballot loops over the classes present -- measured: 0.017 -> 0.092 ms, dropped)
    }
    __syncthreads();
    if (t < KD_WAVE) {      // exclusive scan of the 128 class counts: lane t takes classes 2t and 2t + 1
        const uint32_t v0 = s_cnt[2u * t], v1 = s_cnt[2u * t + 1u];
        const uint32_t ex = kd_wave_scan_add(v0 + v1) - (v0 + v1);
        s_cnt[2u * t] = ex; s_cnt[2u * t + 1u] = ex + v0;
    }
    __syncthreads();
    // (slots of a class are handed out in any order: which of two reads of one length class starts first does not matter; a thread
    // reads back the classes it wrote itself)
    for (uint32_t b0 = t; b0 < n_long; b0 += KD_LONG_ORDER_PER * KD_LONG_ORDER_BLOCK) {
        uint32_t c_[KD_LONG_ORDER_PER];
#pragma unroll
        for (uint32_t u = 0; u < KD_LONG_ORDER_PER; u++) { const uint32_t b = b0 + u * KD_LONG_ORDER_BLOCK; c_[u] = b < n_long ? cl[b] : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < KD_LONG_ORDER_PER; u++) {
            const uint32_t b = b0 + u * KD_LONG_ORDER_BLOCK;
            if (b < n_long) order[atomicAdd(&s_cnt[c_[u]], 1u)] = b;
        }
    }
}

// k_prep_long: the regularity rules of kd_scan_cigar (kd_prep.h) applied op-parallel: every op checks itself against its own
// start coordinates; what the read leaves behind is ONE record (KdLongAcc) and its footprint entry -- k_long_reduce turns the
// records into slots.
// (round 5: ONE wavefront per workgroup here and in k_long_expand -- KD_LONG_BLOCK.  A workgroup of four wavefronts = four reads
// keeps its slots until its LONGEST read is done, and long reads are anything from 2 to 30 kilobases: the maximum of four
// lengths is 1.6 x their mean, three of four wavefront slots idle for the difference.  Neither kernel has a workgroup barrier.)
#define KD_LONG_BLOCK KD_WAVE
#define KD_LONG_WAVES (KD_LONG_BLOCK / KD_WAVE)
__global__ void __launch_bounds__(KD_LONG_BLOCK)
k_prep_long(KdReads rd, KdTabs T, KdRInfo *rinfo, const uint32_t *long_list, uint32_t n_long, KdLongAcc *long_acc, const uint32_t *order) {
    __shared__ kd_u64 s_acc[KD_LONG_WAVES][4];       // aligned, walked, n_ins, ins_bases
    const uint32_t lane = threadIdx.x & (KD_WAVE - 1), wave = threadIdx.x / KD_WAVE;
    const uint32_t x = blockIdx.x * KD_LONG_WAVES + wave;
    if (x >= n_long) return;
    const uint32_t b = order ? order[x] : x;         // (k_long_order: longest first)
    const kd_u64 i = long_list[b];
    const uint32_t c = rd.contig[i];
    // (wave-uniform values, said so: the per-tile limits below are then scalar arithmetic)
    const int64_t L = (int64_t)kd_readfirstlane64((kd_u64)T.contig_len[c]);
    const int64_t pos0 = (int64_t)kd_readfirstlane64((kd_u64)(int64_t)rd.pos0[i]);
    const int64_t sl = (int64_t)kd_readfirstlane64((kd_u64)rd.seq_len[i]);
    const uint32_t nc = kd_readfirstlane(rd.n_cig[i]);
    const uint32_t *cg = rd.cigar + rd.cig_off[i];
    if (lane < 4) s_acc[wave][lane] = 0;
    kd_u64 aligned = 0, walked = 0, insb = 0;
    uint32_t n_ins = 0;
    bool bad = false, cold = false;
    uint32_t first_nfs = 0xffffffffu, last_rel = 0;
    int64_t c_r = pos0, c_q = 0;      // coordinates in front of the tile (wave-uniform)
    const int32_t sl_c = (int32_t)(sl < (int64_t)(1 << 28) ? sl : (int64_t)(1 << 28));     // (an op's length is below 2^28)
    uint32_t w_nxt = lane < nc ? cg[lane] : 15u, w_nxt2 = lane + KD_WAVE < nc ? cg[lane + KD_WAVE] : 15u;      // (op 15, length 0: moves nothing)
    // (round 6) The rules are applied BRANCH-FREE on 32-bit tile-relative coordinates: an op's start is (c_r + rr, c_q + qq) with
    // 0 <= rr, qq < 2^29 (64 advances below 2^23), an op's length is below 2^28, and what they are compared with -- the contig's and
    // the read's ends seen from the tile's origin, L - c_r and sl - c_q -- is clamped to [-2, 2^30] once per tile: every comparison
    // below has the same outcome as on the 64-bit coordinates (a limit of -2 or less is "behind everything", 2^30 or more "never
    // reached": `rr + len > lim`, `rr > lim + 1`, `min(len, lim - rr)` all agree).  The first version walked the four op kinds as
    // divergent branches of 64-bit arithmetic: ~200 instructions per tile of 64 ops, the kernel's whole time (0.094 ms on C5).
    for (uint32_t base = 0; base < nc; base += KD_WAVE) {
        const uint32_t k = base + lane;
        const uint32_t w = w_nxt;
        w_nxt = w_nxt2;
        { const uint32_t kn = k + 2u * KD_WAVE; w_nxt2 = kn < nc ? cg[kn] : 15u; }   // the words two tiles ahead are in flight during this one
        const KdAdv adv = kd_op_advance(w, k);
        const uint32_t ra = adv.r, qa = adv.q;
        const uint32_t ir = kd_wave_scan_add(ra), iq = kd_wave_scan_add(qa);
        const int64_t dL = L - c_r, dQ = sl - c_q;
        const int32_t lim_r = (int32_t)(dL < -2 ? -2 : dL > (int64_t)(1 << 30) ? (int64_t)(1 << 30) : dL);
        const int32_t lim_q = (int32_t)(dQ < -2 ? -2 : dQ > (int64_t)(1 << 30) ? (int64_t)(1 << 30) : dQ);
        const int32_t rr = (int32_t)(ir - ra), qq = (int32_t)(iq - qa);
        const int32_t len = (int32_t)(w >> 4);
        const uint32_t op = w & 15u;
        const bool is_m = op == 0 || op == 7 || op == 8, is_i = op == 1, is_d = op == 2, is_s = op == 4;
        const bool rel = is_m || is_i || is_d || is_s;
        bad |= rel && (uint32_t)len >= KD_LONG_MAX_OP;
        bad |= is_m && (rr + len > lim_r || qq + len > lim_q);
        bad |= is_i && rr > lim_r;
        bad |= is_d && rr + len > lim_r + 1;
        cold |= is_i || is_s;
        aligned += is_m ? (kd_u64)(uint32_t)len : 0ULL;
        walked += rel ? (kd_u64)(uint32_t)len : 0ULL;
        {   // an insertion's bases: seq[q : q + len], clamped like a Python slice
            const int32_t q0 = qq < lim_q ? qq : lim_q, q1 = qq + len < lim_q ? qq + len : lim_q;
            n_ins += is_i ? 1u : 0u;
            insb += is_i ? (kd_u64)(uint32_t)(q1 - q0) : 0ULL;
        }
        if (rel && !(is_s && k == 0)) last_rel = k;
        if (kd_ballot(is_s) != 0ULL) {      // (wave-uniform: a read's first and last tile, if any)
            if (is_s && k == 0) bad |= rr > lim_r || len > sl_c;
            if (is_s && k != 0) {
                first_nfs = k < first_nfs ? k : first_nfs;
                bad |= rr > lim_r + 1;        // clip_starts[r - 1] must exist (kindel.py:75)
                const int32_t n_adv = rr < lim_r ? (len < lim_r - rr ? len : lim_r - rr) : 0;
                bad |= n_adv > lim_q - qq || (len > n_adv && qq + n_adv >= lim_q);
            }
        }
        c_r += (int64_t)kd_readlane(ir, KD_WAVE - 1); c_q += (int64_t)kd_readlane(iq, KD_WAVE - 1);
    }
    KD_WAVE_SYNC();
    if (aligned) atomicAdd(&s_acc[wave][0], aligned);
    if (walked) atomicAdd(&s_acc[wave][1], walked);
    if (n_ins) atomicAdd(&s_acc[wave][2], (kd_u64)n_ins);
    if (insb) atomicAdd(&s_acc[wave][3], insb);
    const bool any_bad = kd_ballot(bad) != 0ULL, any_cold = kd_ballot(cold) != 0ULL;
    first_nfs = kd_wave_min(first_nfs);
    last_rel = kd_wave_max(last_rel);
    KD_WAVE_SYNC();
    if (lane == 0) {
        const int64_t r_end = c_r;
        // the rows and the scans of k_long_expand hold query coordinates in 32 bits
        bool regular = pos0 >= 0 && !any_bad && c_q < (int64_t)0xffffffffLL;
        // a non-first S must be the last op that touches r (M, I, D or S)
        if (first_nfs != 0xffffffffu && last_rel > first_nfs) regular = false;
        int64_t foot_end = r_end;
        uint32_t lead = 0, clip_adv = 0;
        if (regular) {
            if ((cg[0] & 15u) == 4u) { const int64_t l0 = cg[0] >> 4; lead = (uint32_t)(l0 < pos0 ? l0 : pos0); }
            if (first_nfs != 0xffffffffu) {  // trailing clip: r at that op is r_end (nothing after it moves r)
                const int64_t ls = cg[first_nfs] >> 4;
                const int64_t adv = r_end < L ? (ls < L - r_end ? ls : L - r_end) : 0;
                foot_end += adv;
                clip_adv = (uint32_t)adv;
            }
        }
        kd_u64 span = foot_end > pos0 ? (kd_u64)(foot_end - pos0) : 0;
        if (span > 0x07fffff0ULL) { regular = false; span = 0; }
        KdRInfo ri = rinfo[i];
        // a regular long read KEEPS class LONG: k_window's first pass (class REG) leaves it alone, its row is tallied in the
        // second pass, its S / I side effects are k_long_expand's
        ri.span_cls = ((uint32_t)span << KD_SPAN_SHIFT) | (s_acc[wave][2] ? KD_INFO_INS : 0u) | (any_cold ? KD_INFO_COLD : 0u) |
                      (regular ? KD_CLS_LONG : KD_CLS_IRREG);
        ri.lead = regular ? lead : 0u;
        ri.pad = regular ? b + 1u : 0u;
        rinfo[i] = ri;
        KdLongAcc a;
        a.aligned = s_acc[wave][0]; a.walked = s_acc[wave][1]; a.insb = s_acc[wave][3]; a.n_ins = (uint32_t)s_acc[wave][2];
        a.lead = ri.lead; a.regular = regular ? 1u : 0u; a.clip_adv = regular ? clip_adv : 0u; a.pad = 0;
        a.row_span = regular ? (uint32_t)(r_end - pos0) + 1u : 0u;
        long_acc[b] = a;
    }
}

// k_long_reduce: one thread per long read.  Sums k_prep_long's per-read records into the status words and hands every long
// read its insertion-event slots, its pool range, its ROW (dword offset into the row buffer + the entry k_window's second
// pass plans and walks) and -- irregular ones -- its place in irreg_list: block scans give the offsets inside the workgroup,
// one returning atomic per quantity and WORKGROUP reserves its range (slot order is free).
__global__ void __launch_bounds__(KD_BLOCK)
k_long_reduce(const KdLongAcc *long_acc, const uint32_t *long_list, uint32_t n_long, const KdRInfo *rinfo, uint32_t *irreg_list,
              uint32_t *read_ev, kd_u64 *read_pool, KdRInfo *row_info, kd_u64 *row_off, kd_u64 *status) {
    __shared__ kd_u64 s_wave[2 * KD_WAVES_PER_BLOCK], s_base[4], s_sum[3];
    __shared__ uint32_t s_mx[2];
    const uint32_t t = threadIdx.x;
    const uint32_t b = blockIdx.x * KD_BLOCK + t;
    if (t < 3) s_sum[t] = 0;
    if (t < 2) s_mx[t] = 0;
    __syncthreads();
    KdLongAcc a;
    a.aligned = a.walked = a.insb = 0; a.n_ins = a.lead = a.row_span = a.clip_adv = a.pad = 0; a.regular = 1;
    const bool live = b < n_long;
    uint32_t i = 0, gs = 0;      // (the read and its G-start, wanted behind the scans: requested here, next to the record)
    if (live) { a = long_acc[b]; i = long_list[b]; gs = rinfo[i].gstart; }
    const kd_u64 n_irreg = live && !a.regular ? 1 : 0;
    const kd_u64 row_dw = (a.row_span + 7u) / 8u;     // 8 symbols per dword
    kd_u64 i_ev = a.n_ins, i_pool = a.insb, i_irreg = n_irreg, i_row = row_dw, tot_ev, tot_pool, tot_irreg, tot_row;
    kd_block_scan_incl2(i_ev, i_pool, s_wave, tot_ev, tot_pool);
    kd_block_scan_incl2(i_irreg, i_row, s_wave, tot_irreg, tot_row);
    if (a.aligned) atomicAdd(&s_sum[0], a.aligned);
    if (a.walked) atomicAdd(&s_sum[1], a.walked);
    if (live && a.regular) { atomicAdd(&s_sum[2], 1ULL); if (a.lead) atomicMax(&s_mx[0], a.lead); }
    if (a.row_span) atomicMax(&s_mx[1], a.row_span);
    __syncthreads();
    if (t == 0) s_base[0] = tot_ev ? atomicAdd(&status[KDS_N_EV], tot_ev) : 0;
    if (t == 1) s_base[1] = tot_pool ? atomicAdd(&status[KDS_POOL], tot_pool) : 0;
    if (t == 2) s_base[2] = tot_irreg ? atomicAdd(&status[KDS_B_N_IRREG], tot_irreg) : 0;
    if (t == 3 && s_sum[0]) atomicAdd(&status[KDS_ST_ALIGNED], s_sum[0]);
    if (t == 4 && s_sum[1]) atomicAdd(&status[KDS_ST_WALKED], s_sum[1]);
    if (t == 5 && tot_ev) { atomicAdd(&status[KDS_ST_INS], tot_ev); atomicAdd(&status[KDS_B_INS_OPS], tot_ev); }
    if (t == 6 && tot_pool) atomicAdd(&status[KDS_B_INS_BASES], tot_pool);
    if (t == 7 && s_sum[2]) atomicAdd(&status[KDS_B_N_REG], s_sum[2]);
    if (t == 8 && s_mx[0]) atomicMax(&status[KDS_B_MAXLEAD], (kd_u64)s_mx[0]);
    if (t == 9 && s_mx[1]) atomicMax(&status[KDS_B_MAXSEGSPAN], (kd_u64)s_mx[1]);
    if (t == 10) s_base[3] = tot_row ? atomicAdd(&status[KDS_B_ROW_DWORDS], tot_row) : 0;
    __syncthreads();
    if (live) {
        if (a.n_ins) { read_ev[i] = (uint32_t)(s_base[0] + i_ev - a.n_ins); read_pool[i] = s_base[1] + i_pool - a.insb; }
        if (!a.regular) irreg_list[s_base[2] + i_irreg - 1] = i;
        KdRInfo e;
        e.gstart = gs; e.lead = 0; e.pad = 0;
        e.span_cls = a.regular ? ((a.row_span << KD_SPAN_SHIFT) | KD_INFO_PLAIN | KD_CLS_REG) : KD_CLS_SKIP;
        row_info[b] = e;
        row_off[b] = 4ULL * (s_base[3] + i_row - row_dw);     // byte offset, as KdReads::seq_off wants it
    }
}

// 8 query bases from base q on as LINEAR nibbles (base q + i at bits 4i .. 4i+3): BAM packs the even base into the HIGH
// nibble of a byte, so the nibbles of every byte are swapped before the (q & 1) shift.  Reads 8 bytes from byte q / 2.
struct __attribute__((packed, aligned(1))) KdU64u { kd_u64 v; };
struct __attribute__((packed, aligned(1))) KdU32u { uint32_t v; };
__device__ __forceinline__ uint32_t kd_fetch8_lin(const uint8_t *seq, kd_u64 q) {
    kd_u64 y = reinterpret_cast<const KdU64u *>(seq + (q >> 1))->v;
    y = ((y & 0x0f0f0f0f0f0f0f0fULL) << 4) | ((y >> 4) & 0x0f0f0f0f0f0f0f0fULL);
    return (uint32_t)(y >> (4u * (uint32_t)(q & 1ULL)));
}
// The same from a copy of the read's packed bases in LDS (dwords; ob = byte offset of base q's byte in the copy): two aligned
// dword reads and a byte alignment instead of a trip to memory.
__device__ __forceinline__ uint32_t kd_fetch8_lin_lds(const uint32_t *s32, uint32_t ob, uint32_t odd) {
    const uint32_t d = ob >> 2, sh = ob & 3u;
    const uint32_t a = s32[d], b = s32[d + 1u];
    const uint32_t x = kd_alignbyte(b, a, sh);                 // bytes ob .. ob + 3
    const uint32_t xs = ((x & 0x0f0f0f0fu) << 4) | ((x >> 4) & 0x0f0f0f0fu);
    const uint32_t y = (b >> (8u * sh)) & 0xffu;               // byte ob + 4: its high nibble is the ninth base
    return odd ? (xs >> 4) | ((y >> 4) << 28) : xs;
}
// 8 linear BAM nibbles -> 8 linear row symbols: A,T,G,C,N = 1..5 (kd_chan's order + 1), anything else 0 = KD_ROW_SKIP.
// Two 8-entry byte tables looked up with v_perm_b32, a third v_perm selects by bit 3 of the nibble (as kd_codes8).
__device__ __forceinline__ uint32_t kd_rowcodes8(uint32_t z) {
    const uint32_t T0_LO = 0x00040100u, T0_HI = 0x00000003u;   // nibbles 0-7:  '=',A,C,M,G,R,S,V
    const uint32_t T1_LO = 0x00000002u, T1_HI = 0x05000000u;   // nibbles 8-15: T,W,Y,H,K,D,B,N
    const uint32_t lo = z & 0x0f0f0f0fu, hi = (z >> 4) & 0x0f0f0f0fu;
    const uint32_t il = lo & 0x07070707u, ih = hi & 0x07070707u;
    const uint32_t sl = ((lo >> 1) & 0x04040404u) | 0x03020100u, sh = ((hi >> 1) & 0x04040404u) | 0x03020100u;
    const uint32_t ml = kd_perm(kd_perm(T1_HI, T1_LO, il), kd_perm(T0_HI, T0_LO, il), sl);
    const uint32_t mh = kd_perm(kd_perm(T1_HI, T1_LO, ih), kd_perm(T0_HI, T0_LO, ih), sh);
    return ml | (mh << 4);
}
// nibbles of x that are zero, as bit 0 of the nibble
__device__ __forceinline__ uint32_t kd_zero_nibbles(uint32_t x) {
    uint32_t t = x | (x >> 1);
    t |= t >> 2;
    return ~t & 0x11111111u;
}

// k_long_expand: one WAVEFRONT per REGULAR long read (class LONG after k_prep_long), tiles of 64 ops (lane = op), the
// running coordinates carried in registers.  Per tile:
//   * lane = op: start coordinates from DPP scans; an I op writes its insertion event into the read's reserved slots
//     (neighbouring ops -> neighbouring slots); the query bases the tile consumes are copied into LDS;
//   * the row is written PIECE by piece.  A piece is what ONE op contributes to ONE row dword (8 sites): an M / D run of n
//     sites cut at the dword boundaries (an insertion in front of a run sets the "+ins" flag of the run's first site in
//     the run's first piece; the slot behind the last site is a piece too).  An op knows how many pieces it has (the dwords it touches); a DPP scan numbers the tile's pieces; 64
//     pieces at a time, lane = piece: the piece -> op table is the ops' own scatter of their first piece + a running
//     maximum (no search), the piece's 8 symbols are one fetch from the copy + one table look-up (kd_rowcodes8), OR-ed into
//     the chunk's dwords in LDS.  Pieces are in site order, so a chunk's dwords are consecutive and only its last one can
//     continue in the next chunk (or tile): it is carried.  Every lane does the same amount of work whatever the CIGAR looks
//     like -- a 5000-base M run is 625 pieces on 625 lanes, a run of 1-base deletions a piece each.  (One lane per row dword,
//     walking the ops that cover it, diverged on the op count: 110 lane-instructions per site, 0.9 of this kernel's 1.2 ms.)
//   * a second insertion at the SAME site of one read (I ops with nothing but N / P between them) cannot be a flag: it is
//     added to ins_total directly.
// Behind the tiles: the soft clips' weight tallies and clip_starts / clip_ends counters (atomics to HBM: two clips per read).
#ifndef KD_LONG_OCC
#define KD_LONG_OCC 6      // wavefronts per SIMD the register budget is set for (round 5, one wavefront per workgroup, longest reads first: 4 / 5 / 6 / 8 = 0.406 / 0.400 / 0.373 / 0.405 ms on C5; round 4, four reads per workgroup: 0.676 / 0.610 / 0.651)
#endif
__global__ void __launch_bounds__(KD_LONG_BLOCK, KD_LONG_OCC)
k_long_expand(KdReads rd, KdTabs T, KdIns ins, const KdRInfo *rinfo, const uint32_t *long_list, uint32_t n_long,
              const KdLongAcc *long_acc, const kd_u64 *row_off, uint8_t *rows, kd_u64 *status, const uint32_t *order) {
    // per wavefront: the tile's ops (one 16-byte record each: reference start, query start, CIGAR word, first piece | "+ins" flag), the
    // chunk's piece -> op table and dwords, a copy of the query bases the tile consumes
    __shared__ uint4 s_op_[KD_LONG_WAVES][KD_WAVE];
    __shared__ uint32_t s_pt_[KD_LONG_WAVES][KD_WAVE], s_out_[KD_LONG_WAVES][KD_WAVE];
    __shared__ uint32_t s_seq_[KD_LONG_WAVES][KD_LONG_SEQ_LDS / 4 + 4];
    const uint32_t lane = threadIdx.x & (KD_WAVE - 1), wave = threadIdx.x / KD_WAVE;
    const uint32_t x = blockIdx.x * KD_LONG_WAVES + wave;
    if (x >= n_long) return;
    const uint32_t b = order ? order[x] : x;        // (k_long_order: longest first)
    const kd_u64 i = long_list[b];
    const uint32_t sc = rinfo[i].span_cls;
    if ((sc & 3u) != KD_CLS_LONG) return;   // LONG after k_prep_long = regular long read
    uint4 *s_op = s_op_[wave];
    uint32_t *s_pt = s_pt_[wave], *s_out = s_out_[wave], *s_seq = s_seq_[wave];
    const KdLongAcc acc = long_acc[b];
    const uint32_t nc = kd_readfirstlane(rd.n_cig[i]);
    const uint32_t c = rd.contig[i];
    const int64_t L = T.contig_len[c];
    const kd_u64 cb = T.contig_base[c];
    const uint32_t sl = kd_readfirstlane(rd.seq_len[i]);
    const uint32_t pos0 = (uint32_t)rd.pos0[i];
    const kd_u64 g0 = cb + pos0;               // G-site of row symbol 0
    const uint8_t *seq = rd.seq4 + rd.seq_off[i];
    const uint32_t *cg = rd.cigar + rd.cig_off[i];
    uint32_t *row = reinterpret_cast<uint32_t *>(rows + row_off[b]);
    // (acc.row_span - 1 = M / D footprint F: symbols 0 .. F-1 are its sites, symbol F takes trailing insertions)
    const bool has_ins = (sc & KD_INFO_INS) != 0;
    kd_u64 e_base = 0, p_base = 0;
    if (has_ins) { e_base = ins.read_ev[i]; p_base = ins.read_pool[i]; }
    uint32_t *tab = T.tab;
    const kd_u64 S = T.stride;
    uint32_t c_r = 0, c_q = 0;                 // coordinates in front of the tile (r relative to pos0)
    uint32_t last_ins = 0;                     // 1 + r of the last I op seen (0: none)
    uint32_t trail_r = 0, trail_q = 0;         // the trailing clip's coordinates (its lane)
    bool bad = false, trail = false;
    // the row dword under construction: carried from chunk to chunk, tile to tile.  A dword in the making holds a site's symbol in bits
    // 0-2 of its nibble and "an insertion precedes this site" in bit 3 (one OR per piece); KD_ROW_FINISH turns the flag into the "+ins"
    // twin (symbol + KD_ROW_INS = symbol + 8 - 1) and the nibbles into BAM order
    uint32_t cj = 0xffffffffu, cval = 0;
    static_assert(KD_ROW_INS == 7u && KD_ROW_DEL < 8u, "the +ins flag is bit 3 of a symbol's nibble");
#define KD_ROW_FINISH(v) (((((v) - (((v) >> 3) & 0x11111111u)) & 0x0f0f0f0fu) << 4) | ((((v) - (((v) >> 3) & 0x11111111u)) >> 4) & 0x0f0f0f0fu))
    // (CIGAR words two tiles ahead, the first 256 bytes of the next tile's query bases one tile ahead: in flight while a tile is worked on)
    uint32_t w_nxt = lane < nc ? cg[lane] : 15u, w_nxt2 = lane + KD_WAVE < nc ? cg[lane + KD_WAVE] : 15u;
    const uint32_t seq_bytes = (sl >> 1) + (sl & 1u) + 8u;      // the read's packed bases + the 8-byte fetch window (include/kindel_hip.h: seq4_bytes + 16 are readable)
    uint32_t sq_pre = 4u * lane < seq_bytes ? reinterpret_cast<const KdU32u *>(seq + 4u * lane)->v : 0u;
    for (uint32_t base = 0; base <= nc; base += KD_WAVE) {   // (<=: the terminator behind the last op is a piece too)
        const uint32_t k = base + lane;
        const uint32_t w = w_nxt;
        w_nxt = w_nxt2;
        { const uint32_t kn = k + 2u * KD_WAVE; w_nxt2 = kn < nc ? cg[kn] : 15u; }
        const uint32_t len = w >> 4, op = w & 15u;
        const KdAdv adv = kd_op_advance(w, k);
        const uint32_t ra = adv.r, qa = adv.q;
        const uint32_t ir = kd_wave_scan_add(ra), iq = kd_wave_scan_add(qa);
        const uint32_t r_op = c_r + ir - ra, q_op = c_q + iq - qa;
        const uint32_t tot_q = kd_readlane(iq, KD_WAVE - 1);
        // pieces: an M / D run touches the dwords of its first to its last site; an I op and the terminator are one each
        const bool is_run = ra != 0, is_ins = (op == 1) & (k < nc);
        bool dup = false, insf = false;        // insf: an insertion sits in front of this run's first site (or of the terminator)
        if (has_ins) {   // (wave-uniform) a second I op on the same site: the nearest I op before it has the same r
            const uint32_t m = kd_wave_scan_max(is_ins ? r_op + 1u : 0u);
            uint32_t prev = kd_shfl_up(m, 1u);
            if (lane == 0) prev = 0;
            prev = prev > last_ins ? prev : last_ins;
            dup = is_ins & (prev == r_op + 1u);
            insf = (is_run | (k == nc)) & (prev == r_op + 1u);
            const uint32_t tile_last = kd_readlane(m, KD_WAVE - 1);
            last_ins = tile_last > last_ins ? tile_last : last_ins;
        }
        const uint32_t cnt = is_run ? ((r_op + ra - 1u) >> 3) - (r_op >> 3) + 1u : k == nc ? 1u : 0u;
        const uint32_t ipb = kd_wave_scan_add(cnt);
        const uint32_t n_pieces = kd_readlane(ipb, KD_WAVE - 1);
        KD_WAVE_SYNC();                        // (the last tile's pieces are done with the arrays)
        s_op[lane] = make_uint4(r_op, q_op, w, (ipb - cnt) | (insf ? 0x80000000u : 0u));
        // the query bases of the tile (+ the 8-base fetch window): copied when they fit
        const uint32_t qb = c_q & ~7u;                                      // first copied base: a dword boundary of the read's bytes
        uint32_t q_end = c_q + tot_q;
        q_end = q_end < sl ? q_end : sl;
        const uint32_t need = q_end > qb ? ((q_end - qb + 1u) >> 1) + 8u : 0u;   // bytes
        const bool staged = need <= KD_LONG_SEQ_LDS;
        if (staged) {
            s_seq[lane] = sq_pre;              // bytes 0 .. 255 of the copy were requested a tile ago
            for (uint32_t o = 4u * (KD_WAVE + lane); o < need; o += 4u * KD_WAVE)
                s_seq[o >> 2] = reinterpret_cast<const KdU32u *>(seq + (qb >> 1) + o)->v;
        }
        {   // the next tile's copy starts at the dword of its first base
            const uint32_t ob = ((c_q + tot_q) & ~7u) >> 1;
            sq_pre = ob + 4u * lane < seq_bytes ? reinterpret_cast<const KdU32u *>(seq + ob + 4u * lane)->v : 0u;
        }
        KD_WAVE_SYNC();                        // the tile's arrays are written
        if (has_ins) {   // (wave-uniform) event / pool slots of the tile's I ops
            uint32_t ni = 0, nb = 0;
            if (is_ins) {
                const uint32_t q0 = q_op < sl ? q_op : sl, q1 = q_op + len < sl ? q_op + len : sl;
                ni = 1; nb = q1 - q0;
            }
            const uint32_t i_ni = kd_wave_scan_add(ni), i_nb = kd_wave_scan_add(nb);
            if (is_ins) {
                const kd_u64 e = e_base + i_ni - 1, po = p_base + i_nb - nb;
                const kd_u64 g = g0 + r_op;
                if (e >= ins.ev_cap || po + nb > ins.pool_cap) {
                    atomicAdd(&status[KDS_INTERNAL], 1ULL);
                } else if (kd_commit(T, g)) {
                    ins.ev_site[e] = (uint32_t)g; ins.ev_len[e] = nb; ins.ev_off[e] = po;
                    const uint32_t q0 = q_op < sl ? q_op : sl;
                    for (uint32_t x0 = 0; x0 < nb; x0 += 8u) {       // 8 bases per fetch, one base code per pool byte
                        const uint32_t qq = q0 + x0;
                        uint32_t z = staged ? kd_fetch8_lin_lds(s_seq, (qq >> 1) - (qb >> 1), qq & 1u) : kd_fetch8_lin(seq, qq);
                        for (uint32_t x = x0; x < nb && x < x0 + 8u; x++, z >>= 4) ins.pool[po + x] = (uint8_t)(z & 15u);
                    }
                    if (dup) atomicAdd(&tab[(kd_u64)KDC_INS_TOTAL * S + g], 1u);     // (its site's "+ins" flag is taken)
                } else {
                    ins.ev_site[e] = KD_EV_DROPPED; ins.ev_len[e] = 0; ins.ev_off[e] = po;
                }
            }
            e_base += kd_readlane(i_ni, KD_WAVE - 1); p_base += kd_readlane(i_nb, KD_WAVE - 1);
        }
        if (kd_ballot((op == 4) & (k < nc)) != 0ULL) {      // (wave-uniform: the read's first and last tile, if any)
            if (op == 4 && k < nc) {
                if (k == 0) {      // leading clip, kindel.py:64-73: clip_ends[r]
                    if (kd_commit(T, g0)) atomicAdd(&tab[(kd_u64)KDC_CLIP_ENDS * S + g0], 1u);
                } else {           // non-first clip, kindel.py:74-81: clip_starts[r - 1] (index -1 wraps to the last slot)
                    const int64_t x = (int64_t)pos0 + r_op - 1;
                    const kd_u64 g = cb + (kd_u64)(x < 0 ? x + L + 1 : x);
                    if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)KDC_CLIP_STARTS * S + g], 1u);
                    trail = true; trail_r = r_op; trail_q = q_op;
                }
            }
        }
        // ---- the tile's pieces, 64 at a time: the same instructions for every lane whatever its piece is ----
        uint32_t op_in = 0;                    // 1 + the op the chunk's first piece continues (0: it starts an op)
        for (uint32_t p0 = 0; p0 < n_pieces; p0 += KD_WAVE) {
            s_pt[lane] = 0; s_out[lane] = 0;
            KD_WAVE_SYNC();
            { const uint32_t pb = ipb - cnt; if ((cnt != 0) & (pb >= p0) & (pb < p0 + KD_WAVE)) s_pt[pb - p0] = lane + 1u; }
            KD_WAVE_SYNC();
            uint32_t oi = kd_wave_scan_max(s_pt[lane]);
            oi = oi > op_in ? oi : op_in;      // (op indices grow with the piece number)
            op_in = kd_readlane(oi, KD_WAVE - 1);
            const uint32_t p = p0 + lane;
            const bool live = p < n_pieces;
            const uint4 rec = s_op[live ? oi - 1u : 0u];       // (a live piece always has its op: piece 0 starts one)
            const uint32_t r_k = rec.x, q_k = rec.y, w_k = rec.z, pbf = rec.w;
            const uint32_t idx = p - (pbf & 0x7fffffffu);
            const uint32_t ln = w_k >> 4, o = w_k & 15u;
            const uint32_t j = live ? (r_k >> 3) + idx : 0u;
            const bool run = live & (((0x185u >> o) & 1u) != 0);              // M, D, =, X (else: the terminator -- the slot behind the last site exists)
            // the run's sites inside row dword j: [a, e8), n of them from nibble ps on
            const uint32_t a = r_k > 8u * j ? r_k : 8u * j, e8 = r_k + ln < 8u * j + 8u ? r_k + ln : 8u * j + 8u;
            const uint32_t n = run ? e8 - a : 8u, ps = (a - 8u * j) & 7u;
            const uint32_t msk = 0xffffffffu >> (32u - 4u * n);               // (1 <= n <= 8)
            const uint32_t qq = q_k + (a - r_k);
            uint32_t sym;
            if (staged) sym = kd_rowcodes8(kd_fetch8_lin_lds(s_seq, run && o != 2 ? (qq >> 1) - (qb >> 1) : 0u, qq & 1u));      // (wave-uniform)
            else sym = run && o != 2 ? kd_rowcodes8(kd_fetch8_lin(seq, qq)) : 0u;
            if (run & (o != 2) & ((kd_zero_nibbles(sym) & msk) != 0)) bad = true;     // a base outside A,C,G,T,N (KeyError in the reference)
            sym = o == 2 ? 0x11111111u * KD_ROW_DEL : sym;
            uint32_t val = run ? (sym & msk) << (4u * ps) : 0u;
            if (live & (idx == 0) & ((pbf >> 31) != 0)) val |= 8u << (4u * (r_k & 7u));   // "+ins" flag of the run's first site
            const uint32_t jf = kd_readfirstlane(j);
            if (cj != 0xffffffffu && cj != jf) {            // the carried dword is complete
                if (lane == 0) row[cj] = KD_ROW_FINISH(cval);
                cval = 0;
            } else if (cj == 0xffffffffu) cval = 0;
            if (val) atomicOr(&s_out[j - jf], val);
            KD_WAVE_SYNC();
            const uint32_t last_lane = n_pieces - p0 < KD_WAVE ? n_pieces - p0 - 1u : KD_WAVE - 1u;
            const uint32_t jl = kd_readlane(j, last_lane);      // (last_lane: wave-uniform)
            uint32_t v = s_out[lane];
            if (lane == 0) v |= cval;
            if (jf + lane < jl) row[jf + lane] = KD_ROW_FINISH(v);
            cj = jl;
            cval = kd_readlane(v, jl - jf);
        }
        c_r += kd_readlane(ir, KD_WAVE - 1); c_q += tot_q;
    }
    if (cj != 0xffffffffu && lane == 0) row[cj] = KD_ROW_FINISH(cval);
#undef KD_ROW_FINISH
    // ---- soft clips: weights of the clipped bases (clip_end_weights in front of the read, clip_start_weights behind it) ----
    if ((cg[0] & 15u) == 4u && acc.lead) {
        const uint32_t len0 = cg[0] >> 4;
        for (uint32_t x = lane; x < acc.lead; x += KD_WAVE) {       // base len0 - lead + x -> site pos0 - lead + x
            const uint32_t ch = kd_chan(kd_nib(seq, (int64_t)(len0 - acc.lead + x)));
            const kd_u64 g = g0 - acc.lead + x;
            if (ch == 7u) bad = true;
            else if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)(KDC_CEW + ch) * S + g], 1u);
        }
    }
    const kd_u64 tmask = kd_ballot(trail);
    if (tmask && acc.clip_adv) {
        const uint32_t src = (uint32_t)__builtin_ctzll(tmask);
        const uint32_t r_t = kd_shfl(trail_r, src), q_t = kd_shfl(trail_q, src);
        for (uint32_t x = lane; x < acc.clip_adv; x += KD_WAVE) {   // base q + x -> site r + x
            const uint32_t ch = kd_chan(kd_nib(seq, (int64_t)q_t + x));
            const kd_u64 g = g0 + r_t + x;
            if (ch == 7u) bad = true;
            else if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)(KDC_CSW + ch) * S + g], 1u);
        }
    }
    // k_errors (kd_find_bad_base) pins down the read and the contig's first failure (it walks regular long reads too)
    if (bad) atomicAdd(&status[KDS_BAD_BASE], 1ULL);
}
