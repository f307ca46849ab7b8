// kd_kernels.h -- device code of the MI355X (gfx950 / CDNA4) pileup + consensus engine.
//
// Replaces the two Python loops of the reference:
//   parse_records record/CIGAR loop      /root/reference/kindel/kindel.py:40-81
//   consensus_sequence per-site loop     /root/reference/kindel/kindel.py:384-430
// (plus consensus() :369-381 and the depth min/max of build_report :450,477-479).
//
// This is integer histogramming: HBM/LDS-atomic bound, no MFMA.  Layout and kernels are
// described in DESIGN.md.  Short version:
//   tables   u32 tab[KD_NCH][S]  channel-major over "G-space" (all contigs back to back,
//            len+1 slots each, padded to 64) so that 64 lanes walking 64 consecutive sites
//            hit 64 consecutive dwords (coalesced atomics / stores, conflict-free LDS banks).
//   k_prep        lane per read: classify (skip / regular / irregular / long CIGAR, plain), footprint,
//                 stats, deterministic insertion-event slots; k_prep_long for CIGARs of > 16 words
//   k_plan_*      window -> candidate-read range (binary search on sorted starts) -> work items
//   k_window      persistent workgroups pull (window, slice) items; ONE LANE PER READ, 8 bases per
//                 dword, every base one ds_add_u32 into an LDS histogram (weights, deletions and both
//                 soft-clip weight tables; u16 counters, two sites per dword); one coalesced flush of
//                 the non-zero counters per item (atomicAdd, u32) into HBM
//   k_cold_lane   lane per read with S or I: clip start/end counters, insertion events
//   k_pileup_wave one wavefront per read, every reference quirk incl. Python negative-index wrap,
//                 32-bit atomics straight to HBM: irregular reads, unsorted batches, KD_MODE_GLOBAL
//   k_ins_*       insertion events -> open-addressing hash multiset -> per-site unique max
//   k_cns_*       per-site argmax / tie / indel rules, exclusive scan, byte emission
//
// The file has no host API calls and only uses __syncthreads + atomics across lanes, so
// tests/emu/ can execute the same source on the CPU for logic checks (test infrastructure).
#pragma once
#include <stdint.h>

typedef unsigned long long kd_u64;

#ifndef KD_DYN_SHARED
#define KD_DYN_SHARED(type, name)                                                  \
    extern __shared__ __attribute__((aligned(16))) unsigned char kd_dyn_smem_[];  \
    type *name = reinterpret_cast<type *>(kd_dyn_smem_)
#endif

// 24-bit multiply (operands < 2^24): full-rate on the VALU
#ifndef KD_MUL24
#define KD_MUL24(a, b) __umul24((a), (b))
#endif

#define KD_WAVE 64
#define KD_BLOCK 256
#define KD_WAVES_PER_BLOCK (KD_BLOCK / KD_WAVE)

// channel ids (mirrors include/kindel_hip.h)
#define KDC_A 0
#define KDC_T 1
#define KDC_G 2
#define KDC_C 3
#define KDC_N 4
#define KDC_DEL 5
#define KDC_CSW 6
#define KDC_CEW 11
#define KDC_CLIP_STARTS 16
#define KDC_CLIP_ENDS 17
#define KDC_INS_TOTAL 18
#define KDC_NCH 19

// read classes written by k_prep
#define KD_CLS_SKIP 0u    // flag & 4 or len(seq) <= 1                      kindel.py:43-46
#define KD_CLS_REG 1u     // no wrap, no overhang, no reference exception possible except a bad base
#define KD_CLS_IRREG 2u   // everything else: walked with exact Python semantics by k_pileup_wave
#define KD_CLS_LONG 3u    // transient: CIGAR too long for the per-lane scan, k_prep_long decides
#define KD_INFO_COLD 4u   // read has S or I ops (soft-clip tables / insertion events)
#define KD_INFO_INS 8u    // read has I ops: k_prep reserved its insertion-event / pool slots
#define KD_INFO_PLAIN 16u // regular read that is ONE M/=/X run covering the whole read (no clips, no indels)
#define KD_SPAN_SHIFT 5
#define KD_EV_DROPPED 0xffffffffu  // reserved insertion-event slot whose site belongs to another shard

// device status words (kd_u64 each)
enum {
    KDS_ERR_READ = 0,   // atomicMin of the global index of the first failing read (init ~0)
    KDS_ERR_CODE,       // written by k_diagnose
    KDS_N_EV,           // insertion events used
    KDS_POOL,           // insertion pool bytes used
    KDS_ST_READS,       // reads counted
    KDS_ST_ALIGNED,     // aligned-base events
    KDS_ST_WALKED,      // walked events
    KDS_ST_INS,         // insertion ops seen
    KDS_B_INS_OPS,      // per batch: insertion ops
    KDS_B_INS_BASES,    // per batch: insertion bases
    KDS_B_MAXSPAN,      // per batch: max span of regular reads
    KDS_B_MAXLEAD,      // per batch: max leading-clip reach of regular reads
    KDS_B_MAXSEGSPAN,   // per batch: max span of a long read's SEGMENT (k_prep_long; k_window's second pass)
    KDS_B_UNSORTED,     // per batch: reads not sorted by G-start
    KDS_B_N_COLD,       // per batch: entries in the cold list
    KDS_B_N_IRREG,      // per batch: entries in the irregular list
    KDS_B_N_LONG,       // per batch: entries in the long-CIGAR list
    KDS_B_N_REG,        // per batch: regular reads
    KDS_NEXT_ITEM,      // window work queue head
    KDS_TOTAL_ITEMS,    // window work queue length
    KDS_INS_COLLISION,  // hash verification failed
    KDS_INTERNAL,       // capacity overrun etc.
    KDS_BAD_BASE,       // k_window: windows that saw a base outside A,C,G,T,N
#ifdef KD_PHASE_CLOCKS
    KDS_DBG0, KDS_DBG1, KDS_DBG2, KDS_DBG3, KDS_DBG4, KDS_DBG5, KDS_DBG6, KDS_DBG7,   // phase clocks (profiling build only)
#endif
    KDS_COUNT
};

struct KdTabs {
    uint32_t *tab;               // [KDC_NCH][stride]
    kd_u64 stride;               // S = total G-space sites (multiple of 64)
    const uint32_t *contig_len;  // [n_contigs]
    const kd_u64 *contig_base;   // [n_contigs]
    kd_u64 g_lo, g_hi;           // commit increments with g_lo <= g <= g_hi (g_hi = halo site)
};

struct KdReads {
    kd_u64 n;
    kd_u64 base_index;  // global index of read 0 (over all pushed batches)
    const uint32_t *contig;
    const int32_t *pos0;
    const uint32_t *flag;
    const kd_u64 *seq_off;
    const uint32_t *seq_len;
    const kd_u64 *cig_off;
    const uint32_t *n_cig;
    const uint8_t *seq4;
    const uint32_t *cigar;
};

struct KdRInfo {
    uint32_t gstart;    // contig_base + max(pos0, 0)
    uint32_t span_cls;  // span << KD_SPAN_SHIFT | KD_INFO_INS | KD_INFO_COLD | class; span = sites from
                        // gstart to the end of the last M / D / trailing-S write
    uint32_t lead;      // sites before gstart written by a leading soft clip (kindel.py:68-72)
    uint32_t pad;       // long-CIGAR reads: 1 + index of the read's KdCkpt[256] block; 0 otherwise
};

// Long-CIGAR reads (k_prep_long): the state at the first op of each of the 256 per-thread op runs.  Lets
// k_cold_long emit a read's insertion events with 256 threads and lets k_window enter the read near a window
// instead of walking thousands of ops from the start.
struct KdCkpt {
    uint32_t r_rel;   // reference advance (r - pos0) before the run
    uint32_t q;       // query advance before the run
    uint32_t ev;      // insertion events of the read before the run
    uint32_t pool;    // insertion bases of the read before the run
};

struct KdIns {
    uint32_t *ev_site;  // [ev_cap] G-space site
    uint32_t *ev_len;   // [ev_cap] bases
    kd_u64 *ev_off;     // [ev_cap] offset into pool
    uint8_t *pool;      // one 4-bit base code per byte
    kd_u64 ev_cap, pool_cap;
    uint32_t *read_ev;  // [n reads of the batch] first event slot of the read (valid when KD_INFO_INS)
    kd_u64 *read_pool;  // [n reads of the batch] first pool byte of the read
};

// ---------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------

// BAM nibble -> weight channel in the reference's dict order A,T,G,C,N (kindel.py:29);
// 7 = not a key of that dict (KeyError in the reference).
__device__ __forceinline__ uint32_t kd_chan(uint32_t nib) {
    return (uint32_t)((0x4777777177727307ULL >> (nib * 4)) & 7ULL);
}

__device__ __forceinline__ uint32_t kd_nib(const uint8_t *seq, int64_t q) {
    uint32_t b = seq[q >> 1];
    return (q & 1) ? (b & 15u) : (b >> 4);
}

// 16 packed bases-bytes at ANY byte address: gfx950 global loads are unaligned-capable, hipcc emits one
// global_load_dwordx4 for this type.  Chunk c of a read holds its query bases 32c .. 32c+31.
struct __attribute__((packed, aligned(1))) KdChunk { uint32_t x, y, z, w; };
// bit position of base b (0..7) inside a little-endian dword of BAM nibbles (high nibble first)
#define KD_NIB_SHIFT(b) (8 * ((b) >> 1) + (((b) & 1) ? 0 : 4))

__device__ __forceinline__ bool kd_commit(const KdTabs &T, kd_u64 g) { return g >= T.g_lo && g <= T.g_hi; }

// CIGAR words k0 .. k0+3 of a read with nc words: one unaligned 16-byte load when all four are the read's own
// (never touches memory past the batch's CIGAR array), guarded single loads for the read's last group
__device__ __forceinline__ KdChunk kd_load_cigar4(const uint32_t *cg, uint32_t k0, uint32_t nc) {
    if (k0 + 4 <= nc) return *reinterpret_cast<const KdChunk *>(cg + k0);
    KdChunk r;
    r.x = k0 < nc ? cg[k0] : 0u; r.y = k0 + 1 < nc ? cg[k0 + 1] : 0u;
    r.z = k0 + 2 < nc ? cg[k0 + 2] : 0u; r.w = 0u;
    return r;
}


__device__ __forceinline__ void kd_flag_error(kd_u64 *status, kd_u64 gidx) { atomicMin(&status[KDS_ERR_READ], gidx); }

// ---------------------------------------------------------------------------------------
// k_prep: classify reads, compute the reference span of their table writes, count events.
// One lane per read, KD_PREP_PER_THREAD reads per lane so that the per-block reductions
// (stats, list reservations) cost one global atomic per 8192 reads.
// ---------------------------------------------------------------------------------------
#define KD_PREP_PER_THREAD 32
#define KD_PREP_CHUNK (KD_BLOCK * KD_PREP_PER_THREAD)
#define KD_PREP_MAX_OPS 16

// result of scanning one CIGAR
struct KdScan {
    uint32_t cls, cold, lead;
    kd_u64 span, n_ins, ins_bases, aligned, walked;
};

// Serial scan of ops [0, nc) of a read; shared by k_prep (short CIGARs) and k_diagnose-free
// paths.  "Regular" means: k_window / the COLD pass can process the read with plain
// G-space arithmetic and no Python wrap-around or exception can occur (bad bases aside).
// `pre` = the first 4 CIGAR words, already in registers (loaded together with those of other reads), or NULL
__device__ __forceinline__ KdScan kd_scan_cigar(const uint32_t *cg, uint32_t nc, int64_t pos0, int64_t sl, int64_t L,
                                                const uint32_t *pre = nullptr) {
    KdScan s;
    s.cls = KD_CLS_REG; s.cold = 0; s.lead = 0; s.span = 0; s.n_ins = 0; s.ins_bases = 0; s.aligned = 0; s.walked = 0;
    bool regular = pos0 >= 0;
    bool seen_nfs = false;  // a non-first S was seen: r is no longer plain prefix arithmetic
    int64_t r = pos0, q = 0, hot_hi = pos0;
    for (uint32_t k = 0; k < nc; k++) {
        const uint32_t c = (pre && k < 4) ? (k == 0 ? pre[0] : k == 1 ? pre[1] : k == 2 ? pre[2] : pre[3]) : cg[k];
        const int64_t len = c >> 4;
        const uint32_t op = c & 15u;
        if (op == 0 || op == 7 || op == 8) {  // M = X
            if (seen_nfs || r + len > L || q + len > sl) regular = false;
            r += len; q += len; hot_hi = r;
            s.aligned += (kd_u64)len; s.walked += (kd_u64)len;
        } else if (op == 1) {  // I
            s.cold = KD_INFO_COLD;
            if (seen_nfs || r > L) regular = false;
            int64_t q0 = q < sl ? q : sl, q1 = q + len < sl ? q + len : sl;
            s.n_ins += 1; s.ins_bases += (kd_u64)(q1 - q0);
            q += len; s.walked += (kd_u64)len;
        } else if (op == 2) {  // D
            if (seen_nfs || r + len > L + 1) regular = false;
            r += len; hot_hi = r;
            s.walked += (kd_u64)len;
        } else if (op == 4) {  // S
            s.cold = KD_INFO_COLD;
            s.walked += (kd_u64)len;
            if (k == 0) {
                if (r > L || len > sl) regular = false;
                s.lead = (uint32_t)(len < r ? len : (r > 0 ? r : 0));
                q += len;
            } else {
                if (seen_nfs || r - 1 > L) regular = false;   // clip_starts[r - 1] must exist (kindel.py:75)
                seen_nfs = true;
                int64_t n_adv = r < L ? (len < L - r ? len : L - r) : 0;
                if (n_adv > sl - q || (len > n_adv && q + n_adv >= sl)) regular = false;
                r += n_adv; q += n_adv;
                hot_hi = r;  // the clip_start_weights writes extend the read's footprint
            }
        }
    }
    if (!regular) s.cls = KD_CLS_IRREG;
    s.span = hot_hi > pos0 ? (kd_u64)(hot_hi - pos0) : 0;
    return s;
}

__global__ void __launch_bounds__(KD_BLOCK)
k_prep(KdReads rd, KdTabs T, KdRInfo *rinfo, uint32_t *cold_list, uint32_t *irreg_list, uint32_t *long_list,
       uint32_t *read_ev, kd_u64 *read_pool, kd_u64 *status) {
    __shared__ kd_u64 s_red[8];       // reads, aligned, walked, ins_ops, ins_bases, n_reg, unsorted
    __shared__ uint32_t s_maxspan, s_maxlead;
    __shared__ uint32_t s_cnt[3];     // cold, irreg, long (block totals / running offsets)
    __shared__ kd_u64 s_base[5];
    __shared__ kd_u64 s_ins[2];       // insertion events / insertion bases of the block's short-CIGAR reads
    const uint32_t t = threadIdx.x;
    if (t < 8) s_red[t] = 0;
    if (t < 3) s_cnt[t] = 0;
    if (t < 2) s_ins[t] = 0;
    if (t == 0) { s_maxspan = 0; s_maxlead = 0; }
    __syncthreads();
    const kd_u64 chunk0 = (kd_u64)blockIdx.x * KD_PREP_CHUNK;
    kd_u64 a_reads = 0, a_aligned = 0, a_walked = 0, a_ins = 0, a_insb = 0, a_reg = 0, a_unsorted = 0;
    uint32_t a_maxspan = 0, a_maxlead = 0, n_cold = 0, n_irreg = 0, n_long = 0;
    uint32_t m_cold = 0, m_irreg = 0, m_long = 0, m_ins = 0;  // bit `it` = this thread's it-th read is in the list
    uint32_t c_cached = 0xffffffffu;                             // one-entry cache of the contig table
    kd_u64 cb_cached = 0;
    int64_t L_cached = 0;
    // 4 reads per step: all of their metadata loads are issued before any is consumed
    for (int it0 = 0; it0 < KD_PREP_PER_THREAD; it0 += 4) {
        uint32_t v_c[4], v_pc[4], v_nc[4], v_fl[4];
        int64_t v_pos[4], v_ppos[4], v_sl[4];
        kd_u64 v_coff[4];
        bool v_ok[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const kd_u64 i = chunk0 + (kd_u64)(it0 + u) * KD_BLOCK + t;
            v_ok[u] = i < rd.n;
            const kd_u64 j = v_ok[u] ? i : 0, jp = (v_ok[u] && i > 0) ? i - 1 : j;
            v_c[u] = rd.contig[j]; v_pos[u] = rd.pos0[j];
            v_pc[u] = rd.contig[jp]; v_ppos[u] = rd.pos0[jp];
            v_sl[u] = rd.seq_len[j]; v_nc[u] = rd.n_cig[j]; v_fl[u] = rd.flag[j]; v_coff[u] = rd.cig_off[j];
        }
        // second level: the first 4 CIGAR words of each of the 4 reads, again all in flight together
        uint32_t v_cw[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t *cgp = rd.cigar + v_coff[u];
            const uint32_t ncu = v_ok[u] ? v_nc[u] : 0u;
#pragma unroll
            for (int k = 0; k < 4; k++) v_cw[u][k] = (uint32_t)k < ncu ? cgp[k] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!v_ok[u]) continue;
            const int it = it0 + u;
            const kd_u64 i = chunk0 + (kd_u64)it * KD_BLOCK + t;
            const uint32_t c = v_c[u];
            if (c != c_cached) { c_cached = c; cb_cached = T.contig_base[c]; L_cached = (int64_t)T.contig_len[c]; }
            const int64_t pos0 = v_pos[u];
            const kd_u64 gkey = cb_cached + (kd_u64)(pos0 > 0 ? pos0 : 0);
            {   // sortedness of G-start over ALL reads of the batch (window ranges rely on it)
                const kd_u64 pcb = v_pc[u] == c ? cb_cached : T.contig_base[v_pc[u]];
                const kd_u64 pk = pcb + (kd_u64)(v_ppos[u] > 0 ? v_ppos[u] : 0);
                if (pk > gkey) a_unsorted++;
            }
            const int64_t sl = v_sl[u];
            const uint32_t nc = v_nc[u];
            uint32_t cls, cold = 0, lead = 0;
            kd_u64 span = 0, al = 0, n_ins_r = 0, n_insb_r = 0;
            bool has_ins = false;
            if ((v_fl[u] & 4u) || sl <= 1) {
                cls = KD_CLS_SKIP;
            } else if (nc == 0) {
                cls = KD_CLS_IRREG;  // CIGAR '*': k_pileup_wave raises KD_E_CIGAR
                a_reads++;
            } else if (nc > KD_PREP_MAX_OPS) {
                cls = KD_CLS_LONG;
                a_reads++;
            } else {
                KdScan s = kd_scan_cigar(rd.cigar + v_coff[u], nc, pos0, sl, L_cached, v_cw[u]);
                cls = s.cls; cold = s.cold; span = s.span; lead = s.lead; has_ins = s.n_ins != 0; al = s.aligned;
                n_ins_r = s.n_ins; n_insb_r = s.ins_bases;
                a_reads++; a_aligned += s.aligned; a_walked += s.walked; a_ins += s.n_ins; a_insb += s.ins_bases;
            }
            if (span > 0x07ffffffULL) { cls = KD_CLS_IRREG; span = 0; }
            if (cls == KD_CLS_REG) {
                a_reg++;
                if ((uint32_t)span > a_maxspan) a_maxspan = (uint32_t)span;
                if (lead > a_maxlead) a_maxlead = lead;
            }
            if (cls == KD_CLS_REG && cold) { n_cold++; m_cold |= 1u << it; }
            if (cls == KD_CLS_IRREG) { n_irreg++; m_irreg |= 1u << it; }
            if (cls == KD_CLS_LONG) { n_long++; m_long |= 1u << it; }
            if (has_ins) { m_ins |= 1u << it; read_ev[i] = (uint32_t)n_ins_r; read_pool[i] = n_insb_r; }   // counts, see the last loop
            KdRInfo ri;
            ri.gstart = (uint32_t)gkey;
            // plain: the whole read is ONE aligned run: a single op whose aligned length is the read length
            const uint32_t plain = (cls == KD_CLS_REG && nc == 1 && !cold && al == (kd_u64)sl && span == al) ? KD_INFO_PLAIN : 0u;
            ri.span_cls = ((uint32_t)span << KD_SPAN_SHIFT) | plain | (has_ins ? KD_INFO_INS : 0u) | cold | cls;
            ri.lead = lead; ri.pad = 0;
            rinfo[i] = ri;
        }
    }
    // block reduction through LDS atomics, then one global atomic per word per block
    if (a_reads) atomicAdd(&s_red[0], a_reads);
    if (a_aligned) atomicAdd(&s_red[1], a_aligned);
    if (a_walked) atomicAdd(&s_red[2], a_walked);
    if (a_ins) atomicAdd(&s_red[3], a_ins);
    if (a_insb) atomicAdd(&s_red[4], a_insb);
    if (a_reg) atomicAdd(&s_red[5], a_reg);
    if (a_unsorted) atomicAdd(&s_red[6], a_unsorted);
    if (a_maxspan) atomicMax(&s_maxspan, a_maxspan);
    if (a_maxlead) atomicMax(&s_maxlead, a_maxlead);
    // list slots: thread-local offset inside the block
    uint32_t o_cold = n_cold ? atomicAdd(&s_cnt[0], n_cold) : 0;
    uint32_t o_irreg = n_irreg ? atomicAdd(&s_cnt[1], n_irreg) : 0;
    uint32_t o_long = n_long ? atomicAdd(&s_cnt[2], n_long) : 0;
    // insertion event / pool slots: thread-local offsets inside the block, one global reservation per block
    // (a global counter bumped per event serialises at ~11 ns per returning atomic on one address)
    const kd_u64 o_ev = a_ins ? atomicAdd(&s_ins[0], a_ins) : 0;
    const kd_u64 o_pool = a_ins ? atomicAdd(&s_ins[1], a_insb) : 0;
    __syncthreads();
    if (t == 0) {
        if (s_red[0]) atomicAdd(&status[KDS_ST_READS], s_red[0]);
        if (s_red[1]) atomicAdd(&status[KDS_ST_ALIGNED], s_red[1]);
        if (s_red[2]) atomicAdd(&status[KDS_ST_WALKED], s_red[2]);
        if (s_red[3]) { atomicAdd(&status[KDS_ST_INS], s_red[3]); atomicAdd(&status[KDS_B_INS_OPS], s_red[3]); }
        if (s_red[4]) atomicAdd(&status[KDS_B_INS_BASES], s_red[4]);
        s_base[3] = s_ins[0] ? atomicAdd(&status[KDS_N_EV], s_ins[0]) : 0;
        s_base[4] = s_ins[0] ? atomicAdd(&status[KDS_POOL], s_ins[1]) : 0;
        if (s_red[5]) atomicAdd(&status[KDS_B_N_REG], s_red[5]);
        if (s_red[6]) atomicAdd(&status[KDS_B_UNSORTED], s_red[6]);
        if (s_maxspan) atomicMax(&status[KDS_B_MAXSPAN], (kd_u64)s_maxspan);
        if (s_maxlead) atomicMax(&status[KDS_B_MAXLEAD], (kd_u64)s_maxlead);
        s_base[0] = s_cnt[0] ? atomicAdd(&status[KDS_B_N_COLD], (kd_u64)s_cnt[0]) : 0;
        s_base[1] = s_cnt[1] ? atomicAdd(&status[KDS_B_N_IRREG], (kd_u64)s_cnt[1]) : 0;
        s_base[2] = s_cnt[2] ? atomicAdd(&status[KDS_B_N_LONG], (kd_u64)s_cnt[2]) : 0;
    }
    __syncthreads();
    if (m_cold | m_irreg | m_long | m_ins) {
        kd_u64 w_cold = s_base[0] + o_cold, w_irreg = s_base[1] + o_irreg, w_long = s_base[2] + o_long;
        kd_u64 w_ev = s_base[3] + o_ev, w_pool = s_base[4] + o_pool;
        for (uint32_t todo = m_cold | m_irreg | m_long | m_ins; todo; todo &= todo - 1) {
            const int it = __builtin_ctz(todo);
            const kd_u64 i = chunk0 + (kd_u64)it * KD_BLOCK + t;
            const uint32_t bit = 1u << it;
            if (m_ins & bit) {  // the first pass left the read's event / base COUNTS here: turn them into its slots
                const uint32_t n_ev = read_ev[i];
                const kd_u64 n_b = read_pool[i];
                read_ev[i] = (uint32_t)w_ev; read_pool[i] = w_pool;
                w_ev += n_ev; w_pool += n_b;
            }
            if (m_cold & bit) cold_list[w_cold++] = (uint32_t)i;
            if (m_irreg & bit) irreg_list[w_irreg++] = (uint32_t)i;
            if (m_long & bit) long_list[w_long++] = (uint32_t)i;
        }
    }
}

// k_prep_long: one workgroup per read whose CIGAR has more than KD_PREP_MAX_OPS words
// (long-read aligners: thousands of ops).  Each thread sums the reference / query advance
// of a contiguous run of ops, an LDS scan turns the sums into start coordinates, and a
// second sweep applies the same regularity rules as kd_scan_cigar.
__global__ void __launch_bounds__(KD_BLOCK)
k_prep_long(KdReads rd, KdTabs T, KdRInfo *rinfo, const uint32_t *long_list, KdCkpt *ckpt, KdRInfo *seginfo,
            uint32_t *irreg_list, uint32_t *read_ev, kd_u64 *read_pool, kd_u64 *status) {
    __shared__ int64_t s_r[KD_BLOCK], s_q[KD_BLOCK];
    __shared__ uint32_t s_ni[KD_BLOCK], s_nb[KD_BLOCK];
    __shared__ kd_u64 s_acc[6];       // aligned, walked, n_ins, ins_bases, bad, cold
    __shared__ uint32_t s_first_nfs, s_last_rel;
    __shared__ uint32_t s_regular, s_lead, s_gstart, s_nfs_adv, s_maxseg;
    const uint32_t t = threadIdx.x;
    const kd_u64 i = long_list[blockIdx.x];
    const uint32_t c = rd.contig[i];
    const int64_t L = T.contig_len[c];
    const int64_t pos0 = rd.pos0[i];
    const int64_t sl = rd.seq_len[i];
    const uint32_t nc = rd.n_cig[i];
    const uint32_t *cg = rd.cigar + rd.cig_off[i];
    const uint32_t per = (nc + KD_BLOCK - 1) / KD_BLOCK;
    const uint32_t k0 = t * per < nc ? t * per : nc, k1 = k0 + per < nc ? k0 + per : nc;
    if (t < 6) s_acc[t] = 0;
    if (t == 0) { s_first_nfs = 0xffffffffu; s_last_rel = 0; s_regular = 0; s_lead = 0; s_gstart = 0; s_nfs_adv = 0; s_maxseg = 0; }
    int64_t dr = 0, dq = 0;
    for (uint32_t k = k0; k < k1; k++) {
        const uint32_t w = cg[k];
        const int64_t len = w >> 4;
        const uint32_t op = w & 15u;
        if (op == 0 || op == 7 || op == 8) { dr += len; dq += len; }
        else if (op == 1) dq += len;
        else if (op == 2) dr += len;
        else if (op == 4 && k == 0) dq += len;
        // a non-first S contributes nothing here: anything after it makes the read irregular,
        // and if nothing follows its own advance is irrelevant to the span
    }
    s_r[t] = dr; s_q[t] = dq;
    __syncthreads();
    // inclusive Hillis-Steele scan over the 256 partial sums
    for (uint32_t d = 1; d < KD_BLOCK; d <<= 1) {
        int64_t ar = 0, aq = 0;
        if (t >= d) { ar = s_r[t - d]; aq = s_q[t - d]; }
        __syncthreads();
        s_r[t] += ar; s_q[t] += aq;
        __syncthreads();
    }
    int64_t r = pos0 + (t ? s_r[t - 1] : 0), q = t ? s_q[t - 1] : 0;
    const int64_t r_end = pos0 + s_r[KD_BLOCK - 1];
    const int64_t r_run = r, q_run = q;   // checkpoint: state before this thread's run of ops
    kd_u64 aligned = 0, walked = 0, n_ins = 0, insb = 0, bad = 0, cold = 0;
    uint32_t first_nfs = 0xffffffffu, last_rel = 0;
    for (uint32_t k = k0; k < k1; k++) {
        const uint32_t w = cg[k];
        const int64_t len = w >> 4;
        const uint32_t op = w & 15u;
        if (op == 0 || op == 7 || op == 8) {
            if (r + len > L || q + len > sl) bad = 1;
            r += len; q += len; aligned += (kd_u64)len; walked += (kd_u64)len; last_rel = k;
        } else if (op == 1) {
            cold = 1;
            if (r > L) bad = 1;
            int64_t q0 = q < sl ? q : sl, q1 = q + len < sl ? q + len : sl;
            n_ins++; insb += (kd_u64)(q1 - q0); q += len; walked += (kd_u64)len; last_rel = k;
        } else if (op == 2) {
            if (r + len > L + 1) bad = 1;
            r += len; walked += (kd_u64)len; last_rel = k;
        } else if (op == 4) {
            cold = 1; walked += (kd_u64)len;
            if (k == 0) { if (r > L || len > sl) bad = 1; q += len; }
            else {
                if (k < first_nfs) first_nfs = k;
                if (r - 1 > L) bad = 1;   // clip_starts[r - 1] must exist (kindel.py:75)
                int64_t n_adv = r < L ? (len < L - r ? len : L - r) : 0;
                if (n_adv > sl - q || (len > n_adv && q + n_adv >= sl)) bad = 1;
                last_rel = k;
            }
        }
    }
    if (aligned) atomicAdd(&s_acc[0], aligned);
    if (walked) atomicAdd(&s_acc[1], walked);
    if (n_ins) atomicAdd(&s_acc[2], n_ins);
    if (insb) atomicAdd(&s_acc[3], insb);
    if (bad) atomicAdd(&s_acc[4], bad);
    if (cold) atomicAdd(&s_acc[5], cold);
    if (first_nfs != 0xffffffffu) atomicMin(&s_first_nfs, first_nfs);
    if (last_rel) atomicMax(&s_last_rel, last_rel);
    // exclusive prefix of the per-run insertion counts -> event / pool offsets inside the read
    s_ni[t] = (uint32_t)n_ins; s_nb[t] = (uint32_t)insb;
    __syncthreads();
    for (uint32_t d = 1; d < KD_BLOCK; d <<= 1) {
        uint32_t a = 0, b = 0;
        if (t >= d) { a = s_ni[t - d]; b = s_nb[t - d]; }
        __syncthreads();
        s_ni[t] += a; s_nb[t] += b;
        __syncthreads();
    }
    {
        KdCkpt ck;
        ck.r_rel = (uint32_t)(r_run - pos0); ck.q = (uint32_t)q_run;
        ck.ev = s_ni[t] - (uint32_t)n_ins; ck.pool = s_nb[t] - (uint32_t)insb;
        ckpt[(kd_u64)blockIdx.x * KD_BLOCK + t] = ck;
    }
    if (t == 0) {
        bool regular = pos0 >= 0 && s_acc[4] == 0;
        // a non-first S must be the last op that touches r (M, I, D or S)
        if (s_first_nfs != 0xffffffffu && s_last_rel > s_first_nfs) regular = false;
        int64_t foot_end = r_end;
        uint32_t lead = 0;
        if (regular) {
            if ((cg[0] & 15u) == 4u) { const int64_t l0 = cg[0] >> 4; lead = (uint32_t)(l0 < pos0 ? l0 : pos0); }
            if (s_first_nfs != 0xffffffffu) {  // trailing clip: r at that op is r_end (nothing after it moves r)
                const int64_t ls = cg[s_first_nfs] >> 4;
                const int64_t adv = r_end < L ? (ls < L - r_end ? ls : L - r_end) : 0;
                foot_end += adv;
                s_nfs_adv = (uint32_t)adv;
            }
        }
        kd_u64 span = foot_end > pos0 ? (kd_u64)(foot_end - pos0) : 0;
        if (span > 0x07ffffffULL) { regular = false; span = 0; }
        const uint32_t coldbit = s_acc[5] ? KD_INFO_COLD : 0u;
        KdRInfo ri = rinfo[i];
        // a regular long read KEEPS class LONG: k_window's first pass (class REG) leaves it alone, its aligned and
        // deleted bases are tallied segment by segment in the second pass, its S / I side effects by k_cold_long
        ri.span_cls = ((uint32_t)span << KD_SPAN_SHIFT) | (s_acc[2] ? KD_INFO_INS : 0u) | coldbit |
                      (regular ? KD_CLS_LONG : KD_CLS_IRREG);
        ri.lead = regular ? lead : 0u;
        ri.pad = regular ? blockIdx.x + 1u : 0u;
        rinfo[i] = ri;
        s_regular = regular ? 1u : 0u; s_lead = ri.lead; s_gstart = ri.gstart;
        if (s_acc[2]) {
            read_ev[i] = (uint32_t)atomicAdd(&status[KDS_N_EV], s_acc[2]);
            read_pool[i] = atomicAdd(&status[KDS_POOL], s_acc[3]);
        }
        atomicAdd(&status[KDS_ST_ALIGNED], s_acc[0]);
        atomicAdd(&status[KDS_ST_WALKED], s_acc[1]);
        if (s_acc[2]) { atomicAdd(&status[KDS_ST_INS], s_acc[2]); atomicAdd(&status[KDS_B_INS_OPS], s_acc[2]); }
        if (s_acc[3]) atomicAdd(&status[KDS_B_INS_BASES], s_acc[3]);
        if (regular) {
            atomicAdd(&status[KDS_B_N_REG], 1ULL);
            if (lead) atomicMax(&status[KDS_B_MAXLEAD], (kd_u64)lead);
            // (its S / I side effects are done by k_cold_long, 256 threads per read)
        } else {
            irreg_list[atomicAdd(&status[KDS_B_N_IRREG], 1ULL)] = (uint32_t)i;
        }
    }
    __syncthreads();
    // SEGMENTS: this thread's run of ops [k0, k1) as a work unit of its own -- where it starts on the reference
    // (checkpoint), how far its M / D / trailing-clip tallies reach.  k_window's second pass treats the segments of all
    // long reads like a batch of short reads: bucket-sorted by window, one lane per segment, a few ops each,
    // instead of one lane crawling through the hundreds of ops a long read has inside a window.
    {
        KdRInfo v;
        v.gstart = 0; v.span_cls = KD_CLS_SKIP; v.lead = 0; v.pad = 0;
        if (s_regular && k0 < k1) {
            kd_u64 sp = (kd_u64)(r - r_run);                         // M and D advance of the run
            if (first_nfs != 0xffffffffu) sp += s_nfs_adv;           // the trailing clip's clip_start_weights reach
            const uint32_t ld = k0 == 0 ? s_lead : 0u;               // the leading clip reaches back from the read's start
            if (sp > 0 || ld > 0) {
                v.gstart = s_gstart + (uint32_t)(r_run - pos0);
                v.span_cls = ((uint32_t)sp << KD_SPAN_SHIFT) | KD_CLS_REG;
                v.lead = ld; v.pad = blockIdx.x + 1u;
                atomicMax(&s_maxseg, (uint32_t)sp);
            }
        }
        seginfo[(kd_u64)blockIdx.x * KD_BLOCK + t] = v;
    }
    __syncthreads();
    if (t == 0 && s_maxseg) atomicMax(&status[KDS_B_MAXSEGSPAN], (kd_u64)s_maxseg);
}

// ---------------------------------------------------------------------------------------
// k_pileup_wave<HOT, COLD>: one wavefront per read, exact reference semantics
// (kindel.py:40-81 incl. Python negative-index wrap-around), 32-bit atomics into HBM.
//   HOT : commit M/=/X and D tallies (weights, deletions)
//   COLD: commit soft-clip tables and emit insertion events
// <true,true> is what runs: irregular reads, unsorted batches and KD_MODE_GLOBAL.
// All control flow is wave-uniform (every value steering it comes from uniform loads).
// ---------------------------------------------------------------------------------------
template <bool HOT, bool COLD>
__global__ void __launch_bounds__(KD_BLOCK)
k_pileup_wave(KdReads rd, KdTabs T, KdIns ins, const uint32_t *list, kd_u64 n_list, const KdRInfo *rinfo,
              kd_u64 *status) {
    const uint32_t lane = threadIdx.x & (KD_WAVE - 1);
    const kd_u64 slot = (kd_u64)blockIdx.x * KD_WAVES_PER_BLOCK + (threadIdx.x / KD_WAVE);
    if (slot >= n_list) return;
    const kd_u64 i = list ? (kd_u64)list[slot] : slot;
    const int64_t sl = rd.seq_len[i];
    if ((rd.flag[i] & 4u) || sl <= 1) return;  // kindel.py:43-46
    const kd_u64 gidx = rd.base_index + i;
    const uint32_t nc = rd.n_cig[i];
    if (nc == 0) { if (lane == 0) kd_flag_error(status, gidx); return; }  // kindel.py:47
    const uint32_t c = rd.contig[i];
    const int64_t L = T.contig_len[c];
    const kd_u64 cb = T.contig_base[c];
    const uint8_t *seq = rd.seq4 + rd.seq_off[i];
    const uint32_t *cg = rd.cigar + rd.cig_off[i];
    uint32_t *tab = T.tab;
    const kd_u64 S = T.stride;
    int64_t r = rd.pos0[i], q = 0;  // kindel.py:41-42
    kd_u64 ev_next = 0, pool_next = 0;  // this read's reserved insertion slots (k_prep), loaded at its first I
    bool ev_loaded = false;
    for (uint32_t k = 0; k < nc; k++) {
        const uint32_t w = cg[k];
        const int64_t len = w >> 4;
        const uint32_t op = w & 15u;
        if (op == 0 || op == 7 || op == 8) {  // M = X  kindel.py:49-54
            if (len > 0 && (q + len > sl || r + len > L || r < -L)) { if (lane == 0) kd_flag_error(status, gidx); return; }
            if (HOT) {
                for (int64_t j = lane; j < len; j += KD_WAVE) {
                    int64_t idx = r + j;
                    if (idx < 0) idx += L;
                    const uint32_t ch = kd_chan(kd_nib(seq, q + j));
                    const kd_u64 g = cb + (kd_u64)idx;
                    if (ch == 7u) kd_flag_error(status, gidx);
                    else if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)ch * S + g], 1u);
                }
            }
            r += len; q += len;
        } else if (op == 1) {  // I  kindel.py:55-58
            if (r > L || r < -(L + 1)) { if (lane == 0) kd_flag_error(status, gidx); return; }
            if (COLD) {
                if (!ev_loaded) { ev_next = ins.read_ev[i]; pool_next = ins.read_pool[i]; ev_loaded = true; }
                const int64_t q0 = q < sl ? q : sl, q1 = q + len < sl ? q + len : sl;
                const kd_u64 n = (kd_u64)(q1 - q0);
                const kd_u64 e = ev_next, po = pool_next;
                ev_next += 1; pool_next += n;
                if (lane == 0) {
                    const int64_t idx = r < 0 ? r + L + 1 : r;
                    const kd_u64 g = cb + (kd_u64)idx;
                    if (e >= ins.ev_cap || po + n > ins.pool_cap) {
                        atomicAdd(&status[KDS_INTERNAL], 1ULL);
                    } else if (kd_commit(T, g)) {
                        ins.ev_site[e] = (uint32_t)g; ins.ev_len[e] = (uint32_t)n; ins.ev_off[e] = po;
                        for (kd_u64 b = 0; b < n; b++) ins.pool[po + b] = (uint8_t)kd_nib(seq, q0 + (int64_t)b);
                        atomicAdd(&tab[(kd_u64)KDC_INS_TOTAL * S + g], 1u);
                    } else {
                        ins.ev_site[e] = KD_EV_DROPPED; ins.ev_len[e] = 0; ins.ev_off[e] = po;  // other shard's site
                    }
                }
            }
            q += len;
        } else if (op == 2) {  // D  kindel.py:59-62
            if (len > 0 && (r + len - 1 > L || r < -(L + 1))) { if (lane == 0) kd_flag_error(status, gidx); return; }
            if (HOT) {
                for (int64_t j = lane; j < len; j += KD_WAVE) {
                    int64_t idx = r + j;
                    if (idx < 0) idx += L + 1;
                    const kd_u64 g = cb + (kd_u64)idx;
                    if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)KDC_DEL * S + g], 1u);
                }
            }
            r += len;
        } else if (op == 4) {  // S
            if (k == 0) {  // kindel.py:64-73
                if (r > L || r < -(L + 1) || len > sl) { if (lane == 0) kd_flag_error(status, gidx); return; }
                if (COLD) {
                    if (lane == 0) {
                        const kd_u64 g = cb + (kd_u64)(r < 0 ? r + L + 1 : r);
                        if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)KDC_CLIP_ENDS * S + g], 1u);
                    }
                    for (int64_t j = lane; j < len; j += KD_WAVE) {
                        const int64_t rel = r - len + j;
                        if (rel >= 0) {
                            const uint32_t ch = kd_chan(kd_nib(seq, j));
                            const kd_u64 g = cb + (kd_u64)rel;
                            if (ch == 7u) kd_flag_error(status, gidx);
                            else if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)(KDC_CEW + ch) * S + g], 1u);
                        }
                    }
                }
                q += len;
            } else {  // kindel.py:74-81
                const int64_t x = r - 1;
                if (x > L || x < -(L + 1)) { if (lane == 0) kd_flag_error(status, gidx); return; }
                const int64_t n_adv = r < L ? (len < L - r ? len : L - r) : 0;
                if (n_adv > sl - q || (len > n_adv && q + n_adv >= sl) || (n_adv > 0 && r < -L)) {
                    if (lane == 0) kd_flag_error(status, gidx);
                    return;
                }
                if (COLD) {
                    if (lane == 0) {
                        const kd_u64 g = cb + (kd_u64)(x < 0 ? x + L + 1 : x);
                        if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)KDC_CLIP_STARTS * S + g], 1u);
                    }
                    for (int64_t j = lane; j < n_adv; j += KD_WAVE) {
                        int64_t idx = r + j;
                        if (idx < 0) idx += L;
                        const uint32_t ch = kd_chan(kd_nib(seq, q + j));
                        const kd_u64 g = cb + (kd_u64)idx;
                        if (ch == 7u) kd_flag_error(status, gidx);
                        else if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)(KDC_CSW + ch) * S + g], 1u);
                    }
                }
                r += n_adv; q += n_adv;
            }
        }
        // H, N, P, anything else: ignored entirely
    }
    (void)rinfo;
}

// k_cold_lane: what is left of the soft-clip / insertion side of REGULAR reads (kindel.py:55-58, :63-81) once
// k_window has tallied the clipped bases: the clip_ends / clip_starts counters (one 32-bit atomic each) and
// the insertion events, written into the slots k_prep reserved for the read.  One LANE per read of the cold
// list.  Regular reads cannot raise and never wrap (k_prep checked), so this is plain G-space arithmetic.
__global__ void __launch_bounds__(KD_BLOCK)
k_cold_lane(KdReads rd, KdTabs T, KdIns ins, const uint32_t *list, kd_u64 n_list, kd_u64 *status) {
    const kd_u64 slot = (kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x;
    if (slot >= n_list) return;
    const kd_u64 i = list[slot];
    const int64_t sl = rd.seq_len[i];
    const uint32_t nc = rd.n_cig[i];
    const uint32_t c = rd.contig[i];
    const int64_t L = T.contig_len[c];
    const kd_u64 cb = T.contig_base[c];
    const uint8_t *seq = rd.seq4 + rd.seq_off[i];
    const uint32_t *cg = rd.cigar + rd.cig_off[i];
    uint32_t *tab = T.tab;
    const kd_u64 S = T.stride;
    int64_t r = rd.pos0[i], q = 0;
    // everything the walk may need is requested up front (first four CIGAR words in one load, the read's event /
    // pool slots): the kernel is a chain of dependent round trips otherwise
    const KdChunk pre = kd_load_cigar4(cg, 0u, nc);
    kd_u64 ev_next = ins.read_ev[i], pool_next = ins.read_pool[i];   // (garbage for a read without insertions: unused)
    for (uint32_t k = 0; k < nc; k++) {
        const uint32_t w = k == 0 ? pre.x : k == 1 ? pre.y : k == 2 ? pre.z : k == 3 ? pre.w : cg[k];
        const int64_t len = w >> 4;
        const uint32_t op = w & 15u;
        if (op == 0 || op == 7 || op == 8) { r += len; q += len; }
        else if (op == 2) { r += len; }
        else if (op == 1) {
            const int64_t q0 = q < sl ? q : sl, q1 = q + len < sl ? q + len : sl;
            const kd_u64 n = (kd_u64)(q1 - q0);
            const kd_u64 e = ev_next, po = pool_next;
            ev_next += 1; pool_next += n;
            const kd_u64 g = cb + (kd_u64)r;  // 0 <= r <= L for a regular read
            if (e >= ins.ev_cap || po + n > ins.pool_cap) {
                atomicAdd(&status[KDS_INTERNAL], 1ULL);
            } else if (kd_commit(T, g)) {
                ins.ev_site[e] = (uint32_t)g; ins.ev_len[e] = (uint32_t)n; ins.ev_off[e] = po;
                for (kd_u64 b = 0; b < n; b++) ins.pool[po + b] = (uint8_t)kd_nib(seq, q0 + (int64_t)b);
                atomicAdd(&tab[(kd_u64)KDC_INS_TOTAL * S + g], 1u);
            } else {
                ins.ev_site[e] = KD_EV_DROPPED; ins.ev_len[e] = 0; ins.ev_off[e] = po;
            }
            q += len;
        } else if (op == 4) {
            if (k == 0) {  // kindel.py:64-73
                const kd_u64 g = cb + (kd_u64)r;
                if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)KDC_CLIP_ENDS * S + g], 1u);
                // query bases [xa, len) land on sites r - len + x  (those with r - len + x >= 0)
                // (clip_end_weights of these bases are tallied by k_window)
                q += len;
            } else {  // kindel.py:74-81; regular: the last op that touches r
                const int64_t x = r - 1;
                const kd_u64 g = cb + (kd_u64)(x < 0 ? x + L + 1 : x);
                if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)KDC_CLIP_STARTS * S + g], 1u);
                const int64_t n_adv = r < L ? (len < L - r ? len : L - r) : 0;
                // query bases [q, q + n_adv) land on sites r + (x - q)
                // clip_start_weights are tallied by k_window (LDS)
                r += n_adv; q += n_adv;
            }
        }
    }
}

// k_cold_long: k_cold_lane's work for regular long-CIGAR reads, one WORKGROUP per read: thread t starts from
// checkpoint t (state before its run of ops, incl. how many insertion events / bases precede it).
__global__ void __launch_bounds__(KD_BLOCK)
k_cold_long(KdReads rd, KdTabs T, KdIns ins, const KdRInfo *rinfo, const uint32_t *long_list, const KdCkpt *ckpt,
            kd_u64 *status) {
    const uint32_t t = threadIdx.x;
    const kd_u64 i = long_list[blockIdx.x];
    const uint32_t sc = rinfo[i].span_cls;
    if ((sc & 3u) != KD_CLS_LONG || !(sc & KD_INFO_COLD)) return;   // LONG after k_prep_long = regular long read
    const uint32_t nc = rd.n_cig[i];
    const uint32_t per = (nc + KD_BLOCK - 1) / KD_BLOCK;
    const uint32_t k0 = t * per < nc ? t * per : nc, k1 = k0 + per < nc ? k0 + per : nc;
    if (k0 >= k1) return;
    const uint32_t c = rd.contig[i];
    const int64_t L = T.contig_len[c];
    const kd_u64 cb = T.contig_base[c];
    const int64_t sl = rd.seq_len[i];
    const uint8_t *seq = rd.seq4 + rd.seq_off[i];
    const uint32_t *cg = rd.cigar + rd.cig_off[i];
    const KdCkpt ck = ckpt[(kd_u64)blockIdx.x * KD_BLOCK + t];
    int64_t r = rd.pos0[i] + (int64_t)ck.r_rel, q = ck.q;
    kd_u64 e = 0, po = 0;
    if (sc & KD_INFO_INS) { e = (kd_u64)ins.read_ev[i] + ck.ev; po = ins.read_pool[i] + ck.pool; }
    uint32_t *tab = T.tab;
    const kd_u64 S = T.stride;
    for (uint32_t k = k0; k < k1; k++) {
        const uint32_t w = cg[k];
        const int64_t len = w >> 4;
        const uint32_t op = w & 15u;
        if (op == 0 || op == 7 || op == 8) { r += len; q += len; }
        else if (op == 2) { r += len; }
        else if (op == 1) {
            const int64_t q0 = q < sl ? q : sl, q1 = q + len < sl ? q + len : sl;
            const kd_u64 n = (kd_u64)(q1 - q0);
            const kd_u64 g = cb + (kd_u64)r;
            if (e >= ins.ev_cap || po + n > ins.pool_cap) {
                atomicAdd(&status[KDS_INTERNAL], 1ULL);
            } else if (kd_commit(T, g)) {
                ins.ev_site[e] = (uint32_t)g; ins.ev_len[e] = (uint32_t)n; ins.ev_off[e] = po;
                for (kd_u64 b = 0; b < n; b++) ins.pool[po + b] = (uint8_t)kd_nib(seq, q0 + (int64_t)b);
                atomicAdd(&tab[(kd_u64)KDC_INS_TOTAL * S + g], 1u);
            } else {
                ins.ev_site[e] = KD_EV_DROPPED; ins.ev_len[e] = 0; ins.ev_off[e] = po;
            }
            e += 1; po += n; q += len;
        } else if (op == 4) {
            if (k == 0) {
                const kd_u64 g = cb + (kd_u64)r;
                if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)KDC_CLIP_ENDS * S + g], 1u);
                q += len;
            } else {  // regular: the last op that touches r
                const int64_t x = r - 1;
                const kd_u64 g = cb + (kd_u64)(x < 0 ? x + L + 1 : x);
                if (kd_commit(T, g)) atomicAdd(&tab[(kd_u64)KDC_CLIP_STARTS * S + g], 1u);
            }
        }
    }
}

// k_diagnose: one thread re-walks the first failing read serially, in the reference's own
// statement order, to decide WHICH exception the reference raises (KeyError vs IndexError
// vs RuntimeError).  Error classification only -- it writes no table.
__global__ void k_diagnose(KdReads rd, KdTabs T, kd_u64 *status) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const kd_u64 gidx = status[KDS_ERR_READ];
    if (gidx == ~0ULL || gidx < rd.base_index || gidx >= rd.base_index + rd.n) return;
    const kd_u64 i = gidx - rd.base_index;
    const int64_t sl = rd.seq_len[i];
    const uint32_t nc = rd.n_cig[i];
    const int64_t L = T.contig_len[rd.contig[i]];
    const uint8_t *seq = rd.seq4 + rd.seq_off[i];
    const uint32_t *cg = rd.cigar + rd.cig_off[i];
    kd_u64 code = 8;  // KD_E_INTERNAL magnitude: flagged but no exception reproduced
    if (nc == 0) { status[KDS_ERR_CODE] = 3; return; }
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
    status[KDS_ERR_CODE] = code;
}

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
              kd_u64 *win_lo, kd_u64 *win_hi, kd_u64 *item_off, const kd_u64 *status) {
    const uint32_t w = blockIdx.x * KD_BLOCK + threadIdx.x;   // local index; the window is w0 + w (shard-local planning)
    if (w >= n_win) return;
    const kd_u64 maxspan = status[KDS_B_MAXSPAN];
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
template <int RUN>
__global__ void __launch_bounds__(KD_BLOCK)
k_sort_count(const KdRInfo *rinfo, kd_u64 n_reads, uint32_t W, uint32_t *bin_cnt) {
    const kd_u64 i0 = ((kd_u64)blockIdx.x * KD_BLOCK + threadIdx.x) * RUN;
    uint32_t cur = 0xffffffffu, run = 0;
    for (kd_u64 i = i0; i < i0 + RUN && i < n_reads; i++) {
        const KdRInfo ri = rinfo[i];
        if ((ri.span_cls & 3u) != KD_CLS_REG) continue;
        const uint32_t b = ri.gstart / W;
        if (b != cur) {
            if (run) atomicAdd(&bin_cnt[cur], run);
            cur = b; run = 0;
        }
        run++;
    }
    if (run) atomicAdd(&bin_cnt[cur], run);
}
// one workgroup: bin_off = exclusive scan of bin_cnt (n_bins + 1 entries), bin_cnt is reset to 0 (it becomes
// the fill cursor of k_sort_scatter)
__global__ void __launch_bounds__(KD_BLOCK)
k_sort_scan(uint32_t *bin_cnt, kd_u64 *bin_off, uint32_t n_bins) {
    __shared__ kd_u64 s_scan[KD_BLOCK];
    __shared__ kd_u64 s_carry;
    const uint32_t t = threadIdx.x;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_bins; b0 += KD_BLOCK) {
        const uint32_t b = b0 + t;
        const kd_u64 v = b < n_bins ? bin_cnt[b] : 0;
        s_scan[t] = v;
        __syncthreads();
        for (uint32_t d = 1; d < KD_BLOCK; d <<= 1) {
            kd_u64 a = t >= d ? s_scan[t - d] : 0;
            __syncthreads();
            s_scan[t] += a;
            __syncthreads();
        }
        if (b < n_bins) { bin_off[b] = s_carry + s_scan[t] - v; bin_cnt[b] = 0; }
        __syncthreads();
        if (t == KD_BLOCK - 1) s_carry += s_scan[t];
        __syncthreads();
    }
    if (t == 0) bin_off[n_bins] = s_carry;
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
// candidate range of window w0 + w in `order`: whole bins covering [wlo - maxspan, whi + maxlead)
__global__ void __launch_bounds__(KD_BLOCK)
k_plan_ranges_sorted(const kd_u64 *bin_off, uint32_t n_bins, uint32_t w0, uint32_t n_win, uint32_t W, uint32_t slice,
                     kd_u64 *win_lo, kd_u64 *win_hi, kd_u64 *item_off, const kd_u64 *status, uint32_t span_slot) {
    const uint32_t w = blockIdx.x * KD_BLOCK + threadIdx.x;
    if (w >= n_win) return;
    const kd_u64 wlo = (kd_u64)(w0 + w) * W, whi = wlo + W;
    const kd_u64 maxspan = status[span_slot], maxlead = status[KDS_B_MAXLEAD];   // span_slot: KDS_B_MAXSPAN / KDS_B_MAXSEGSPAN
    const kd_u64 blo = (wlo > maxspan ? wlo - maxspan : 0) / W;
    kd_u64 bhi = (whi + maxlead + W - 1) / W;   // exclusive
    if (bhi > n_bins) bhi = n_bins;
    const kd_u64 lo = bin_off[blo < n_bins ? blo : n_bins], hi = bin_off[bhi];
    win_lo[w] = lo; win_hi[w] = hi;
    item_off[w] = (hi - lo + slice - 1) / slice;
}

// k_plan_scan: one workgroup, in-place exclusive scan of the per-window item counts.
__global__ void __launch_bounds__(KD_BLOCK)
k_plan_scan(kd_u64 *item_off, uint32_t n_win, kd_u64 *status) {
    __shared__ kd_u64 s_scan[KD_BLOCK];
    __shared__ kd_u64 s_carry;
    const uint32_t t = threadIdx.x;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t w0 = 0; w0 < n_win; w0 += KD_BLOCK) {
        const uint32_t w = w0 + t;
        const kd_u64 items = w < n_win ? item_off[w] : 0;
        s_scan[t] = items;
        __syncthreads();
        for (uint32_t d = 1; d < KD_BLOCK; d <<= 1) {
            kd_u64 a = t >= d ? s_scan[t - d] : 0;
            __syncthreads();
            s_scan[t] += a;
            __syncthreads();
        }
        if (w < n_win) item_off[w] = s_carry + s_scan[t] - items;
        __syncthreads();
        if (t == KD_BLOCK - 1) s_carry += s_scan[t];
        __syncthreads();
    }
    if (t == 0) { item_off[n_win] = s_carry; status[KDS_TOTAL_ITEMS] = s_carry; status[KDS_NEXT_ITEM] = 0; }
}

// k_plan_items: work item -> window table (k_window then needs one load, not a binary search over item_off,
// to find the window of the item it dequeued).
__global__ void __launch_bounds__(KD_BLOCK)
k_plan_items(const kd_u64 *item_off, uint32_t n_win, uint32_t *item_win, kd_u64 cap, kd_u64 *status) {
    const uint32_t w = blockIdx.x * KD_BLOCK + threadIdx.x;
    if (w >= n_win) return;
    for (kd_u64 it = item_off[w]; it < item_off[w + 1]; it++) {
        if (it < cap) item_win[it] = w;
        else status[KDS_INTERNAL] = 1;
    }
}

// k_window: persistent workgroups pull (window, slice) work items.
//
// LDS (dynamic): u32 hist[19][W], channel-major, three groups of {A,T,G,C,N, bad}: weights (0-5),
// clip_start_weights (7-12), clip_end_weights (13-18), plus deletions (6).  "bad" collects bases
// outside A,C,G,T,N (KeyError in the reference; checked at flush).  76 B per site: W = 512 lets four
// workgroups (16 wavefronts) share a CU's 160 KB.  Soft clips are tallied here too because 4 x 10^7
// scattered device-scope atomics cost more than the whole LDS pass (measured: 1.4 ms vs 1.5 ms).
//
// One LANE per read; thread t owns a contiguous run of the item's reads, so the 64 lanes of a
// wavefront sit ~16 reads apart in the coordinate-sorted batch and rarely hit the same site in the
// same instruction.  A read's packed bases are fetched with up to six 16-byte loads (all in flight
// together) into registers -- aligned 16-byte chunks as they lie in HBM, no re-alignment: memory
// dword m of the chunk holds query bases 2*(4m - mis) .. +7.  An M run is then consumed one dword
// (8 bases) at a time by fully unrolled code, every base one ds_add_u32 into hist:
//   nibble -> bfe, channel -> 64-bit LUT shift, address -> mad, ds_add with the base index as the
//   instruction's immediate offset: 6 instructions per base when the dword is fully inside the run
//   and the window, a masked variant at run / window edges.
// Reads whose bases do not fit six chunks (long reads: thousands of short ops) are walked op by op
// with dword loads straight from HBM/L2, same arithmetic.
// Only REGULAR reads are handled here; their S/I side effects are done by k_cold_lane.
#define KD_HCH 19
#define KD_HCH_DEL 6u
#define KD_HCH_CSW 7u
#define KD_HCH_CEW 13u

// BAM nibble -> channel inside a group: A,T,G,C,N -> 0..4, everything else -> 5 (the group's bad slot)
__device__ __forceinline__ uint32_t kd_hchan(uint32_t nib) {
    return (uint32_t)((0x4555555155525305ULL >> (nib * 4)) & 7ULL);
}

// The LDS histogram packs TWO sites per dword (u16 halves): a work item tallies at most `slice` <= 32768 reads
// and a read adds at most 1 to a counter, so a half cannot overflow into its neighbour.  Every channel has
// KD_HALO extra sites on both sides of the window: window clipping is done at DWORD granularity (a dword
// that straddles the window edge is added whole, its outside bases land in the halo and are never flushed),
// so the masked path below is only needed at the ends of a run -- which are the same step for all lanes of
// a wavefront of equal-length reads -- and not wherever some lane happens to cross the window edge.
// Counter of (channel ch, window-relative site s in [-KD_HALO, W + KD_HALO)) = half (s & 1) of word
// ch*Wh + (s >> 1) of `hist0` = hist + KD_HALO/2, with Wh = (W + 2*KD_HALO) / 2.
#define KD_HALO 8
__device__ __forceinline__ void kd_hadd(uint32_t *hist0, int32_t Wh, uint32_t ch, int32_t s) {
    atomicAdd(&hist0[(int32_t)KD_MUL24(ch, (uint32_t)Wh) + (s >> 1)], 1u << (16 * (s & 1)));
}
// all 8 bases of dword v are added; s0 = window-relative site of its first base.  Even bases go through pointer h
// with add value vp, odd bases through hq = h + (s0 & 1) with vq: no per-base parity arithmetic.
__device__ __forceinline__ void kd_add8_full(uint32_t *hist0, int32_t Wh, uint32_t v, int32_t s0) {
    const int32_t p = s0 & 1;
    // byte addressing: address = row base + ch * (row bytes) + constant, one 24-bit multiply-add per base
    // (v_mad_u32_u24 is full rate; a 32-bit v_mul_lo_u32 is not)
    unsigned char *h = reinterpret_cast<unsigned char *>(hist0 + (s0 >> 1));
    unsigned char *hq = h + 4 * p;
    const uint32_t rowb = (uint32_t)Wh * 4u;
    const uint32_t vp = 1u << (16 * p), vq = 0x10000u >> (16 * p);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const uint32_t ch = kd_hchan((v >> KD_NIB_SHIFT(b)) & 15u);
        unsigned char *a = ((b & 1) ? hq : h) + KD_MUL24(ch, rowb) + 4 * (b >> 1);
        atomicAdd(reinterpret_cast<uint32_t *>(a), (b & 1) ? vq : vp);
    }
}
// only bases [blo, bhi) belong to the run
__device__ __forceinline__ void kd_add8_part(uint32_t *hist0, int32_t Wh, uint32_t v, int32_t s0, int32_t blo, int32_t bhi) {
#pragma unroll
    for (int b = 0; b < 8; b++)
        if (b >= blo && b < bhi) kd_hadd(hist0, Wh, kd_hchan((v >> KD_NIB_SHIFT(b)) & 15u), s0 + b);
}
// One memory dword of a run.  xs = query index of the dword's first base; [xa, xb) = the run's query bases that
// fall inside the window (decides whether the dword is touched at all); [ra, rb) = the run's own query bases
// (decides which of its 8 bases exist); site of base x is sx + x (for a clip run sx also carries the
// channel-group offset, an even number of sites).
__device__ __forceinline__ void kd_add_dword(uint32_t *hist0, int32_t Wh, uint32_t v, int32_t xs, int32_t xa, int32_t xb,
                                             int32_t ra, int32_t rb, int32_t sx) {
    if (xs + 8 <= xa || xs >= xb) return;
    if (xs >= ra && xs + 8 <= rb) kd_add8_full(hist0, Wh, v, sx + xs);
    else kd_add8_part(hist0, Wh, v, sx + xs, ra - xs, rb - xs);
}

// General per-lane walk of one regular read against the window (reads with clips, indels, long CIGARs).
// A state machine over WORK UNITS (one 16-byte chunk = up to 32 live bases of the current run), not over
// CIGAR ops, so that lanes keep adding bases together whatever their op structure.  A soft clip is a
// run of its own on the clip_start / clip_end channel group.
// Bases of dword v whose bit is set in m (bit b = base b) are added; the others add 0 to a counter at most
// 7 sites away from a live one, i.e. inside the row (halo included).  No branches: lanes whose dword is cut
// by a run end, a clip end or the window edge stay in step with lanes whose dword is whole.
__device__ __forceinline__ void kd_add8_masked(uint32_t *hist0, int32_t Wh, uint32_t v, int32_t s0, uint32_t m) {
    const int32_t p = s0 & 1;
    unsigned char *h = reinterpret_cast<unsigned char *>(hist0 + (s0 >> 1));
    unsigned char *hq = h + 4 * p;
    const uint32_t rowb = (uint32_t)Wh * 4u;
    const uint32_t vp = 1u << (16 * p), vq = 0x10000u >> (16 * p);
    const uint32_t me = m << (16 * p), mo = m << (16 - 16 * p);   // bit b of m moved onto the add value's bit
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const uint32_t ch = kd_hchan((v >> KD_NIB_SHIFT(b)) & 15u);
        unsigned char *a = ((b & 1) ? hq : h) + KD_MUL24(ch, rowb) + 4 * (b >> 1);
        atomicAdd(reinterpret_cast<uint32_t *>(a), (b & 1) ? ((mo >> b) & vq) : ((me >> b) & vp));
    }
}
// one 16-byte chunk (query bases xs .. xs+31) against the live query range [lo, hi) of a segment
__device__ __forceinline__ void kd_add_chunk_masked(uint32_t *hist0, int32_t Wh, const KdChunk &cur, int32_t xs, int32_t lo,
                                                    int32_t hi, int32_t sx) {
    const uint32_t dw[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const int32_t x0 = xs + 8 * d;
        int32_t l = lo - x0, h = hi - x0;
        if (h <= 0 || l >= 8) continue;
        l = l < 0 ? 0 : l;
        h = h > 8 ? 8 : h;
        kd_add8_masked(hist0, Wh, dw[d], sx + x0, (0xffu >> (8 - h)) & (0xffu << l));
    }
}

// Ops [k, k_end) of read i, entered with the reference cursor at window-relative site `grel` and the query cursor
// at q: the whole CIGAR of a short read with many segments (k = 0, k_end = n_cig), or ONE SEGMENT of a long read
// (k_prep_long's checkpoint).  `lead` / `foot_end`: reach of the leading clip / window-relative end of the footprint
// (used by the clip ops, which sit in the first / last segment).
__device__ __forceinline__ void kd_walk_ops(const KdReads &rd, kd_u64 i, uint32_t k, uint32_t k_end, int32_t grel, int32_t q,
                                            int32_t lead, int32_t foot_end, int32_t Wi, int32_t Wh, uint32_t *hist0) {
    const int32_t Wp = 2 * Wh;   // sites per channel row, halos included
    const uint32_t nc = rd.n_cig[i];
    const uint32_t *cg = rd.cigar + rd.cig_off[i];
    const KdChunk *src = reinterpret_cast<const KdChunk *>(rd.seq4 + rd.seq_off[i]);
    // Per-lane state machine over WORK UNITS (one 16-byte chunk = up to 32 live bases of the
    // current M run), not over CIGAR ops: lanes whose reads have clips or indels still add
    // bases in the same wavefront instructions as their single-run neighbours.
    int32_t xa = 0, xb = 0, sx = 0, c = 1, cb = 0;   // live query range [xa, xb) of the current run; c > cb: none
    // CIGAR words four at a time (one unaligned 16-byte load, the next four already in flight), and the last
    // 16-byte chunk of bases kept: a long read's runs are a few bases each, so consecutive runs share a chunk
    // and one load per op would make the walk a chain of dependent HBM round trips.
    uint32_t kw = k & ~3u;                     // cw_cur holds words kw .. kw + 3
    KdChunk cw_cur = kd_load_cigar4(cg, kw, nc), cw_nxt = cw_cur;
    if (kw + 4 < nc) cw_nxt = kd_load_cigar4(cg, kw + 4, nc);
    int32_t c_have = -1;
    KdChunk cur = cw_cur;
    for (;;) {
        while (c > cb && k < k_end) {   // advance to the next run with live bases
            if (k >= kw + 4) {
                kw += 4; cw_cur = cw_nxt;
                if (kw + 4 < nc) cw_nxt = kd_load_cigar4(cg, kw + 4, nc);
            }
            const uint32_t kk = k & 3u;
            const uint32_t cw = kk == 0 ? cw_cur.x : kk == 1 ? cw_cur.y : kk == 2 ? cw_cur.z : cw_cur.w;
            const int32_t len = (int32_t)(cw >> 4);
            const uint32_t op = cw & 15u;
            k++;
            if (op == 0 || op == 7 || op == 8) {
                // live query range: inside the run and inside the window
                xa = grel < 0 ? q - grel : q;
                xb = Wi - grel < len ? q + (Wi - grel) : q + len;
                sx = grel - q;                      // site of query base x is sx + x
                if (xb > xa) { c = xa >> 5; cb = (xb - 1) >> 5; }
                q += len; grel += len;
                if (grel >= Wi) k = k_end;
            } else if (op == 2) {
                for (int32_t j = grel < 0 ? -grel : 0; j < len && grel + j < Wi; j++)
                    kd_hadd(hist0, Wh, KD_HCH_DEL, grel + j);
                grel += len;
                if (grel >= Wi) k = k_end;
            } else if (op == 1) {
                q += len;
            } else if (op == 4) {
                if (k == 1) {
                    // leading clip, kindel.py:64-73: base x -> site r - len + x, kept if >= contig start
                    // (`lead` of the len bases); a run on the clip_end_weights channels
                    const int32_t s_first = grel - len;           // site of base 0
                    xa = -s_first > len - lead ? -s_first : len - lead;
                    xb = Wi - s_first < len ? Wi - s_first : len;
                    sx = s_first + (int32_t)KD_HCH_CEW * Wp;
                    if (xb > xa) { c = xa >> 5; cb = (xb - 1) >> 5; }
                    q += len;
                } else {
                    // non-first clip, kindel.py:74-81: bases q.. -> sites r.. while r < L; for a regular
                    // read it is the last op that moves r, so its reach is the end of the footprint
                    const int32_t n_adv = foot_end - grel;
                    xa = grel < 0 ? q - grel : q;
                    xb = Wi - grel < n_adv ? q + (Wi - grel) : q + n_adv;
                    sx = grel - q + (int32_t)KD_HCH_CSW * Wp;
                    if (xb > xa) { c = xa >> 5; cb = (xb - 1) >> 5; }
                    k = k_end;
                }
            }
        }
        if (c > cb) break;
        if (c != c_have) { cur = src[c]; c_have = c; }
        const int32_t xs = 32 * c;
        kd_add_chunk_masked(hist0, Wh, cur, xs, xa, xb, sx);   // [xa, xb) lies inside the run: branch-free masked adds
        c++;
    }
}

// SHORT regular reads with clips / indels (at most KD_PREP_MAX_OPS ops, the bulk of the non-plain reads of a
// short-read batch).  Two phases: (1) decode the CIGAR into at most three SEGMENTS -- runs of query bases that
// land on consecutive sites of one channel group: an M/=/X run, the leading clip (clip_end_weights), the non-first
// clip (clip_start_weights) -- each already cut to the window; deletions are tallied on the way; (2) one flat,
// software-pipelined loop over (segment, 16-byte chunk) steps, every step a branch-free masked add, so that the
// lanes of a wavefront stay in step whatever their op structure.  Returns false (nothing added) when the read
// has more than three segments: the caller then takes the general walk.
__device__ __forceinline__ bool kd_walk_short(const KdReads &rd, kd_u64 i, const KdRInfo ri, kd_u64 wlo, int32_t Wi, int32_t Wh,
                                              uint32_t *hist0) {
    const uint32_t nc = rd.n_cig[i];
    const uint32_t *cg = rd.cigar + rd.cig_off[i];
    // the first four CIGAR words, all in flight together (a short read rarely has more)
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    if (nc > 0) w0 = cg[0];
    if (nc > 1) w1 = cg[1];
    if (nc > 2) w2 = cg[2];
    if (nc > 3) w3 = cg[3];
    const KdChunk *src = reinterpret_cast<const KdChunk *>(rd.seq4 + rd.seq_off[i]);
    uint32_t n_seg_ops = 0;
    for (uint32_t k = 0; k < nc; k++) {
        const uint32_t cw = k == 0 ? w0 : k == 1 ? w1 : k == 2 ? w2 : k == 3 ? w3 : cg[k];
        const uint32_t op = cw & 15u;
        n_seg_ops += (op == 0 || op == 7 || op == 8 || op == 4) ? 1u : 0u;
    }
    if (n_seg_ops > 3) return false;
    const int32_t Wp = 2 * Wh;   // sites per channel row, halos included
    const kd_u64 gs = ri.gstart, span = ri.span_cls >> KD_SPAN_SHIFT;
    const int32_t lead = (int32_t)ri.lead;
    const int32_t foot_end = (int32_t)((uint32_t)(gs + span) - (uint32_t)wlo);  // window-relative end of the footprint
    int32_t grel = (int32_t)((uint32_t)gs - (uint32_t)wlo);                        // window-relative site, may be negative
    int32_t q = 0;
    // segment slots: live query range [a, b) and site offset x (site of query base j is x + j, channel group included)
    int32_t a0 = 0, b0 = 0, x0 = 0, a1 = 0, b1 = 0, x1 = 0, a2 = 0, b2 = 0, x2 = 0;
    uint32_t ns = 0;
    for (uint32_t k = 0; k < nc; k++) {
        const uint32_t cw = k == 0 ? w0 : k == 1 ? w1 : k == 2 ? w2 : k == 3 ? w3 : cg[k];
        const int32_t len = (int32_t)(cw >> 4);
        const uint32_t op = cw & 15u;
        int32_t xa = 0, xb = 0, sx = 0;
        if (op == 0 || op == 7 || op == 8) {
            xa = grel < 0 ? q - grel : q;
            xb = Wi - grel < len ? q + (Wi - grel) : q + len;
            sx = grel - q;
            q += len; grel += len;
        } else if (op == 2) {
            for (int32_t j = grel < 0 ? -grel : 0; j < len && grel + j < Wi; j++)
                kd_hadd(hist0, Wh, KD_HCH_DEL, grel + j);
            grel += len;
        } else if (op == 1) {
            q += len;
        } else if (op == 4) {
            if (k == 0) {   // leading clip, kindel.py:64-73: base j -> site r - len + j, the last `lead` bases are kept
                const int32_t s_first = grel - len;
                xa = -s_first > len - lead ? -s_first : len - lead;
                xb = Wi - s_first < len ? Wi - s_first : len;
                sx = s_first + (int32_t)KD_HCH_CEW * Wp;
                q += len;
            } else {        // non-first clip, kindel.py:74-81: it is the last op that moves r (regular read)
                const int32_t n_adv = foot_end - grel;
                xa = grel < 0 ? q - grel : q;
                xb = Wi - grel < n_adv ? q + (Wi - grel) : q + n_adv;
                sx = grel - q + (int32_t)KD_HCH_CSW * Wp;
                k = nc;
            }
        }
        if (xb > xa) {
            if (ns == 0) { a0 = xa; b0 = xb; x0 = sx; }
            else if (ns == 1) { a1 = xa; b1 = xb; x1 = sx; }
            else { a2 = xa; b2 = xb; x2 = sx; }
            ns++;
        }
        if (grel >= Wi) break;   // everything further right is outside the window
    }
    if (ns == 0) return true;
    int32_t c = a0 >> 5, cb = (b0 - 1) >> 5;
    KdChunk cur = src[c];
    for (;;) {
        // the step after this one: next chunk of the segment, or the first chunk of the next segment
        const bool adv = c + 1 > cb;
        const bool more = !adv || ns > 1;
        const int32_t cn = adv ? (a1 >> 5) : c + 1;
        KdChunk nxt = cur;
        if (more) nxt = src[cn];
        kd_add_chunk_masked(hist0, Wh, cur, 32 * c, a0, b0, x0);
        if (!more) break;
        if (adv) {
            a0 = a1; b0 = b1; x0 = x1; a1 = a2; b1 = b2; x1 = x2;
            ns--;
            cb = (b0 - 1) >> 5;
        }
        c = cn;
        cur = nxt;
    }
    return true;
}

// A PLAIN read: one M/=/X run covering the whole read, no clips (k_prep: KD_INFO_PLAIN).  Nothing to decode:
// query base x lands on site grel + x, for x in [0, span).
__device__ __forceinline__ void kd_walk_plain(const KdReads &rd, kd_u64 i, const KdRInfo ri, kd_u64 wlo, int32_t Wi,
                                              int32_t Wh, uint32_t *hist0) {
    const int32_t grel = (int32_t)(ri.gstart - (uint32_t)wlo);
    const int32_t len = (int32_t)(ri.span_cls >> KD_SPAN_SHIFT);
    const int32_t xa = grel < 0 ? -grel : 0;
    const int32_t xb = Wi - grel < len ? Wi - grel : len;
    if (xb <= xa) return;
    const KdChunk *src = reinterpret_cast<const KdChunk *>(rd.seq4 + rd.seq_off[i]);
    const int32_t ca = xa >> 5, cb = (xb - 1) >> 5;
    // three chunks of prefetch: a 150-base read is 5 chunks, so its loads are (almost) all in flight at once
    KdChunk cur = src[ca], n1 = cur, n2 = cur;
    if (ca + 1 <= cb) n1 = src[ca + 1];
    if (ca + 2 <= cb) n2 = src[ca + 2];
    for (int32_t c = ca; c <= cb; c++) {
        KdChunk n3 = n2;
        if (c + 3 <= cb) n3 = src[c + 3];
        const int32_t xs = 32 * c;
        kd_add_dword(hist0, Wh, cur.x, xs, xa, xb, 0, len, grel);
        kd_add_dword(hist0, Wh, cur.y, xs + 8, xa, xb, 0, len, grel);
        kd_add_dword(hist0, Wh, cur.z, xs + 16, xa, xb, 0, len, grel);
        kd_add_dword(hist0, Wh, cur.w, xs + 24, xa, xb, 0, len, grel);
        cur = n1; n1 = n2; n2 = n3;
    }
}

#define KD_TILE 1024   // reads classified together (a multiple of KD_BLOCK)
#define KD_TILE_PER_THREAD (KD_TILE / KD_BLOCK)
#define KD_WINDOW_LDS_BYTES(Wh) ((size_t)KD_HCH * (Wh) * 4 + (size_t)2 * KD_TILE * 2)   // Wh = dwords per channel row

__global__ void __launch_bounds__(KD_BLOCK, 5)   // 5 wavefronts per SIMD = the 5 workgroups per CU the LDS footprint allows
k_window(KdReads rd, const KdRInfo *rinfo, const uint32_t *order, const KdCkpt *ckpt, const uint32_t *seg_read, KdTabs T,
         const kd_u64 *win_lo, const kd_u64 *win_hi, const kd_u64 *item_off, const uint32_t *item_win, kd_u64 items_cap, uint32_t w0,
         uint32_t W, uint32_t Wh_, uint32_t slice, kd_u64 *status) {
    // seg_read == NULL: `rinfo` describes the batch's reads (first pass, class REG = short regular reads).
    // seg_read != NULL: `rinfo` describes SEGMENTS of long reads (k_prep_long; entry e = 256 * b + t is thread t's
    // run of ops of the long read seg_read[b], entered through checkpoint ckpt[e]); `order` is then never NULL.
    KD_DYN_SHARED(uint32_t, hist);
    const int32_t Wh = (int32_t)Wh_;   // dwords per channel row (two u16 counters each, halos included; >= (W + 2*KD_HALO)/2)
    uint32_t *hist0 = hist + KD_HALO / 2;                 // word of window-relative site 0
    uint16_t *l_plain = reinterpret_cast<uint16_t *>(hist + (size_t)KD_HCH * Wh);  // tile-relative read indices
    uint16_t *l_cplx = l_plain + KD_TILE;
    __shared__ kd_u64 s_item;
    __shared__ uint32_t s_cnt[2][2];   // [tile parity][plain, complex] list lengths
    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & (KD_WAVE - 1), wave = t / KD_WAVE;
    const kd_u64 total = status[KDS_TOTAL_ITEMS];
    const uint32_t nh = (uint32_t)KD_HCH * (uint32_t)Wh;   // histogram dwords
    const int32_t Wi = (int32_t)W;
#ifdef KD_PHASE_CLOCKS
    long long c_zero = 0, c_cls = 0, c_plain = 0, c_cplx = 0, c_wait = 0, c_flush = 0, c_deq = 0, c_mark;
#define KD_MARK(acc) { const long long n_ = clock64(); acc += n_ - c_mark; c_mark = n_; }
    c_mark = clock64();
#else
#define KD_MARK(acc)
#endif
    for (;;) {
        if (t == 0) { s_item = atomicAdd(&status[KDS_NEXT_ITEM], 1ULL); s_cnt[0][0] = 0; s_cnt[0][1] = 0; }
        __syncthreads();
        const kd_u64 item = s_item;
        if (item >= total || item >= items_cap) break;   // (>= items_cap: k_plan_items has raised KDS_INTERNAL)
        KD_MARK(c_deq)
        const uint32_t w = item_win[item];   // k_plan_items: the window with item_off[w] <= item < item_off[w + 1]
        const kd_u64 wlo = (kd_u64)(w0 + w) * W, whi = wlo + W;
        const kd_u64 first = win_lo[w] + (item - item_off[w]) * slice;
        const kd_u64 last = first + slice < win_hi[w] ? first + slice : win_hi[w];
        // The classification keys (start, span | flags, lead) of a tile are fetched ONE TILE AHEAD into registers:
        // the loads of tile k + 1 are in flight while tile k is walked.  `order`: bucket-sorted permutation.
        uint32_t p_gs[KD_TILE_PER_THREAD], p_sc[KD_TILE_PER_THREAD], p_ld[KD_TILE_PER_THREAD];
#pragma unroll
        for (uint32_t u = 0; u < KD_TILE_PER_THREAD; u++) {
            const kd_u64 j = first + u * KD_BLOCK + t;
            p_sc[u] = KD_CLS_SKIP; p_gs[u] = 0; p_ld[u] = 0;
            if (j < last) {
                const KdRInfo ri = rinfo[order ? (kd_u64)order[j] : j];
                p_gs[u] = ri.gstart; p_sc[u] = ri.span_cls; p_ld[u] = ri.lead;
            }
        }
        {   // Wh is a multiple of 4 (W is a multiple of 64): zero with 16-byte stores
            uint4 *h4 = reinterpret_cast<uint4 *>(hist);
            for (uint32_t x = t; x < nh / 4; x += KD_BLOCK) h4[x] = make_uint4(0u, 0u, 0u, 0u);
        }
        KD_MARK(c_zero)
        uint32_t par = 0;
        for (kd_u64 tb = first; tb < last; tb += KD_TILE, par ^= 1u) {
            // classify the tile's reads: plain (single aligned run) / complex; drop those outside the window
#pragma unroll
            for (uint32_t u = 0; u < KD_TILE_PER_THREAD; u++) {
                const kd_u64 gs = p_gs[u], span = p_sc[u] >> KD_SPAN_SHIFT;
                if ((p_sc[u] & 3u) == KD_CLS_REG && gs + span > wlo && gs - p_ld[u] < whi) {
                    const uint32_t rel = u * KD_BLOCK + t;
                    if (p_sc[u] & KD_INFO_PLAIN) l_plain[atomicAdd(&s_cnt[par][0], 1u)] = (uint16_t)rel;
                    else l_cplx[atomicAdd(&s_cnt[par][1], 1u)] = (uint16_t)rel;
                }
            }
            __syncthreads();
            KD_MARK(c_cls)
            const uint32_t np = s_cnt[par][0], ncx = s_cnt[par][1];
            if (t == 0) { s_cnt[par ^ 1u][0] = 0; s_cnt[par ^ 1u][1] = 0; }   // next tile's counters (idle until its classify)
#pragma unroll
            for (uint32_t u = 0; u < KD_TILE_PER_THREAD; u++) {
                const kd_u64 j = tb + KD_TILE + u * KD_BLOCK + t;
                p_sc[u] = KD_CLS_SKIP;
                if (j < last) {
                    const KdRInfo ri = rinfo[order ? (kd_u64)order[j] : j];
                    p_gs[u] = ri.gstart; p_sc[u] = ri.span_cls; p_ld[u] = ri.lead;
                }
            }
            // homogeneous wavefronts: first the plain reads, then the complex ones.  Lane l of a wavefront takes
            // list entries l*rows + r: neighbours in a wavefront are `rows` reads apart in the sorted batch,
            // which keeps them off the same LDS counters in the same instruction.
            const uint32_t rows_p = (np + KD_WAVE - 1) / KD_WAVE, rows_c = (ncx + KD_WAVE - 1) / KD_WAVE;
            for (uint32_t r = wave; r < rows_p; r += KD_WAVES_PER_BLOCK) {
                const uint32_t e = lane * rows_p + r;
                if (e < np) {
                    const kd_u64 j = tb + l_plain[e], i = order ? (kd_u64)order[j] : j;
                    kd_walk_plain(rd, i, rinfo[i], wlo, Wi, Wh, hist0);
                }
            }
            KD_MARK(c_plain)
            // the complex rows start at the wavefront after the one that took the last plain row
            for (uint32_t r = (wave + KD_WAVES_PER_BLOCK - rows_p % KD_WAVES_PER_BLOCK) % KD_WAVES_PER_BLOCK; r < rows_c;
                 r += KD_WAVES_PER_BLOCK) {
                const uint32_t e = lane * rows_c + r;
                if (e < ncx) {
                    const kd_u64 j = tb + l_cplx[e], i = order ? (kd_u64)order[j] : j;
                    const KdRInfo ri = rinfo[i];
                    const int32_t grel = (int32_t)(ri.gstart - (uint32_t)wlo);
                    const int32_t foot_end = grel + (int32_t)(ri.span_cls >> KD_SPAN_SHIFT);
                    if (seg_read) {          // one segment of a long read
                        const kd_u64 ir = seg_read[i / KD_BLOCK];
                        const uint32_t nc = rd.n_cig[ir], per = (nc + KD_BLOCK - 1) / KD_BLOCK;
                        const uint32_t k0 = (uint32_t)(i % KD_BLOCK) * per, k1 = k0 + per < nc ? k0 + per : nc;
                        kd_walk_ops(rd, ir, k0, k1, grel, (int32_t)ckpt[i].q, (int32_t)ri.lead, foot_end, Wi, Wh, hist0);
                    } else if (!kd_walk_short(rd, i, ri, wlo, Wi, Wh, hist0)) {   // more than three segments: general walk
                        kd_walk_ops(rd, i, 0u, rd.n_cig[i], grel, 0, (int32_t)ri.lead, foot_end, Wi, Wh, hist0);
                    }
                }
            }
            KD_MARK(c_cplx)
            __syncthreads();
            KD_MARK(c_wait)
        }
        // flush: channel-major, consecutive lanes -> consecutive HBM dwords; zeros are skipped.
        // LDS channel -> table channel (KD_CH_*): weights 0-4, deletions 5, csw 6-10, cew 11-15; 0xff = bad slot
        bool bad = false;
        uint32_t ch = 0, xw = t;   // word x = ch * Wh + xw, kept without a division
        for (uint32_t x = t; x < nh; x += KD_BLOCK, xw += KD_BLOCK) {
            while (xw >= (uint32_t)Wh) { xw -= (uint32_t)Wh; ch++; }
            const uint32_t v = hist[x];
            if (v) {
                const uint32_t tch = ch < 5 ? ch : ch == KD_HCH_DEL ? (uint32_t)KDC_DEL
                                   : (ch >= 7 && ch < 12) ? ch - 1 : (ch >= 13 && ch < 18) ? ch - 2 : 0xffu;
                // the word holds window-relative sites s (low half) and s + 1 (high half); halo sites are dropped
                const int32_t sw = 2 * (int32_t)xw - KD_HALO;
                uint32_t *row = T.tab + (kd_u64)(tch == 0xffu ? 0u : tch) * T.stride;
                const kd_u64 g0 = wlo + (kd_u64)sw;   // even: W, the halo and the G-space rows are all even / 8-byte aligned
                if (tch != 0xffu && sw >= 0 && sw + 1 < Wi && g0 + 1 < T.stride && kd_commit(T, g0) && kd_commit(T, g0 + 1)) {
                    // both sites of the word live: ONE 64-bit add on the two adjacent u32 counters (the low counter
                    // cannot carry into the high one: a u32 table counter never wraps)
                    atomicAdd(reinterpret_cast<kd_u64 *>(row + g0), (kd_u64)(v & 0xffffu) | ((kd_u64)(v >> 16) << 32));
                    continue;
                }
                for (int hlf = 0; hlf < 2; hlf++) {
                    const uint32_t cnt = hlf ? v >> 16 : v & 0xffffu;
                    const int32_t sw2 = sw + hlf;
                    if (!cnt || sw2 < 0 || sw2 >= Wi) continue;
                    const kd_u64 g = wlo + (kd_u64)sw2;
                    if (tch == 0xffu) bad = true;
                    else if (g < T.stride && kd_commit(T, g)) atomicAdd(&row[g], cnt);
                }
            }
        }
        // a base outside A,C,G,T,N inside an aligned or clipped segment: k_find_bad_base pins down the read
        if (bad) atomicAdd(&status[KDS_BAD_BASE], 1ULL);
        KD_MARK(c_flush)
        __syncthreads();
        KD_MARK(c_wait)
    }
#ifdef KD_PHASE_CLOCKS
    if ((t & 63u) == 0) {   // lane 0 of every wavefront
        atomicAdd(&status[KDS_DBG0], (kd_u64)c_deq); atomicAdd(&status[KDS_DBG1], (kd_u64)c_zero);
        atomicAdd(&status[KDS_DBG2], (kd_u64)c_cls); atomicAdd(&status[KDS_DBG3], (kd_u64)c_plain);
        atomicAdd(&status[KDS_DBG4], (kd_u64)c_cplx); atomicAdd(&status[KDS_DBG5], (kd_u64)c_wait);
        atomicAdd(&status[KDS_DBG6], (kd_u64)c_flush); atomicAdd(&status[KDS_DBG7], 1ULL);
    }
#endif
}

// Rare path: k_window saw a base outside A,C,G,T,N.  One workgroup walks the regular reads of the
// batch and records the first offender (atomicMin of the read index), for k_diagnose to classify.
__global__ void __launch_bounds__(KD_BLOCK)
k_find_bad_base(KdReads rd, KdTabs T, const KdRInfo *rinfo, kd_u64 *status) {
    if (status[KDS_BAD_BASE] == 0) return;
    for (kd_u64 i = threadIdx.x; i < rd.n; i += KD_BLOCK) {
        const uint32_t cls_i = rinfo[i].span_cls & 3u;
        if (cls_i != KD_CLS_REG && cls_i != KD_CLS_LONG) continue;   // regular reads, short and long
        if (rd.base_index + i >= status[KDS_ERR_READ]) continue;
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
        if (found) kd_flag_error(status, rd.base_index + i);
    }
}

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
    if (g >= T.stride || g < C.g_lo || g >= C.g_hi) return s;
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
    const kd_u64 S = T.stride;
#pragma unroll
    for (int c = 0; c < 7; c++) {
        const uint32_t *row = T.tab + (kd_u64)chs[c] * S;
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
    const uint32_t cfirst = tile0 < T.stride ? C.seg_contig[tile0 >> 6] : 0;
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
        if (g >= T.stride) break;
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
