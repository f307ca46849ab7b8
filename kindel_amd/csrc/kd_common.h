// kd_common.h -- shared by all kernels: status words, table / read / insertion descriptors, small helpers.
// Part of the device code of kd_kernels.h (included from there, in order; not a stand-alone header).
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
#define KD_MUL24S(a, b) __mul24((a), (b))   // signed (|operands| < 2^23)
#endif

// Wavefront operations (64 lanes, gfx950).  tests/emu/hip_emu.h supplies functional stand-ins (KD_EMU).
#ifndef KD_EMU
// lane-private LDS writes -> cross-lane LDS reads inside ONE wavefront: the LDS queue of a wavefront is in order, so
// this only has to stop the compiler from moving accesses across it (no s_barrier, no other wavefront involved)
#define KD_WAVE_SYNC()                                          \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
__device__ __forceinline__ unsigned long long kd_ballot(bool pred) { return __ballot(pred); }
__device__ __forceinline__ uint32_t kd_lane_id() { return __lane_id(); }
// number of set bits of `mask` below this lane (v_mbcnt_lo/hi)
__device__ __forceinline__ uint32_t kd_mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t kd_shfl(uint32_t v, unsigned src_lane) { return (uint32_t)__shfl((int)v, (int)src_lane, 64); }
__device__ __forceinline__ unsigned long long kd_shfl64(unsigned long long v, unsigned src_lane) {
    return ((unsigned long long)kd_shfl((uint32_t)(v >> 32), src_lane) << 32) | kd_shfl((uint32_t)v, src_lane);
}
__device__ __forceinline__ uint32_t kd_shfl_up(uint32_t v, unsigned d) { return (uint32_t)__shfl_up((int)v, d, 64); }
__device__ __forceinline__ unsigned long long kd_shfl_up64(unsigned long long v, unsigned d) {
    return ((unsigned long long)kd_shfl_up((uint32_t)(v >> 32), d) << 32) | kd_shfl_up((uint32_t)v, d);
}
__device__ __forceinline__ uint32_t kd_shfl_xor(uint32_t v, unsigned m) { return (uint32_t)__shfl_xor((int)v, (int)m, 64); }
__device__ __forceinline__ uint32_t kd_readfirstlane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long kd_readfirstlane64(unsigned long long v) {
    return ((unsigned long long)kd_readfirstlane((uint32_t)(v >> 32)) << 32) | kd_readfirstlane((uint32_t)v);
}
__device__ __forceinline__ uint32_t kd_readlane(uint32_t v, unsigned src_lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src_lane); }   // src_lane: a constant
__device__ __forceinline__ int kd_popcll(unsigned long long m) { return __popcll(m); }
// OR over the 64 lanes (wave-uniform result): an inclusive DPP scan inside each row of 16 lanes, two cross-row broadcasts, v_readlane 63
__device__ __forceinline__ uint32_t kd_wave_or(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8: lane 15 of a row = the row's OR
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// Inclusive prefix sum over the 64 lanes with DPP adds (row_shr 1 / 2 / 4 / 8 inside each row of 16 lanes, then the last lane
// of a row broadcast into the rows behind it): six VALU instructions.  __shfl_up is a ds_bpermute_b32 per step and operand
// -- the LDS crossbar, 2.6 ns per CU-instruction (profiles/valu_issue_calibration.json) -- and kernels that scan per tile of
// 256 CIGAR ops spent most of their time there.
__device__ __forceinline__ uint32_t kd_wave_scan_add(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}
// running maximum over the lanes (inclusive), the same six DPP steps
__device__ __forceinline__ uint32_t kd_wave_scan_max(uint32_t v) {
#define KD_DPP_MAX(ctrl, rows) { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, false); v = o_ > v ? o_ : v; }
    KD_DPP_MAX(0x111, 0xf) KD_DPP_MAX(0x112, 0xf) KD_DPP_MAX(0x114, 0xf) KD_DPP_MAX(0x118, 0xf) KD_DPP_MAX(0x142, 0xa) KD_DPP_MAX(0x143, 0xc)
#undef KD_DPP_MAX
    return v;
}
// v_perm_b32: byte i of the result = byte sel.byte[i] of {hi (bytes 4-7), lo (bytes 0-3)}
__device__ __forceinline__ uint32_t kd_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// v_alignbyte_b32: ({hi, lo} >> 8 * (sh & 3)) [31:0]
__device__ __forceinline__ uint32_t kd_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
// a word another workgroup of the SAME kernel publishes (k_window's work queue): device-scope acquire load; the pause keeps a
// waiting wavefront off the issue slots of the ones it waits for
__device__ __forceinline__ unsigned long long kd_ld_acquire(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void kd_spin_pause() { __builtin_amdgcn_s_sleep(8); }
#endif

// Profiling hooks of k_window (where its wavefronts spend their clocks).  EMPTY in the product; a profiling build force-includes
// scripts/exp/kd_phase_clocks.h (hipcc -include ...), which defines them and KD_PHASE_CLOCKS (eight more status words).  The
// product headers hold no measurement-only code paths: the wrong-on-purpose attribution builds of rounds 2 - 3 (packed bases
// from cache, flush skipped, ...) are recorded in profiles/ and gone from the tree.
#ifndef KD_PHASE_DECL
#define KD_PHASE_DECL
#define KD_MARK(acc)
#define KD_PHASE_COMMIT(status, rows)
#define KD_PHASE_REPORT(h)
// ... and of k_prep (when its wavefronts start, leave their loop and end: what a small batch's floor is made of)
#define KD_PREP_CLK_DECL
#define KD_PREP_CLK_LOOP_END
#define KD_PREP_CLK_COMMIT(status)
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

// device status words (kd_u64 each), KDS_STRIDE words apart: every wavefront of k_prep ends with a dozen atomics on them, and
// atomics on different words of ONE 128-byte line queue up behind each other like atomics on one word (~4.4 ns each measured:
// 8136 wavefronts x 14 atomics = 0.5 ms).  One line per counter: status[KDS_X] with KDS_X = ordinal KDO_X * KDS_STRIDE.
#ifndef KDS_STRIDE
#define KDS_STRIDE 16
#endif
enum {
    KDO_ERR_READ = 0,   // smallest global index of a failing read (init ~0): "some read failed"; WHICH exception is
                        // reported is decided per contig (KdTabs::err_first / err_code)
    KDO_ERR_CODE,       // (unused)
    KDO_N_EV,           // insertion events used
    KDO_POOL,           // insertion pool bytes used
    KDO_ST_READS,       // reads counted
    KDO_ST_ALIGNED,     // aligned-base events
    KDO_ST_WALKED,      // walked events
    KDO_ST_INS,         // insertion ops seen
    KDO_B_INS_OPS,      // per batch: insertion ops
    KDO_B_INS_BASES,    // per batch: insertion bases
    KDO_B_MAXSPAN,      // per batch: max span of regular reads
    KDO_B_MAXLEAD,      // per batch: max leading-clip reach of regular reads
    KDO_B_MAXSEGSPAN,   // per batch: longest ROW of a long read (k_long_reduce; k_window's second pass), in sites
    KDO_B_ROW_DWORDS,   // per batch: dwords of the row buffer handed out to the long reads (k_long_reduce)
    KDO_B_UNSORTED,     // per batch: reads not sorted by G-start
    KDO_B_N_COLD,       // per batch: entries in the cold list
    KDO_B_N_IRREG,      // per batch: entries in the irregular list
    KDO_B_N_LONG,       // per batch: entries in the long-CIGAR list
    KDO_B_N_REG,        // per batch: regular reads
    KDO_B_FILL,         // per batch: entries of k_window's boundary table written by gap fills (k_prep's budget)
    KDO_B_BOUND_LO,     // per batch: the boundary table's first written entry (the granule of the batch's first read): everything in front
                        //   of it reads as that entry (0)
    KDO_B_BOUND_HI1,    // per batch: 1 + the table's last written entry (behind the granule of the batch's last read: "no more reads"):
                        //   everything behind it reads as that entry; 0 = the table's own last entry
    KDO_WQ_LEFT,        // k_window's work queue: hot windows that still have slices nobody has taken
    KDO_WQ_TICKET,      // k_window's self-planned work queue (kd_window.h): window tickets handed out,
    KDO_WQ_PUB,         //   windows whose owner has published its slice count,
    KDO_WQ_HOT,         //   entries of the list of windows with more than one slice
    KDO_NEXT_ITEM,      // window work queue head (planned queues: k_window_coop, k_strip)
    KDO_TOTAL_ITEMS,    // window work queue length
    KDO_INS_COLLISION,  // hash verification failed
    KDO_INTERNAL,       // capacity overrun etc.
    KDO_BAD_BASE,       // k_window / k_strip: work items that saw a base outside A,C,G,T,N
    KDO_QUEUE0,         // k_strip: heads of the eight work queues (reset by k_plan_scan)
    KDO_QUEUE7 = KDO_QUEUE0 + 7,
#ifdef KD_PHASE_CLOCKS
    KDO_DBG0, KDO_DBG1, KDO_DBG2, KDO_DBG3, KDO_DBG4, KDO_DBG5, KDO_DBG6, KDO_DBG7,   // phase clocks (profiling build only)
    KDO_DBG8, KDO_DBG9, KDO_DBG10, KDO_DBG11, KDO_DBG12, KDO_DBG13,                   // k_prep's wavefront times (the same)
#endif
    KDO_COUNT
};
#define KDS_ERR_READ (KDO_ERR_READ * KDS_STRIDE)
#define KDS_ERR_CODE (KDO_ERR_CODE * KDS_STRIDE)
#define KDS_N_EV (KDO_N_EV * KDS_STRIDE)
#define KDS_POOL (KDO_POOL * KDS_STRIDE)
#define KDS_ST_READS (KDO_ST_READS * KDS_STRIDE)
#define KDS_ST_ALIGNED (KDO_ST_ALIGNED * KDS_STRIDE)
#define KDS_ST_WALKED (KDO_ST_WALKED * KDS_STRIDE)
#define KDS_ST_INS (KDO_ST_INS * KDS_STRIDE)
#define KDS_B_INS_OPS (KDO_B_INS_OPS * KDS_STRIDE)
#define KDS_B_INS_BASES (KDO_B_INS_BASES * KDS_STRIDE)
#define KDS_B_MAXSPAN (KDO_B_MAXSPAN * KDS_STRIDE)
#define KDS_B_MAXLEAD (KDO_B_MAXLEAD * KDS_STRIDE)
#define KDS_B_MAXSEGSPAN (KDO_B_MAXSEGSPAN * KDS_STRIDE)
#define KDS_B_ROW_DWORDS (KDO_B_ROW_DWORDS * KDS_STRIDE)
#define KDS_B_UNSORTED (KDO_B_UNSORTED * KDS_STRIDE)
#define KDS_B_N_COLD (KDO_B_N_COLD * KDS_STRIDE)
#define KDS_B_N_IRREG (KDO_B_N_IRREG * KDS_STRIDE)
#define KDS_B_N_LONG (KDO_B_N_LONG * KDS_STRIDE)
#define KDS_B_N_REG (KDO_B_N_REG * KDS_STRIDE)
#define KDS_B_FILL (KDO_B_FILL * KDS_STRIDE)
#define KDS_B_BOUND_LO (KDO_B_BOUND_LO * KDS_STRIDE)
#define KDS_B_BOUND_HI1 (KDO_B_BOUND_HI1 * KDS_STRIDE)
#define KDS_WQ_LEFT (KDO_WQ_LEFT * KDS_STRIDE)
#define KDS_WQ_TICKET (KDO_WQ_TICKET * KDS_STRIDE)
#define KDS_WQ_PUB (KDO_WQ_PUB * KDS_STRIDE)
#define KDS_WQ_HOT (KDO_WQ_HOT * KDS_STRIDE)
#define KDS_NEXT_ITEM (KDO_NEXT_ITEM * KDS_STRIDE)
#define KDS_TOTAL_ITEMS (KDO_TOTAL_ITEMS * KDS_STRIDE)
#define KDS_INS_COLLISION (KDO_INS_COLLISION * KDS_STRIDE)
#define KDS_INTERNAL (KDO_INTERNAL * KDS_STRIDE)
#define KDS_BAD_BASE (KDO_BAD_BASE * KDS_STRIDE)
#define KDS_QUEUE0 (KDO_QUEUE0 * KDS_STRIDE)
#define KDS_DBG0 (KDO_DBG0 * KDS_STRIDE)
#define KDS_COUNT (KDO_COUNT * KDS_STRIDE)
#define KDS_QUEUE7 (KDO_QUEUE7 * KDS_STRIDE)


struct KdTabs {
    uint32_t *tab;               // counter of (channel ch, G-site g) = tab[ch * stride + g].  The allocation only covers the
                                 // shard's sites [alloc_lo, alloc_hi) (+ slack); `tab` is biased by -alloc_lo so that kernels
                                 // index with global sites.  Every access is either guarded by kd_commit() or made by the
                                 // consensus tiles of the shard, which lie inside the allocation.
    kd_u64 stride;               // dwords between channels (the shard's site count + slack)
    kd_u64 sites;                // S = total G-space sites (multiple of 1024): bound of valid g
    const uint32_t *contig_len;  // [n_contigs]
    const kd_u64 *contig_base;   // [n_contigs]
    kd_u64 g_lo, g_hi;           // commit increments with g_lo <= g <= g_hi (g_hi = halo site)
    // The reference walks the records contig by contig, contigs in order of first appearance (kindel.py:143-151), so the
    // exception it raises is that of the first failing read of the EARLIEST-APPEARING contig that has one:
    kd_u64 *first_idx;           // [n_contigs] global index of the contig's first record (k_prep), ~0 = none yet
    kd_u64 *err_first;           // [n_contigs] global index of the contig's first failing read, ~0 = none
    uint32_t *err_code;          // [n_contigs] its exception (k_errors): 1 KeyError, 2 IndexError, 3 RuntimeError
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
    kd_u64 n_cigar;     // words in `cigar` (>= 4: the engine pads a smaller array; k_prep loads a read's first four words at once)
    uint32_t osh;       // 0: seq_off / cig_off (and the KdRInfo array beside them) are the batch's own arrays; 1: they are
                        // the fields of KdSortRec[] (an unsorted batch's regular reads in window order): KD_RI / KD_SOFF / KD_COFF
};

struct KdRInfo {
    uint32_t gstart;    // contig_base + max(pos0, 0)
    uint32_t span_cls;  // span << KD_SPAN_SHIFT | KD_INFO_INS | KD_INFO_COLD | class; span = sites from
                        // gstart to the end of the last M / D / trailing-S write
    uint32_t lead;      // sites before gstart written by a leading soft clip (kindel.py:68-72)
    uint32_t pad;       // long-CIGAR reads: 1 + index of the read in the long list; regular short-CIGAR reads: query length
                        // (< 2^20) | CIGAR words << 24; 0 otherwise
};

// An unsorted batch's regular read in window order (k_sort_scatter_recs): everything the window walk needs of it in ONE
// 32-byte record -- one scattered write per read when sorting, one sector per read when walking.  (A regular read's
// CIGAR length sits in ri.pad >> 24.)
struct alignas(16) KdSortRec { KdRInfo ri; kd_u64 seq_off, cig_off; };
#define KD_RI(rinfo, rd, i) ((rinfo)[(kd_u64)(i) << (rd).osh])
#define KD_SOFF(rd, i) ((rd).seq_off[(kd_u64)(i) << ((rd).osh * 2u)])
#define KD_COFF(rd, i) ((rd).cig_off[(kd_u64)(i) << ((rd).osh * 2u)])

// Long-CIGAR reads (> KD_PREP_MAX_OPS words; kd_long.h).  A regular long read is EXPANDED into a ROW: one 4-bit symbol per
// reference site of its footprint (packed like BAM bases: high nibble first), so that the window pass tallies it as ONE plain
// run whatever its CIGAR looks like -- an ONT read has an op every ~7 bases, and walking those op by op left most lanes idle.
// Row symbols = LDS histogram channels of k_window<ROWS>:
#define KD_ROW_SKIP 0u      // nothing aligned here (padding, the slot behind the last site, a base outside A,C,G,T,N)
#define KD_ROW_DEL 6u       // a deleted site                      (1..5 = A,T,G,C,N in the reference's dict order: 1 + kd_chan)
#define KD_ROW_INS 7u       // added to the symbol of a site that an insertion precedes: SKIP+ins 7, A+ins .. N+ins 8..12, DEL+ins 13

// A regular short-CIGAR read with a soft clip or an insertion, as k_cold_lane needs it.  k_prep has all of this in
// registers when it classifies the read; k_cold_lane, one lane per such read (one read in nine on C3), would gather it again
// through six arrays behind an index list: a chain of dependent scattered loads.  Every wavefront of k_prep owns a region
// of the record array; the lanes whose read qualifies take consecutive slots of it (ballot + mbcnt) and store.
struct alignas(16) KdColdRec {
    kd_u64 cig_off;
    uint32_t read;      // index of the read in the batch
    uint32_t pos0;      // >= 0 for a regular read
    uint32_t contig;
    uint32_t len_ops;   // seq_len (< 2^20: longer short-CIGAR reads take the general path) | n_cig << 20 | KD_COLD_HAS_INS
    uint32_t ev_rel;    // reads with insertions: first event slot / first pool byte, relative to the region's base
    uint32_t pool_rel;  //   (ev_base[region], pool_base[region]: written by k_prep once the block has reserved its range)
};
#define KD_COLD_HAS_INS 0x80000000u
#define KD_COLD_MAX_SEQ (1u << 20)

// What k_prep_long learned about one long read (one workgroup each): summed into the status words and turned into event /
// pool / irregular-list slots by ONE small kernel (k_long_reduce) instead of ten same-address atomics per workgroup.
struct KdLongAcc {
    kd_u64 aligned, walked, insb;
    uint32_t n_ins;
    uint32_t lead;      // leading-clip reach of a regular read
    uint32_t row_span;  // regular: symbols of its row = M / D footprint + 1 (the slot behind the last site takes trailing insertions)
    uint32_t regular;   // 1: stays class LONG, 0: irregular (goes to irreg_list)
    uint32_t clip_adv;  // regular: sites a trailing soft clip writes behind the footprint (clip_start_weights)
    uint32_t pad;
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


// Inclusive scan of two values per thread over the workgroup in THREE barriers: 16 threads scan 16-element segments
// serially in LDS, then every thread adds the totals of the segments before its own.  (A Hillis-Steele scan is 8 steps
// of two barriers each; the kernels that scan are latency chains already.)
// sa / sb: [KD_BLOCK] scratch, ga / gb: [KD_BLOCK / 16] scratch; returns the inclusive prefixes and the block totals.
#define KD_SCAN_SEG 16
template <typename TA, typename TB>
__device__ __forceinline__ void kd_block_scan2(TA *sa, TB *sb, TA *ga, TB *gb, TA va, TB vb, TA &incl_a, TB &incl_b, TA &tot_a,
                                               TB &tot_b) {
    const uint32_t t = threadIdx.x;
    sa[t] = va; sb[t] = vb;
    __syncthreads();
    if (t < KD_BLOCK / KD_SCAN_SEG) {
        TA a = 0; TB b = 0;
        for (uint32_t k = 0; k < KD_SCAN_SEG; k++) {
            a += sa[KD_SCAN_SEG * t + k]; sa[KD_SCAN_SEG * t + k] = a;
            b += sb[KD_SCAN_SEG * t + k]; sb[KD_SCAN_SEG * t + k] = b;
        }
        ga[t] = a; gb[t] = b;
    }
    __syncthreads();
    TA oa = 0, ta = 0; TB ob = 0, tb = 0;
    for (uint32_t j = 0; j < KD_BLOCK / KD_SCAN_SEG; j++) {
        const TA xa = ga[j]; const TB xb = gb[j];
        if (j < t / KD_SCAN_SEG) { oa += xa; ob += xb; }
        ta += xa; tb += xb;
    }
    incl_a = sa[t] + oa; incl_b = sb[t] + ob; tot_a = ta; tot_b = tb;
    __syncthreads();   // the scratch arrays may be reused by the caller
}

// Runs of neighbouring lanes aiming at the SAME key (reads sorted by position meeting at one deep site: ten thousand
// atomics on one address take 0.1 ms however idle the rest of the GPU is).  The first lane of each run acts for all of
// them: returns the number of lanes in this lane's run if it is the run's head, 0 otherwise; head_lane = the lane that
// acts for this lane.  Every lane of the wavefront must call (wave-level exchange inside).
__device__ __forceinline__ uint32_t kd_run_heads(bool valid, kd_u64 key, uint32_t &head_lane) {
    const uint32_t lane = kd_lane_id();
    const kd_u64 amask = kd_ballot(valid);
    const kd_u64 below = amask & ((1ULL << lane) - 1ULL);
    const uint32_t prev = below ? 63u - (uint32_t)__builtin_clzll(below) : lane;
    const kd_u64 kp = kd_shfl64(key, prev);
    const bool head = valid && (!below || kp != key);
    const kd_u64 hmask = kd_ballot(head);
    const kd_u64 upto = hmask & ((2ULL << lane) - 1ULL);       // heads at or below this lane
    head_lane = upto ? 63u - (uint32_t)__builtin_clzll(upto) : lane;
    if (!head) return 0;
    const kd_u64 above = hmask & ~((2ULL << lane) - 1ULL);     // the next run's head, if any
    const kd_u64 range = above ? (1ULL << __builtin_ctzll(above)) - 1ULL : ~0ULL;
    return (uint32_t)kd_popcll(amask & range & ~((1ULL << lane) - 1ULL));
}
// Inclusive scan of one 64-bit value per thread over the 256-thread workgroup: six __shfl_up steps inside each wavefront,
// the four wavefront totals through LDS (two barriers instead of the sixteen of a Hillis-Steele scan in LDS).
// s_wave: [KD_WAVES_PER_BLOCK] scratch.  Returns the inclusive prefix; total = sum over the workgroup.
__device__ __forceinline__ kd_u64 kd_block_scan_incl(kd_u64 v, kd_u64 *s_wave, kd_u64 &total) {
    const uint32_t lane = threadIdx.x & (KD_WAVE - 1), wave = threadIdx.x / KD_WAVE;
#pragma unroll
    for (uint32_t d = 1; d < KD_WAVE; d <<= 1) {
        const kd_u64 t = kd_shfl_up64(v, d);
        if (lane >= d) v += t;
    }
    if (lane == KD_WAVE - 1) s_wave[wave] = v;
    __syncthreads();
    kd_u64 off = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < KD_WAVES_PER_BLOCK; w++) { const kd_u64 x = s_wave[w]; if (w < wave) off += x; tot += x; }
    total = tot;
    __syncthreads();   // s_wave may be reused by the caller
    return v + off;
}
// the same for a workgroup of KD_SCAN_WIDE threads (16 wavefronts): the one-workgroup scans of a step (k_plan_scan, k_sort_scan,
// k_bam_scan; C3: k_plan_scan 0.032 -> 0.022 ms) are latency chains -- every thread walks its own contiguous run of the array -- and 1024 threads
// make the runs a quarter as long.  s_wave: [KD_SCAN_WIDE / KD_WAVE].
#define KD_SCAN_WIDE 1024
__device__ __forceinline__ kd_u64 kd_block_scan_incl_wide(kd_u64 v, kd_u64 *s_wave, kd_u64 &total) {
    const uint32_t lane = threadIdx.x & (KD_WAVE - 1), wave = threadIdx.x / KD_WAVE;
#pragma unroll
    for (uint32_t d = 1; d < KD_WAVE; d <<= 1) {
        const kd_u64 t = kd_shfl_up64(v, d);
        if (lane >= d) v += t;
    }
    if (lane == KD_WAVE - 1) s_wave[wave] = v;
    __syncthreads();
    kd_u64 off = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < KD_SCAN_WIDE / KD_WAVE; w++) { const kd_u64 x = s_wave[w]; if (w < wave) off += x; tot += x; }
    total = tot;
    __syncthreads();   // s_wave may be reused by the caller
    return v + off;
}
// min / max / sum over the 64 lanes by butterfly shuffles (every lane gets the result)
__device__ __forceinline__ uint32_t kd_wave_min(uint32_t v) {
#pragma unroll
    for (uint32_t m = 1; m < KD_WAVE; m <<= 1) { const uint32_t t = kd_shfl_xor(v, m); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ uint32_t kd_wave_max(uint32_t v) {
#pragma unroll
    for (uint32_t m = 1; m < KD_WAVE; m <<= 1) { const uint32_t t = kd_shfl_xor(v, m); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ uint32_t kd_wave_sum(uint32_t v) {
#pragma unroll
    for (uint32_t m = 1; m < KD_WAVE; m <<= 1) v += kd_shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ void kd_flag_error(const KdTabs &T, kd_u64 *status, uint32_t contig, kd_u64 gidx) {
    atomicMin(&T.err_first[contig], gidx);
    atomicMin(&status[KDS_ERR_READ], gidx);
}
