// kd_ingest.h -- device-side ingest of a BGZF-compressed BAM file: what parse_bam's record iteration
// (kindel.py:131-153, through simplesam + samtools in the reference; kd_decode.cpp on the host here) hands the record loop, built
// on the GPU from the FILE's bytes: k_gpu_inflate (kd_gpu_inflate.h) inflates every BGZF block, the kernels below walk the BAM
// records of the inflated stream and write the structure-of-arrays batch (kd_batch) that kd_push_batch_device takes.
// Part of the device code of kd_kernels.h (included from there, in order; not a stand-alone header).
//
// The block_size chain of BAM records is sequential by nature (a record's length says where the next one begins).  It is walked
// in parallel from SPECULATIVE, VERIFIED starts, the rule of the host's parallel walk (kd_decode.cpp: parse_bam_records,
// kd_decode_open_span): the part of the chain that starts inside the inflated bytes of BGZF block b is walked by one thread, from
// the first offset in the block from which KD_BAM_CHAIN consecutive records look well-formed (k_bam_starts: 64 candidates per step
// and wavefront); a walk must END exactly on the start the next part guessed (k_bam_count), otherwise the ingest reports
// KD_INGEST_CHAIN and the caller decodes on the host.  Records of unmapped reads (refID < 0) are dropped, as the host decoder
// drops them; a CIGAR placeholder with a CG:B,I tag (> 65535 operations, SAMv1 4.2.2) is followed to the tag's array
// (kd_bam_real_cigar, round 4; rounds 1 - 3 sent such a file to the host decoder).
//
//   k_bam_starts   wavefront per block   -> start[b]                      (first record that starts in block b, or NONE)
//   k_bam_count    thread per block      -> kept records / packed-base bytes / CIGAR words of block b's part, chain check
//   k_bam_scan     one workgroup         -> exclusive prefix sums over the blocks, totals
//   k_bam_fields   thread per block      -> contig, pos0, flag, seq_len, n_cig, seq_off, cig_off of every kept record + where it lies
//   k_bam_payload  wavefront per 64 records -> packed bases (last odd nibble cleared) and CIGAR words, copied by all lanes
#pragma once
#include "kd_common.h"
#include "kd_gpu_inflate.h"

#define KD_BAM_CHAIN 16
#define KD_BAM_NONE (~0ULL)
// bits of the ingest status word
#define KD_INGEST_INFLATE 1u   // a block did not inflate
#define KD_INGEST_RECORD 2u    // a malformed / truncated record on a walked chain
#define KD_INGEST_CHAIN 4u     // a walk did not end on the next part's start (a start was guessed wrong)
#define KD_INGEST_HOST 8u      // (unused since round 4: CG-tag CIGARs are read here)

struct KdBam {
    const uint8_t *d;        // the inflated stream
    kd_u64 n;                // its length
    kd_u64 hdr_end;          // offset of the first record
    const GiBlock *blocks;   // out_off / out_len: the part of the stream every BGZF block holds
    uint32_t n_blocks, n_ref;
};

__device__ __forceinline__ uint32_t kd_rd32(const uint8_t *p) { return reinterpret_cast<const GiU32 *>(p)->v; }

// Does d[q, n) begin with a well-formed BAM record?  -> the offset of the record behind it, 0 if not.  The host's rule
// (kd_decode.cpp: bam_record_plausible: block_size, refID, pos, l_read_name, the lengths fitting, the read name NUL-terminated)
// and more, because here a start is guessed for every 64 KiB block, not for every megabyte, and a wrong guess costs the whole
// device-side ingest: the mate's refID / pos in range, and the AUXILIARY FIELDS must parse -- tag letters, a known type, every
// value inside the record -- and end exactly where block_size says.  (Measured need: on a 5 Mbp x 50 x file without base
// qualities -- every record the same shape, runs of 0xff -- the host's rule accepted the offset two bytes in front of a true
// record in one block of 6 857: block_size 0x010fffff, refID 0, a NUL where the shifted name length pointed, and sixteen hops
// on from there.  The shifted mate refID is 0xffff0000, and 17 MB of other records do not parse as tags.)
__device__ __forceinline__ kd_u64 kd_bam_plausible(const KdBam &B, kd_u64 q) {
    if (q + 36 > B.n) return 0;
    const uint32_t bs = kd_rd32(B.d + q);
    if (bs < 32 || q + 4 + (kd_u64)bs > B.n) return 0;
    const uint8_t *r = B.d + q + 4;
    const int32_t refid = (int32_t)kd_rd32(r), pos = (int32_t)kd_rd32(r + 4);
    const uint32_t w2 = kd_rd32(r + 8), w3 = kd_rd32(r + 12), l_seq = kd_rd32(r + 16);
    const int32_t m_refid = (int32_t)kd_rd32(r + 20), m_pos = (int32_t)kd_rd32(r + 24);
    const uint32_t l_rn = w2 & 0xffu, n_cig = w3 & 0xffffu;
    if (refid < -1 || (refid >= 0 && (uint32_t)refid >= B.n_ref) || pos < -1 || l_rn == 0) return 0;
    if (m_refid < -1 || (m_refid >= 0 && (uint32_t)m_refid >= B.n_ref) || m_pos < -1) return 0;
    kd_u64 a = 32ull + l_rn + 4ull * n_cig + ((kd_u64)l_seq + 1) / 2 + (kd_u64)l_seq;
    if (a > bs) return 0;
    if (r[32 + l_rn - 1] != 0) return 0;        // read name is NUL-terminated
    while (a < bs) {                            // auxiliary fields: TAG (2 letters / digits), type, value
        if (a + 3 > bs) return 0;
        const uint32_t t0 = r[a], t1 = r[a + 1], ty = r[a + 2];
        const bool alpha0 = (t0 | 32u) - 'a' < 26u, alnum1 = (t1 | 32u) - 'a' < 26u || t1 - '0' < 10u;
        if (!alpha0 || !alnum1) return 0;
        a += 3;
        kd_u64 len;
        if (ty == 'A' || ty == 'c' || ty == 'C') len = 1;
        else if (ty == 's' || ty == 'S') len = 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') len = 4;
        else if (ty == 'Z' || ty == 'H') { len = 0; while (a + len < bs && r[a + len]) len++; if (a + len >= bs) return 0; len++; }
        else if (ty == 'B') {
            if (a + 5 > bs) return 0;
            const uint32_t sub = r[a], cnt = kd_rd32(r + a + 1);
            const kd_u64 es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
            if (!es) return 0;
            len = 5 + es * (kd_u64)cnt;
        } else return 0;
        a += len;
        if (a > bs) return 0;
    }
    return q + 4 + bs;
}

__global__ void __launch_bounds__(KD_WAVE)
k_bam_starts(KdBam B, kd_u64 *start) {
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (b >= B.n_blocks) return;
    const kd_u64 lo = B.blocks[b].out_off, hi = lo + B.blocks[b].out_len;
    if (hi <= B.hdr_end || lo >= B.n) { if (lane == 0) start[b] = KD_BAM_NONE; return; }      // header bytes only
    if (lo <= B.hdr_end) { if (lane == 0) start[b] = B.hdr_end < B.n ? B.hdr_end : KD_BAM_NONE; return; }   // the first record is given
    kd_u64 found = KD_BAM_NONE;
    for (kd_u64 base = lo; base < hi && found == KD_BAM_NONE; base += KD_WAVE) {
        const kd_u64 q = base + lane;
        bool ok = q < hi;
        kd_u64 p = q;
        for (int k = 0; ok && k < KD_BAM_CHAIN; k++) {
            const kd_u64 nx = kd_bam_plausible(B, p);
            if (!nx) ok = false;
            else if (nx == B.n) break;          // the stream ends exactly behind this record
            p = nx;
        }
        const unsigned long long m = kd_ballot(ok);
        if (m) found = base + (kd_u64)__builtin_ctzll(m);
    }
    if (lane == 0) start[b] = found;
}

// the block whose part of the stream holds offset q (blocks are in stream order)
__device__ __forceinline__ uint32_t kd_bam_block_of(const KdBam &B, kd_u64 q) {
    uint32_t lo = 0, hi = B.n_blocks;           // last block with out_off <= q
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (B.blocks[mid].out_off <= q) lo = mid; else hi = mid; }
    return lo;
}

// One step of a walk: the record at q (block_size field at q).  false = malformed / cut off.
struct KdBamRec { uint32_t bs, l_rn, n_cig, l_seq, flag; int32_t refid, pos; };
__device__ __forceinline__ bool kd_bam_read(const KdBam &B, kd_u64 q, KdBamRec &R) {
    if (q + 4 > B.n) return false;
    R.bs = kd_rd32(B.d + q);
    if (R.bs < 32 || q + 4 + (kd_u64)R.bs > B.n) return false;
    const uint8_t *r = B.d + q + 4;
    R.refid = (int32_t)kd_rd32(r); R.pos = (int32_t)kd_rd32(r + 4);
    const uint32_t w2 = kd_rd32(r + 8), w3 = kd_rd32(r + 12);
    R.l_rn = w2 & 0xffu; R.n_cig = w3 & 0xffffu; R.flag = w3 >> 16; R.l_seq = kd_rd32(r + 16);
    if (R.refid >= 0) {
        if ((uint32_t)R.refid >= B.n_ref) return false;
        if (32ull + R.l_rn + 4ull * R.n_cig + ((kd_u64)R.l_seq + 1) / 2 > R.bs) return false;
    }
    return true;
}

// The CIGAR of the record at q (R = its fixed fields): count and absolute offset of its words.  A read with more than 65535
// operations is stored with the placeholder <l_seq>S<ref_len>N in the record and the real CIGAR in a CG:B,I tag (SAMv1 4.2.2): the
// auxiliary fields are walked for it (the host decoder's rule, kd_decode.cpp: real_cigar); without the tag the placeholder stands.
__device__ __forceinline__ uint32_t kd_bam_real_cigar(const KdBam &B, kd_u64 q, const KdBamRec &R, kd_u64 &at) {
    const uint8_t *r = B.d + q + 4;
    at = q + 4 + 32 + R.l_rn;
    if (R.n_cig != 2) return R.n_cig;
    const uint32_t c0 = kd_rd32(r + 32 + R.l_rn), c1 = kd_rd32(r + 32 + R.l_rn + 4);
    if ((c0 & 15u) != 4u || (c0 >> 4) != R.l_seq || (c1 & 15u) != 3u) return R.n_cig;
    const kd_u64 bs = R.bs;
    kd_u64 a = 32ull + R.l_rn + 8ull + ((kd_u64)R.l_seq + 1) / 2 + (kd_u64)R.l_seq;
    while (a + 3 <= bs) {
        const uint8_t t0 = r[a], t1 = r[a + 1], ty = r[a + 2];
        a += 3;
        kd_u64 len = 0;
        if (ty == 'A' || ty == 'c' || ty == 'C') len = 1;
        else if (ty == 's' || ty == 'S') len = 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') len = 4;
        else if (ty == 'Z' || ty == 'H') { while (a + len < bs && r[a + len]) len++; len++; }
        else if (ty == 'B') {
            if (a + 5 > bs) return R.n_cig;
            const uint8_t sub = r[a];
            const uint32_t cnt = kd_rd32(r + a + 1);
            const kd_u64 es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            if (t0 == 'C' && t1 == 'G' && sub == 'I' && a + 5 + 4ull * cnt <= bs) { at = q + 4 + a + 5; return cnt; }
            len = 5 + es * (kd_u64)cnt;
        } else return R.n_cig;   // unknown type: cannot skip it
        a += len;
    }
    return R.n_cig;
}

__global__ void __launch_bounds__(KD_BLOCK)
k_bam_count(KdBam B, const kd_u64 *start, const uint32_t *inflate_status, kd_u64 *cnt_rec, kd_u64 *cnt_seq, kd_u64 *cnt_cig, kd_u64 *n_seen,
            uint32_t *status) {
    const uint32_t b = blockIdx.x * KD_BLOCK + threadIdx.x;
    if (b >= B.n_blocks) return;
    if (inflate_status[b] != GI_OK) atomicOr(status, KD_INGEST_INFLATE);
    kd_u64 kept = 0, sb = 0, cw = 0, seen = 0;
    const kd_u64 s = start[b], hi = B.blocks[b].out_off + B.blocks[b].out_len;
    if (s != KD_BAM_NONE) {
        kd_u64 q = s;
        while (q < hi) {
            KdBamRec R;
            if (!kd_bam_read(B, q, R)) { atomicOr(status, KD_INGEST_RECORD); q = B.n; break; }
            seen++;
            if (R.refid >= 0) {
                kd_u64 cg_at;
                kept++; sb += ((kd_u64)R.l_seq + 1) / 2; cw += kd_bam_real_cigar(B, q, R, cg_at);
            }
            q += 4 + (kd_u64)R.bs;
        }
        // the walk ended on the first record that starts behind this block: the part that begins there must have guessed it
        // -- and a block that a record longer than a block skips whole (b < c < nb) must have guessed NOTHING: a start found there
        // (record-shaped bytes inside a long record's qualities or auxiliary data) would be walked as phantom reads that neither the
        // host decoder nor the reference ever sees.  Either way the file goes to the host decoder (KD_INGEST_CHAIN).
        if (q > B.n) atomicOr(status, KD_INGEST_RECORD);
        else {
            const uint32_t nb = q < B.n ? kd_bam_block_of(B, q) : B.n_blocks;
            bool chain_ok = q == B.n || start[nb] == q;
            for (uint32_t c = b + 1; c < nb && chain_ok; c++) chain_ok = start[c] == KD_BAM_NONE;
            if (!chain_ok) { atomicOr(status, KD_INGEST_CHAIN); atomicMax((kd_u64 *)(status + 2), ((kd_u64)(b + 1) << 32) | (nb + 1)); }   // (diagnosis: which hand-off)
        }
    }
    cnt_rec[b] = kept; cnt_seq[b] = sb; cnt_cig[b] = cw;
    if (seen) atomicAdd(n_seen, seen);
}

// exclusive prefix sums of three arrays over the blocks (in place), tot[0..2] = the totals; one workgroup
__global__ void __launch_bounds__(KD_SCAN_WIDE)
k_bam_scan(kd_u64 *a0, kd_u64 *a1, kd_u64 *a2, uint32_t n, kd_u64 *tot) {
    __shared__ kd_u64 s_wave[KD_SCAN_WIDE / KD_WAVE];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + KD_SCAN_WIDE - 1) / KD_SCAN_WIDE;
    const uint32_t b0 = t * per < n ? t * per : n, b1 = b0 + per < n ? b0 + per : n;
    kd_u64 *arr[3] = {a0, a1, a2};
    for (int k = 0; k < 3; k++) {
        kd_u64 mine = 0;
        for (uint32_t b = b0; b < b1; b++) mine += arr[k][b];
        kd_u64 total;
        kd_u64 o = kd_block_scan_incl_wide(mine, s_wave, total) - mine;
        for (uint32_t b = b0; b < b1; b++) { const kd_u64 v = arr[k][b]; arr[k][b] = o; o += v; }
        if (t == 0) tot[k] = total;
    }
}

struct KdBamOut {
    uint32_t *contig; int32_t *pos0; uint32_t *flag; kd_u64 *seq_off; uint32_t *seq_len; kd_u64 *cig_off; uint32_t *n_cig;
    uint8_t *seq4; uint32_t *cigar;
    kd_u64 *rec_at;          // scratch: offset of every kept record's packed bases in the stream,
    kd_u64 *cig_at;          //   and of its CIGAR words (in the record, or the array of its CG:B,I tag)
};

__global__ void __launch_bounds__(KD_BLOCK)
k_bam_fields(KdBam B, const kd_u64 *start, const kd_u64 *rec_base, const kd_u64 *seq_base, const kd_u64 *cig_base, KdBamOut O) {
    const uint32_t b = blockIdx.x * KD_BLOCK + threadIdx.x;
    if (b >= B.n_blocks) return;
    const kd_u64 s = start[b], hi = B.blocks[b].out_off + B.blocks[b].out_len;
    if (s == KD_BAM_NONE) return;
    kd_u64 k = rec_base[b], so = seq_base[b], co = cig_base[b];
    for (kd_u64 q = s; q < hi;) {
        KdBamRec R;
        if (!kd_bam_read(B, q, R)) return;       // (k_bam_count has reported it)
        if (R.refid >= 0) {
            kd_u64 cg_at;
            const uint32_t nc = kd_bam_real_cigar(B, q, R, cg_at);
            O.contig[k] = (uint32_t)R.refid; O.pos0[k] = R.pos; O.flag[k] = R.flag; O.seq_len[k] = R.l_seq; O.n_cig[k] = nc;
            O.seq_off[k] = so; O.cig_off[k] = co; O.rec_at[k] = q + 4 + 32 + R.l_rn + 4ull * R.n_cig; O.cig_at[k] = cg_at;
            k++; so += ((kd_u64)R.l_seq + 1) / 2; co += nc;
        }
        q += 4 + (kd_u64)R.bs;
    }
}

// 64 records per wavefront: their CIGAR words and packed bases are contiguous in the outputs (offsets are prefix sums), so the
// lanes copy the wavefront's whole output range element by element, finding each element's record by a binary search over
// the 64 records' output offsets (LDS).
__global__ void __launch_bounds__(KD_WAVE)
k_bam_payload(KdBam B, kd_u64 n_rec, KdBamOut O, kd_u64 seq_total, kd_u64 cig_total) {
    __shared__ kd_u64 s_so[KD_WAVE + 1], s_co[KD_WAVE + 1], s_src[KD_WAVE], s_csrc[KD_WAVE];
    __shared__ uint32_t s_odd[KD_WAVE];
    const uint32_t lane = threadIdx.x;
    const kd_u64 i0 = (kd_u64)blockIdx.x * KD_WAVE, i = i0 + lane;
    const uint32_t cnt = (uint32_t)(n_rec - i0 < KD_WAVE ? n_rec - i0 : KD_WAVE);
    if (lane < cnt) {
        s_so[lane] = O.seq_off[i]; s_co[lane] = O.cig_off[i];
        s_src[lane] = O.rec_at[i];                    // the record's packed bases
        s_csrc[lane] = O.cig_at[i];                   // its CIGAR words
        s_odd[lane] = O.seq_len[i] & 1u;
    }
    if (lane == 0) {   // the end of the wavefront's output ranges
        s_so[cnt] = i0 + cnt < n_rec ? O.seq_off[i0 + cnt] : seq_total;
        s_co[cnt] = i0 + cnt < n_rec ? O.cig_off[i0 + cnt] : cig_total;
    }
    KD_WAVE_SYNC();
    auto find = [&](const kd_u64 *pre, kd_u64 x) -> uint32_t {   // last record j < cnt with pre[j] <= x (records without output are skipped)
        uint32_t lo = 0, hi = cnt;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (pre[mid] <= x) lo = mid; else hi = mid; }
        return lo;
    };
    for (kd_u64 x = s_co[0] + lane; x < s_co[cnt]; x += KD_WAVE) {
        const uint32_t j = find(s_co, x);
        O.cigar[x] = kd_rd32(B.d + s_csrc[j] + 4 * (x - s_co[j]));
    }
    for (kd_u64 x = s_so[0] + lane; x < s_so[cnt]; x += KD_WAVE) {
        const uint32_t j = find(s_so, x);
        uint8_t v = B.d[s_src[j] + (x - s_so[j])];
        if (s_odd[j] && x + 1 == s_so[j + 1]) v &= 0xf0u;      // the unused low nibble behind an odd-length read
        O.seq4[x] = v;
    }
}
