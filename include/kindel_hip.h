/*
 * kindel_hip.h -- C-ABI of libkindel_hip.so, the MI355X (gfx950) pileup + consensus engine.
 *
 * The reference (bede/kindel, /root/reference) is pure Python and has no FFI layer; its
 * de-facto boundary for the hot path is the pair of Python calls
 *     parse_records(ref_id, ref_len, records) -> alignment     kindel/kindel.py:21-128
 *     consensus_sequence(weights, insertions, deletions, cdr_patches, trim_ends,
 *                        min_depth, uppercase) -> (str, changes) kindel/kindel.py:384-430
 * called from parse_bam (:131-153) and bam_to_consensus (:488-555).  Every entry point
 * below names the reference lines it replaces.  The Python host mirror that keeps the
 * reference's API on top of this ABI is kindel_amd/kindel.py (ctypes binding in
 * kindel_amd/_native.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - return 0 on success, a negative KD_E_* code otherwise; kd_last_error() gives text.
 *   - one kd_ctx = one GPU + one HIP stream; a context is NOT thread-safe, distinct
 *     contexts (one per GPU / per process) are independent.
 *   - "G-space": all contigs are laid out back to back on one site axis,
 *     g = kd_contig_base(ctx, c) + site; each contig owns len+1 slots (the reference keeps
 *     len+1 slots for deletions / clip_starts / clip_ends / insertions, kindel.py:36-39)
 *     rounded up to a multiple of 64.
 */
#ifndef KINDEL_HIP_H
#define KINDEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KD_ABI_VERSION 2

/* error codes; the Python mirror re-raises the exception the reference would raise */
#define KD_OK 0
#define KD_E_BASE (-1)   /* KeyError   : base outside A,C,G,T,N in M/=/X or clip   kindel.py:52,72,79 */
#define KD_E_RANGE (-2)  /* IndexError : read/indel/clip runs off the contig        kindel.py:51-52,57,61,67,75 */
#define KD_E_CIGAR (-3)  /* RuntimeError: mapped read, real SEQ, CIGAR '*'          kindel.py:47 */
#define KD_E_HIP (-4)    /* HIP runtime failure (incl. no device)                            */
#define KD_E_NOMEM (-5)
#define KD_E_ARG (-6)    /* bad argument / bad state                                         */
#define KD_E_IO (-7)     /* decoder: unreadable or malformed SAM/BAM                         */
#define KD_E_INTERNAL (-8)
#define KD_E_UNSUPPORTED (-10) /* kd_push_bam_gpu: this file needs the host decoder (CG-tag CIGARs, a chain the device walk could not verify) */
#define KD_E_NOREF (-9)  /* KeyError   : a record names a reference without @SQ line   kindel.py:151 (message = the name) */

/* table channels of kd_get_tables(); order of the reference's dicts is A,T,G,C,N (kindel.py:29) */
enum {
    KD_CH_A = 0, KD_CH_T = 1, KD_CH_G = 2, KD_CH_C = 3, KD_CH_N = 4, /* weights            :29    */
    KD_CH_DEL = 5,                                                    /* deletions          :39    */
    KD_CH_CSW = 6,  /* ..10  clip_start_weights A,T,G,C,N                                   :30-32 */
    KD_CH_CEW = 11, /* ..15  clip_end_weights   A,T,G,C,N                                   :33-35 */
    KD_CH_CLIP_STARTS = 16,                                           /* clip_starts        :36    */
    KD_CH_CLIP_ENDS = 17,                                             /* clip_ends          :37    */
    KD_CH_INS_TOTAL = 18, /* sum(insertions[site].values())                                 :402   */
    KD_NCH = 19
};

/* pileup kernel selection for kd_set_mode() */
#define KD_MODE_AUTO 0   /* windowed LDS histograms (first pass: k_window), falling back per read where needed */
#define KD_MODE_GLOBAL 1 /* one wavefront per read, 32-bit atomics straight into HBM     */
#define KD_MODE_WINDOW 2 /* force the windowed path (first pass by k_window, one lane per read)                     */
#define KD_MODE_STRIP 3  /* windowed planning, first pass by the site-major kernel k_strip (wavefront per 64-site
                            strip, counters in registers, no LDS atomics): an independent second implementation  */
#define KD_MODE_COOP 4   /* windowed path, first pass by k_window_coop (loads one lane per read, tile staged in LDS, walk with
                            16 lanes per read on conflict-free LDS rows): bit-identical, measured slower on MI355X
                            (DESIGN.md section 3), kept as an independent implementation                               */

typedef struct kd_ctx kd_ctx;

/*
 * A batch of decoded reads, structure-of-arrays, BAM-native encodings
 * (what simplesam.Reader hands parse_bam, kindel.py:136-145, minus the text).
 * Records with RNAME '*' must already be dropped (kindel.py:147-148); everything else --
 * unmapped-but-placed, secondary, supplementary, SEQ '*' -- is passed through and the
 * engine applies the reference's own skip rule (flag & 4, len(seq) <= 1; kindel.py:43-46).
 */
typedef struct kd_batch {
    uint64_t n_reads;
    const uint32_t *contig;  /* [n] index into the contig table given to kd_create          */
    const int32_t *pos0;     /* [n] POS-1 (kindel.py:42); -1 when SAM POS is 0              */
    const uint32_t *flag;    /* [n] SAM FLAG                                                */
    const uint64_t *seq_off; /* [n] byte offset of the read's first base in seq4            */
    const uint32_t *seq_len; /* [n] l_seq in bases (0 for SEQ '*')                          */
    const uint64_t *cig_off; /* [n] index of the read's first CIGAR word in cigar           */
    const uint32_t *n_cig;   /* [n] number of CIGAR words (0 for CIGAR '*')                 */
    const uint8_t *seq4;     /* 4-bit bases "=ACMGRSVTWYHKDBN", high nibble first, each read
                                starts on a byte boundary (BAM layout)                      */
    uint64_t seq4_bytes;
    const uint32_t *cigar;   /* len<<4 | op, op in "MIDNSHP=X" (BAM layout)                 */
    uint64_t cigar_words;
} kd_batch;

/* ---- lifetime -------------------------------------------------------------------- */

/* Replaces the table allocation of parse_records, kindel.py:29-39.  Allocates and zeroes
 * KD_NCH u32 channels over G-space on `device`.  stream: a hipStream_t to launch on
 * (e.g. torch's current stream) or NULL to create a private one. */
int kd_create(kd_ctx **out, int device, uint32_t n_contigs, const uint32_t *contig_len,
              void *stream);
void kd_destroy(kd_ctx *ctx);
const char *kd_last_error(const kd_ctx *ctx); /* ctx may be NULL: last create/decoder error */
int kd_abi_version(void);

/* zero all tables and drop all insertion events (fresh parse_records state) */
int kd_reset(kd_ctx *ctx);
int kd_set_mode(kd_ctx *ctx, int mode);
/* window-path tuning (0 = keep default): sites per LDS window, reads per work item */
int kd_set_tuning(kd_ctx *ctx, uint32_t window_sites, uint32_t slice_reads);
/* the tuning in effect: out[0] = sites per LDS window, out[1] = reads per work item (0 = chosen per batch) */
int kd_get_tuning(const kd_ctx *ctx, uint32_t out[2]);

/* G-space geometry */
uint64_t kd_contig_base(const kd_ctx *ctx, uint32_t contig);
uint64_t kd_total_sites(const kd_ctx *ctx);

/* Multi-GPU: restrict this context to G-space interval [g_lo, g_hi).  Table increments
 * landing outside [g_lo, g_hi] (one halo site for aligned_depth_next, kindel.py:405-410)
 * are dropped; consensus is emitted for [g_lo, g_hi) only.  Default: everything. */
int kd_set_shard(kd_ctx *ctx, uint64_t g_lo, uint64_t g_hi);

/* ---- pileup: the record loop of parse_records, kindel.py:40-81 ---------------------- */

/* Host arrays; copied to the device before returning, buffers are reusable at once. */
int kd_push_batch(kd_ctx *ctx, const kd_batch *host_batch);
/* Same, but every pointer in *dev_batch is a DEVICE pointer on ctx's GPU (inputs already
 * resident in HBM); the arrays must stay valid until the next kd_sync/kd_finalize.  seq4 must be
 * 16-byte aligned with 16 readable bytes past seq4_bytes (the kernels stage bases with 16-byte loads).
 * The library launches on a stream of its own: whatever produces the arrays (another stream's kernels or
 * copies) must have COMPLETED when the call is made (hipStreamSynchronize / torch.cuda.synchronize) --
 * the kernels follow cig_off / seq_off as they find them. */
int kd_push_batch_device(kd_ctx *ctx, const kd_batch *dev_batch);
int kd_sync(kd_ctx *ctx);

/* After the last batch: reduce the insertion events into per-site
 * {total, unique-majority string | tie}  (insertions dicts + consensus(insertions[pos]),
 * kindel.py:55-58, :402, :420-421).  Raises the deferred reference exception, if any
 * (returns KD_E_BASE / KD_E_RANGE / KD_E_CIGAR; *err_read = index of the first failing
 * read counted over all pushed batches, may be NULL). */
int kd_finalize(kd_ctx *ctx, uint64_t *err_read);

/* counters since the last kd_reset: [0] reads counted (kindel.py:43-46 passed),
 * [1] aligned-base events (M/=/X lengths), [2] walked events (M,=,X,I,D,S lengths),
 * [3] insertion events */
int kd_get_stats(kd_ctx *ctx, uint64_t out[4]);

/* what the last kd_push_batch* did: [0] 1 if the windowed LDS path ran (0: global atomics),
 * [1] regular reads, [2] reads with soft-clip / insertion side effects (cold pass),
 * [3] irregular reads (exact-semantics wave kernel), [4] long-CIGAR reads, [5] work items,
 * [6] max reference span of a regular read, [7] reads out of G-start order */
int kd_get_batch_info(kd_ctx *ctx, uint64_t out[8]);

/* ---- tables: the `alignment` fields, kindel.py:97-128 -------------------------------- */

/* Copy `n_ch` channels of one contig to host: out is [n_ch][len+1] u32, row i = channel
 * channels[i] (KD_CH_*).  Slot len of the weight channels is always 0. */
int kd_get_tables(kd_ctx *ctx, uint32_t contig, uint32_t n_ch, const uint32_t *channels,
                  uint32_t *out);

/* insertions[site] dicts of one contig (kindel.py:38,55-58) as flat arrays: call with
 * all-NULL outputs to get *n_keys / *n_bytes, then again with buffers.  Keys of one site
 * are adjacent; string k is bytes[off[k] .. off[k]+len[k]) in upper-case ASCII. */
int kd_get_insertions(kd_ctx *ctx, uint32_t contig, uint64_t *n_keys, uint64_t *n_bytes,
                      uint32_t *site, uint32_t *count, uint32_t *len, uint64_t *off,
                      uint8_t *bytes);

/* ---- consensus: consensus_sequence, kindel.py:384-430 -------------------------------- */

/* Run the per-site pass over every contig (this context's shard).
 * patches: n_patches site ranges [patch_start, patch_end) in G-space that emit nothing and
 * record no change (the skip logic of kindel.py:393-401; the patch text itself is spliced
 * by the caller at patch_off).  Raw output: before trim_ends / uppercase (kindel.py:425-428). */
int kd_consensus_run(kd_ctx *ctx, uint32_t min_depth, uint32_t n_patches,
                     const uint64_t *patch_start, const uint64_t *patch_end);
/* Fetch one contig's result of the last kd_consensus_run: sequence bytes (upper-case site
 * bases, lower-case insertions), changes[len] in {0,'D','N','I'} (may be NULL),
 * depth_minmax[2] = min,max of A+C+G+T (build_report, kindel.py:450,477-479; may be NULL),
 * patch_off[n_patches] = offset in seq_out where each patch starting in this contig splices
 * (UINT64_MAX for patches of other contigs; may be NULL). */
int kd_consensus_fetch(kd_ctx *ctx, uint32_t contig, uint8_t *seq_out, uint64_t cap,
                       uint64_t *len_out, uint8_t *changes, uint32_t *depth_minmax,
                       uint64_t *patch_off);
/* All contigs in ONE device-to-host copy (multi-contig inputs: kindel.py:515-551 loops over the contigs):
 * seq_out[cap] receives the concatenated consensus bytes, contig c = seq_out[contig_off[c] .. contig_off[c+1])
 * with contig_off[n_contigs+1] filled in; changes (may be NULL) receives kd_total_sites() change codes in G-space
 * (contig c = changes[kd_contig_base(c) ..+len]).  *len_out = total bytes (always set; call with seq_out = NULL
 * to size the buffer). */
int kd_consensus_fetch_all(kd_ctx *ctx, uint8_t *seq_out, uint64_t cap, uint64_t *len_out, uint64_t *contig_off,
                           uint8_t *changes);
/* Device-side view of the whole shard's consensus (for the multi-GPU all-gather):
 * *dev_ptr = device pointer to the concatenated bytes, *n_bytes its length. */
int kd_consensus_device(kd_ctx *ctx, void **dev_ptr, uint64_t *n_bytes);
/* Same for the per-site change codes: *dev_ptr + g = the change code of G-space site g, valid for the sites of the context's
 * interval [g_lo, g_hi) only (the array is shard-local like the tables; the pointer is biased to G-space indexing). */
int kd_changes_device(kd_ctx *ctx, void **dev_ptr);
/* The EXCHANGE ROW of this context's shard -- what the multi-GPU all-gather moves (kindel_amd/shard.py; the stitch of the
 * per-contig loop kindel.py:515-551 across ranks) -- written by the context into a caller's DEVICE buffer of cap bytes:
 *   u64 row_bytes | u64 0 | contig_off u64[n_contigs + 1] | depth min / max u32[2 n_contigs] | change codes of the shard's
 *   sites [g_lo, g_hi) | consensus bytes (kd_consensus_device's)
 * row_bytes > cap: the row did not fit, only its 16-byte header is there (every rank reads that in the gathered rows and the
 * gather is repeated with the announced size).  kd_exchange_row: on demand after kd_consensus_run / kd_finish / kd_step;
 * complete when it returns.  kd_set_exchange: registers a row that kd_finish and kd_step then fill on their way -- the two
 * device-to-device copies queued behind the consensus kernels, the header with the run's collected metadata -- so that a
 * multi-GPU step is kd_step + ONE collective; complete when kd_finish / kd_step return.  dev_row = NULL unregisters.  cap >= 16. */
int kd_exchange_row(kd_ctx *ctx, void *dev_row, uint64_t cap, uint64_t *row_bytes);
int kd_set_exchange(kd_ctx *ctx, void *dev_row, uint64_t cap);
/* One whole step over a DEVICE-resident batch in one call: kd_reset + kd_push_batch_device + kd_finalize + kd_consensus_run (no
 * patches) + kd_consensus_fetch_all(seq_out ...), i.e. parse_records' loop and consensus_sequence's loop (kindel.py:40-81,
 * :384-430) for every contig of the batch, queued back to back with two host round trips.  seq_out should be pinned host memory
 * (hipHostMalloc / hipHostRegister: the consensus kernel then writes the bytes there itself).  Errors as kd_finalize (the batch's error classification runs when the step's status words ask for it).
 * (ABI 1 had an opt-in hipGraph replay of a repeated step -- kd_set_step_graph, an int *replayed here -- removed in
 * ABI 2: it measured nothing over the eager sequence and was the one path that faulted on hardware) */
int kd_step(kd_ctx *ctx, const kd_batch *dev_batch, uint32_t min_depth, uint8_t *seq_out, uint64_t cap, uint64_t *len_out,
            uint64_t *contig_off);
/* kd_finish: everything behind the pushes in one call and ONE host round trip -- kd_finalize + kd_consensus_run(min_depth, no
 * patches) + kd_consensus_fetch_all -- i.e. consensus(insertions[pos]) :420 and consensus_sequence :384-430 for all contigs, the
 * bytes of all contigs in G-space order into seq_out (cap bytes; pinned memory makes the copy asynchronous), *len_out their number,
 * contig_off[n_contigs + 1] (may be NULL) each contig's offset.  Errors as kd_finalize (the deferred reference exceptions).  Afterwards
 * the context is finalized and has a consensus run: kd_get_tables / kd_get_insertions / kd_consensus_fetch / kd_changes_device work. */
int kd_finish(kd_ctx *ctx, uint32_t min_depth, uint8_t *seq_out, uint64_t cap, uint64_t *len_out, uint64_t *contig_off);
/* Host-side metadata of the last run: contig_off[n_contigs+1] = byte offset of each contig in the
 * concatenated consensus (last entry = total), depth_minmax[2*n_contigs]. Either may be NULL. */
int kd_consensus_offsets(kd_ctx *ctx, uint64_t *contig_off, uint32_t *depth_minmax);

/* ---- profiling ---------------------------------------------------------------------- */

/* on = 1: every kernel launch is bracketed by hipEvents on ctx's stream; on = 2: only the launches of the
 * pileup kernel k_window (what a timed run wants: two events per batch instead of two per launch); 0: off. */
int kd_profile_enable(kd_ctx *ctx, int on);
/* Accumulated since enable/reset: number of kernel rows (call with names==NULL), then
 * per row: name (<=63 chars + NUL in names[i*64]), launches, total milliseconds. */
int kd_profile_get(kd_ctx *ctx, uint32_t *n_rows, char *names, uint64_t *launches, double *ms);
int kd_profile_reset(kd_ctx *ctx);

/* ---- host decoder: what simplesam.Reader does for parse_bam, kindel.py:136-145 -------- */

typedef struct kd_file kd_file;
/* Decode a SAM text or BAM (BGZF) file into one kd_batch held by the handle.  No GPU use. */
int kd_decode_open(kd_file **out, const char *path, int n_threads);
/* Multi-GPU ingest (every rank opens the file and decodes only its share; kindel.py:143-151 distributed over the ranks):
 * kd_bgzf_index -- number of BGZF blocks and, if in_off != NULL, the compressed file offset of each (to cut the file into byte
 * shares); KD_E_IO when the file is not BGZF.
 * kd_decode_open_span -- the records that begin in blocks [block_lo, block_hi): from the first offset at or behind the start
 * of block_lo where 16 consecutive well-formed records begin (block 0: the first record) up to the offset found the same way
 * for block_hi (or the end of the file).  The boundaries are a function of the file alone -- all ranks agree on them -- and
 * the record chain that starts on one must end exactly on the next, else KD_E_IO (the caller then reads the whole file).
 * info[4]: absolute uncompressed offsets of the span's start / end, records walked, 1 if the span reaches the end of file. */
int kd_bgzf_index(const char *path, uint64_t *n_blocks, uint64_t *in_off, uint64_t cap);
int kd_decode_open_span(kd_file **out, const char *path, int n_threads, uint64_t block_lo, uint64_t block_hi, uint64_t *info);
const kd_batch *kd_decode_batch(const kd_file *f);
uint32_t kd_decode_n_contigs(const kd_file *f);
const char *kd_decode_contig_name(const kd_file *f, uint32_t i);
uint32_t kd_decode_contig_len(const kd_file *f, uint32_t i);
uint64_t kd_decode_n_records(const kd_file *f); /* all records in the file, incl. dropped RNAME '*' */
void kd_decode_close(kd_file *f);
const char *kd_decode_last_error(void);
/* host threads the decoder starts by default (n_threads = 0): the visible cores, capped at 1.5 x the cgroup CPU quota */
uint32_t kd_host_threads(void);
/* The BGZF reader's block decoder on its own (kd_inflate.h): the raw DEFLATE stream in[0, in_len) must decode to exactly
   out_len bytes; KD_OK or KD_E_IO (malformed / truncated stream, other size).  Never writes outside out[0, out_len).
   Replaces what pysam / htslib's bgzf_read do under kindel.py:131-134. */
int kd_host_inflate(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_len);
/* The CRC-32 the BGZF reader checks every inflated block with (kd_crc32.h; zlib's crc32 convention): htslib refuses a block whose
   trailer does not match, and so does this reader. */
uint32_t kd_host_crc32(const uint8_t *data, uint64_t len);

/* ---- device-side ingest (opt-in) ----------------------------------------
 * parse_bam's record iteration (kindel.py:131-153) with the FILE's bytes as the only thing the host touches: the BGZF blocks are
 * inflated on the GPU (one wavefront per block), the BAM record chain is walked there from speculative, verified starts, the
 * kd_batch arrays are written in HBM and pushed like kd_push_batch_device would.  kd_bgzf_plan_*: the host's share -- the file
 * mapped, the BGZF block table, the BAM header (the first blocks inflated on the host) -- also what a caller needs to create
 * the context (contig table).  kd_push_bam_gpu returns KD_E_UNSUPPORTED when the file needs the host decoder (SAM text / plain
 * gzip never get here: kd_bgzf_plan_open refuses them with KD_E_UNSUPPORTED too); KD_E_IO for a corrupt file.
 * stats (may be NULL): [0] records seen, [1] records kept (mapped), [2] inflated bytes, [3] BGZF blocks,
 * [4] us host plan, [5] us H2D + kernels until the batch exists, [6] us push. */
typedef struct kd_bgzf_plan kd_bgzf_plan;
int kd_bgzf_plan_open(kd_bgzf_plan **out, const char *path);
uint32_t kd_bgzf_plan_n_contigs(const kd_bgzf_plan *p);
const char *kd_bgzf_plan_contig_name(const kd_bgzf_plan *p, uint32_t i);
uint32_t kd_bgzf_plan_contig_len(const kd_bgzf_plan *p, uint32_t i);
/* the plan's arrays: file bytes, n_blocks x {payload offset, inflated offset, payload bytes, inflated bytes} (blocks that inflate
   to nothing left out), length of the inflated stream, offset of the first record in it */
int kd_bgzf_plan_view(const kd_bgzf_plan *p, const uint8_t **file, uint64_t *file_bytes, const void **blocks, uint32_t *n_blocks,
                      uint64_t *total_out, uint64_t *hdr_end);
void kd_bgzf_plan_close(kd_bgzf_plan *p);
int kd_push_bam_gpu(kd_ctx *ctx, const kd_bgzf_plan *plan, uint64_t stats[8]);

/* ---- streaming ingest: the record iteration of parse_bam (kindel.py:143-145) without holding the whole file -------- */

typedef struct kd_stream kd_stream;
/* Open a SAM / BAM file for chunked reading: batches of about chunk_bytes uncompressed bytes (0 = 64 MiB), whole records
 * each, in file order.  The header (@SQ table) is available right after the call. */
int kd_stream_open(kd_stream **out, const char *path, int n_threads, uint64_t chunk_bytes);
uint32_t kd_stream_n_contigs(const kd_stream *s);
const char *kd_stream_contig_name(const kd_stream *s, uint32_t i);
uint32_t kd_stream_contig_len(const kd_stream *s, uint32_t i);
/* Next batch, or *batch = NULL at the end of the file.  Two batches are kept: a batch stays valid until the call AFTER the
 * next one, so one thread can decode batch k+1 while another still reads batch k. */
int kd_stream_next(kd_stream *s, const kd_batch **batch);
/* Records come out with contig = map[refID] from the next kd_stream_next on (map: one entry per @SQ line; 0xffffffff = "no record
 * may lie here": KD_E_ARG if one does; 0xfffffffe = "records here are dropped": one of several passes over a file whose reference
 * does not fit the device at once, each with a group of contigs laid out; n = 0 restores the identity).  For a header far larger than what the records touch (a
 * human-genome @SQ table, reads on one small contig): the reference lays out only the RNAMEs it sees (kindel.py:143-151) -- the
 * caller scans the file once for the contigs in use, creates its context over those, and streams with this map. */
int kd_stream_set_contig_map(kd_stream *s, const uint32_t *map, uint32_t n);
uint64_t kd_stream_n_records(const kd_stream *s); /* records seen so far, incl. dropped RNAME '*' */
const char *kd_stream_last_error(const kd_stream *s);
void kd_stream_close(kd_stream *s);

/* Push every remaining batch of `s` into ctx (kd_create'd over the stream's contig table): a producer thread decodes batch
 * k+1 while this thread copies batch k to the device and launches its kernels.  stats (may be NULL): [0] batches,
 * [1] microseconds the decoder was busy, [2] microseconds the pushes took, [3] wall microseconds. */
int kd_push_stream(kd_ctx *ctx, kd_stream *s, uint64_t stats[4]);
/* kd_stream_open + kd_push_stream + kd_stream_close for a file whose @SQ table equals ctx's contig table */
int kd_decode_push_file(kd_ctx *ctx, const char *path, int n_threads, uint64_t chunk_bytes, uint64_t stats[4]);
/* first_idx[n_contigs]: index (over all pushed records) of each contig's first record, UINT64_MAX = no record: the
 * contigs parse_bam returns and their order (first appearance, kindel.py:143-151) */
int kd_get_contig_first(kd_ctx *ctx, uint64_t *first_idx);

#ifdef __cplusplus
}
#endif
#endif /* KINDEL_HIP_H */
