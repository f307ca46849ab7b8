// kd_abi.inl -- extern "C" entry points of include/kindel_hip.h over KdEngine<KD_RT>.
// Included once by the translation unit that defines KD_RT (kindel_hip.hip for the product,
// tests/emu/emu_lib.cpp for the kernel-logic emulator).  Decoder entry points live in
// kd_decode.cpp.
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>

struct kd_ctx {
    KdEngine<KD_RT> e;
};

static std::string g_kd_create_error;

extern "C" {

int kd_abi_version(void) { return KD_ABI_VERSION; }

int kd_create(kd_ctx **out, int device, uint32_t n_contigs, const uint32_t *contig_len, void *stream) {
    if (!out) return KD_E_ARG;
    *out = nullptr;
    kd_ctx *c = new (std::nothrow) kd_ctx();
    if (!c) return KD_E_NOMEM;
    int rc = c->e.create(device, n_contigs, contig_len, stream);
    if (rc) {
        g_kd_create_error = c->e.err;
        c->e.destroy();
        delete c;
        return rc;
    }
    *out = c;
    return KD_OK;
}

void kd_destroy(kd_ctx *ctx) {
    if (!ctx) return;
    ctx->e.destroy();
    delete ctx;
}

const char *kd_last_error(const kd_ctx *ctx) { return ctx ? ctx->e.err.c_str() : g_kd_create_error.c_str(); }

int kd_reset(kd_ctx *ctx) { return ctx ? ctx->e.reset() : KD_E_ARG; }

int kd_set_mode(kd_ctx *ctx, int mode) {
    if (!ctx || mode < KD_MODE_AUTO || mode > KD_MODE_COOP) return KD_E_ARG;
    ctx->e.mode = mode;
    return KD_OK;
}

int kd_set_tuning(kd_ctx *ctx, uint32_t window_sites, uint32_t slice_reads) {
    if (!ctx) return KD_E_ARG;
    if (window_sites) {
        if (window_sites < 64 || window_sites > 4096 || (window_sites & 63)) return ctx->e.fail(KD_E_ARG, "kd_set_tuning: window must be a multiple of 64 in [64, 4096]");
        ctx->e.W = window_sites;
    }
    ctx->e.slice_cfg = slice_reads;
    return KD_OK;
}

int kd_get_tuning(const kd_ctx *ctx, uint32_t out[2]) {
    if (!ctx || !out) return KD_E_ARG;
    out[0] = ctx->e.window_sites(ctx->e.coop_mode()); out[1] = ctx->e.slice_cfg;
    return KD_OK;
}

uint64_t kd_contig_base(const kd_ctx *ctx, uint32_t contig) {
    return (ctx && contig < ctx->e.n_contigs) ? ctx->e.cbase[contig] : ~0ULL;
}
uint64_t kd_total_sites(const kd_ctx *ctx) { return ctx ? ctx->e.S : 0; }

int kd_set_shard(kd_ctx *ctx, uint64_t g_lo, uint64_t g_hi) { return ctx ? ctx->e.set_shard(g_lo, g_hi) : KD_E_ARG; }

int kd_push_batch(kd_ctx *ctx, const kd_batch *b) { return (ctx && b) ? ctx->e.push_host(*b) : KD_E_ARG; }
int kd_push_batch_device(kd_ctx *ctx, const kd_batch *b) { return (ctx && b) ? ctx->e.push_device(*b) : KD_E_ARG; }
int kd_sync(kd_ctx *ctx) {
    if (!ctx) return KD_E_ARG;
    return ctx->e.rt.sync() ? ctx->e.hipfail("kd_sync") : KD_OK;
}
int kd_finalize(kd_ctx *ctx, uint64_t *err_read) { return ctx ? ctx->e.finalize(err_read) : KD_E_ARG; }
int kd_get_stats(kd_ctx *ctx, uint64_t out[4]) { return (ctx && out) ? ctx->e.get_stats(out) : KD_E_ARG; }

int kd_get_batch_info(kd_ctx *ctx, uint64_t out[8]) {
    if (!ctx || !out) return KD_E_ARG;
    int rc = ctx->e.fetch_status();
    if (rc) return rc;
    const auto &h = ctx->e.h_status;
    out[0] = ctx->e.last_windowed; out[1] = h[KDS_B_N_REG]; out[2] = h[KDS_B_N_COLD]; out[3] = h[KDS_B_N_IRREG];
    out[4] = h[KDS_B_N_LONG]; out[5] = ctx->e.last_nwin + h[KDS_TOTAL_ITEMS] /* work items: windows planned + the extra slices of deep windows */; out[6] = h[KDS_B_MAXSPAN]; out[7] = h[KDS_B_UNSORTED];
    return KD_OK;
}

int kd_get_tables(kd_ctx *ctx, uint32_t contig, uint32_t n_ch, const uint32_t *channels, uint32_t *out) {
    return (ctx && channels && out) ? ctx->e.get_tables(contig, n_ch, channels, out) : KD_E_ARG;
}
int kd_get_insertions(kd_ctx *ctx, uint32_t contig, uint64_t *n_keys, uint64_t *n_bytes, uint32_t *site,
                      uint32_t *count, uint32_t *len, uint64_t *off, uint8_t *bytes) {
    return ctx ? ctx->e.get_insertions(contig, n_keys, n_bytes, site, count, len, off, bytes) : KD_E_ARG;
}

int kd_consensus_run(kd_ctx *ctx, uint32_t min_depth, uint32_t n_patches, const uint64_t *patch_start,
                     const uint64_t *patch_end) {
    if (!ctx || (n_patches && (!patch_start || !patch_end))) return KD_E_ARG;
    return ctx->e.consensus_run(min_depth, n_patches, patch_start, patch_end);
}
int kd_consensus_fetch(kd_ctx *ctx, uint32_t contig, uint8_t *seq_out, uint64_t cap, uint64_t *len_out,
                       uint8_t *changes, uint32_t *depth_minmax, uint64_t *patch_off) {
    return ctx ? ctx->e.consensus_fetch(contig, seq_out, cap, len_out, changes, depth_minmax, patch_off) : KD_E_ARG;
}
int kd_consensus_fetch_all(kd_ctx *ctx, uint8_t *seq_out, uint64_t cap, uint64_t *len_out, uint64_t *contig_off,
                           uint8_t *changes) {
    return ctx ? ctx->e.consensus_fetch_all(seq_out, cap, len_out, contig_off, changes) : KD_E_ARG;
}
int kd_consensus_device(kd_ctx *ctx, void **dev_ptr, uint64_t *n_bytes) {
    if (!ctx || !dev_ptr || !n_bytes) return KD_E_ARG;
    if (!ctx->e.have_cns) return ctx->e.fail(KD_E_ARG, "kd_consensus_device: call kd_consensus_run first");
    *dev_ptr = ctx->e.b_cns.p;
    *n_bytes = ctx->e.h_coff[ctx->e.n_contigs];
    return KD_OK;
}

int kd_finish(kd_ctx *ctx, uint32_t min_depth, uint8_t *seq_out, uint64_t cap, uint64_t *len_out, uint64_t *contig_off) {
    return ctx ? ctx->e.finish(min_depth, seq_out, cap, len_out, contig_off) : KD_E_ARG;
}
int kd_step(kd_ctx *ctx, const kd_batch *dev_batch, uint32_t min_depth, uint8_t *seq_out, uint64_t cap, uint64_t *len_out,
            uint64_t *contig_off) {
    if (!ctx || !dev_batch) return KD_E_ARG;
    return ctx->e.step(*dev_batch, min_depth, seq_out, cap, len_out, contig_off);
}

int kd_changes_device(kd_ctx *ctx, void **dev_ptr) {
    if (!ctx || !dev_ptr) return KD_E_ARG;
    if (!ctx->e.have_cns) return ctx->e.fail(KD_E_ARG, "kd_changes_device: call kd_consensus_run first");
    // (the array is shard-local; the pointer is biased so that  pointer + g  is the change code of G-space site g of the shard)
    *dev_ptr = (uint8_t *)ctx->e.b_changes.p - ctx->e.alloc_lo;
    return KD_OK;
}
int kd_set_exchange(kd_ctx *ctx, void *dev_row, uint64_t cap) {
    if (!ctx || (dev_row && cap < 16)) return KD_E_ARG;
    ctx->e.exch_row = (uint8_t *)dev_row; ctx->e.exch_cap = dev_row ? cap : 0;
    return KD_OK;
}
int kd_exchange_row(kd_ctx *ctx, void *dev_row, uint64_t cap, uint64_t *row_bytes) {
    if (!ctx) return KD_E_ARG;
    return ctx->e.exchange_row((uint8_t *)dev_row, cap, row_bytes);
}
int kd_consensus_offsets(kd_ctx *ctx, uint64_t *contig_off, uint32_t *depth_minmax) {
    if (!ctx) return KD_E_ARG;
    if (!ctx->e.have_cns) return ctx->e.fail(KD_E_ARG, "kd_consensus_offsets: call kd_consensus_run first");
    if (contig_off) memcpy(contig_off, ctx->e.h_coff.data(), ctx->e.h_coff.size() * 8);
    if (depth_minmax) memcpy(depth_minmax, ctx->e.h_minmax.data(), ctx->e.h_minmax.size() * 4);
    return KD_OK;
}

int kd_get_contig_first(kd_ctx *ctx, uint64_t *first_idx) {
    if (!ctx || !first_idx) return KD_E_ARG;
    if (ctx->e.rt.d2h(first_idx, ctx->e.d_first_idx, (size_t)ctx->e.n_contigs * 8)) return ctx->e.hipfail("kd_get_contig_first");
    return KD_OK;
}

int kd_push_stream(kd_ctx *ctx, kd_stream *s, uint64_t stats[4]) {
    if (!ctx || !s) return KD_E_ARG;
    typedef std::chrono::steady_clock clk;
    auto us = [](clk::time_point a, clk::time_point b) { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
    const clk::time_point t_begin = clk::now();
    // one-slot hand-off: the producer decodes batch k+1 (into the stream's other slot) while the consumer pushes batch k
    std::mutex mu;
    std::condition_variable cv;
    const kd_batch *ready = nullptr;
    bool have = false, finished = false;
    int prod_rc = KD_OK;
    uint64_t decode_us = 0;
    std::thread producer([&]() {
        for (;;) {
            const kd_batch *b = nullptr;
            const clk::time_point t0 = clk::now();
            const int rc = kd_stream_next(s, &b);
            decode_us += us(t0, clk::now());
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !have; });     // the consumer has taken the previous batch (and finished reading the one before)
            if (rc) { prod_rc = rc; finished = true; cv.notify_all(); return; }
            if (!b) { finished = true; cv.notify_all(); return; }
            ready = b; have = true;
            cv.notify_all();
        }
    });
    uint64_t n_batches = 0, push_us = 0;
    int rc = KD_OK;
    for (;;) {
        const kd_batch *b = nullptr;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return have || finished; });
            if (!have) break;
            b = ready;
        }
        const clk::time_point t0 = clk::now();
        if (rc == KD_OK) rc = ctx->e.push_host(*b);   // after an error the remaining batches are only drained
        push_us += us(t0, clk::now());
        n_batches++;
        {
            std::lock_guard<std::mutex> lk(mu);
            have = false;                             // the slot may be decoded into again
        }
        cv.notify_all();
    }
    producer.join();
    if (rc == KD_OK && prod_rc) rc = ctx->e.fail(prod_rc, kd_stream_last_error(s));
    if (stats) { stats[0] = n_batches; stats[1] = decode_us; stats[2] = push_us; stats[3] = us(t_begin, clk::now()); }
    return rc;
}

int kd_push_bam_gpu(kd_ctx *ctx, const kd_bgzf_plan *plan, uint64_t stats[8]) {
    if (!ctx || !plan) return KD_E_ARG;
    const uint8_t *file = nullptr; const void *blocks = nullptr;
    uint64_t file_bytes = 0, total = 0, hdr_end = 0;
    uint32_t n_blocks = 0;
    int rc = kd_bgzf_plan_view(plan, &file, &file_bytes, &blocks, &n_blocks, &total, &hdr_end);
    if (rc) return ctx->e.fail(rc, "kd_push_bam_gpu: more than 2^32 BGZF blocks");
    bool same = kd_bgzf_plan_n_contigs(plan) == ctx->e.n_contigs;
    for (uint32_t c = 0; same && c < ctx->e.n_contigs; c++) same = kd_bgzf_plan_contig_len(plan, c) == ctx->e.clen[c];
    if (!same) return ctx->e.fail(KD_E_ARG, "kd_push_bam_gpu: the file's @SQ table differs from the context's contig table");
    return ctx->e.ingest_bam(file, file_bytes, blocks, n_blocks, total, hdr_end, stats);
}

int kd_decode_push_file(kd_ctx *ctx, const char *path, int n_threads, uint64_t chunk_bytes, uint64_t stats[4]) {
    if (!ctx || !path) return KD_E_ARG;
    kd_stream *s = nullptr;
    int rc = kd_stream_open(&s, path, n_threads, chunk_bytes);
    if (rc) return ctx->e.fail(rc, kd_stream_last_error(nullptr));
    bool same = kd_stream_n_contigs(s) == ctx->e.n_contigs;
    for (uint32_t c = 0; same && c < ctx->e.n_contigs; c++) same = kd_stream_contig_len(s, c) == ctx->e.clen[c];
    if (!same) { kd_stream_close(s); return ctx->e.fail(KD_E_ARG, "kd_decode_push_file: the file's @SQ table differs from the context's contig table"); }
    rc = kd_push_stream(ctx, s, stats);
    kd_stream_close(s);
    return rc;
}

int kd_profile_enable(kd_ctx *ctx, int on) {
    if (!ctx) return KD_E_ARG;
    ctx->e.rt.profile_enable(on == 2 ? 2 : on != 0);
    return KD_OK;
}
int kd_profile_get(kd_ctx *ctx, uint32_t *n_rows, char *names, uint64_t *launches, double *ms) {
    if (!ctx || !n_rows) return KD_E_ARG;
    return ctx->e.rt.profile_get(n_rows, names, launches, ms) ? ctx->e.hipfail("kd_profile_get") : KD_OK;
}
int kd_profile_reset(kd_ctx *ctx) {
    if (!ctx) return KD_E_ARG;
    ctx->e.rt.profile_reset();
    return KD_OK;
}

}  // extern "C"
