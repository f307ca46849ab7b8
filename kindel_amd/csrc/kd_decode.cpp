// kd_decode.cpp -- host decoder: SAM text / BAM (BGZF) -> kd_batch (include/kindel_hip.h).
//
// Replaces what simplesam.Reader + `samtools view` do for parse_bam
// (/root/reference/kindel/kindel.py:136-148): header @SQ -> {name: LN}, record iteration,
// dropping RNAME '*'.  No arithmetic of the hot path lives here and no GPU is touched.
// Formats: SAMv1 spec sections 1.3-1.4 (text), 4.1 (BGZF), 4.2 (BAM records).
// Everything is parallel over the host cores: BGZF blocks are independent deflate streams and are inflated into their
// final positions (ISIZE prefix sum); the BAM record chain is walked in ranges from speculative, verified starts; SAM
// text is parsed in line-aligned ranges; the SoA arrays are filled / merged range by range.
// (KD_DECODE_RANGE_BYTES shrinks the ranges so that the tests can exercise the range logic on small files.)
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include "kd_inflate.h"
#include "kd_crc32.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/kindel_hip.h"

namespace {

// Threads worth starting: the visible cores, capped by the cgroup CPU quota (a container that sees 256 cores but may use 16
// of them per scheduling period is throttled to a standstill by 256 busy threads: measured on the GPU boxes).
unsigned hw_threads() {
    static const unsigned n = [] {
        unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        long long quota = -1, period = 100000;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {             // cgroup v2: "<quota|max> <period>"
            char q[64] = {0};
            if (fscanf(f, "%63s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
            if (fscanf(g, "%lld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
        }
        // 1.5 x the quota: the workers also wait (hand-offs, page faults), and on the MI355X boxes (quota 16 of 256 cores)
        // 24 threads decoded a 4.4 GB BAM in 0.33 s, 16 in 1.08 s, 48 in 0.62 s, 256 in 1.6 s (profiles/r02_e2e_*.json)
        if (quota > 0 && period > 0) hw = (unsigned)std::min<long long>(hw, std::max<long long>(1, (3 * quota + 2 * period - 1) / (2 * period)));
        return hw;
    }();
    return n;
}

// Worker threads kept across calls: a chunk of a streamed file is a few tens of milliseconds of work split three times
// (inflate, record walk, fill), and starting 255 threads for each of those cost more than the work itself.
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu, run_mu;
    std::condition_variable cv, done_cv;
    const std::function<void(unsigned)> *fn = nullptr;
    unsigned n_tasks = 0, remaining = 0, active = 0;   // active: workers inside drain() for the current generation
    unsigned limit = 0;                                 // workers 0 .. limit-1 take part in the current generation
    std::atomic<unsigned> next{0};
    uint64_t gen = 0;
    bool stop = false;
    ~Pool() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
    void drain(const std::function<void(unsigned)> &f, unsigned n) {
        unsigned done = 0;
        for (;;) {
            const unsigned t = next.fetch_add(1);
            if (t >= n) break;
            f(t);
            done++;
        }
        if (done) {
            std::lock_guard<std::mutex> g(mu);
            remaining -= done;
            if (!remaining) done_cv.notify_all();
        }
    }
    void worker(unsigned idx) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(unsigned)> *f;
            unsigned n;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen; f = fn; n = n_tasks;
                if (!f || idx >= limit) continue;   // woke up after its generation was over / not wanted for this one
                active++;
            }
            drain(*f, n);
            {
                std::lock_guard<std::mutex> g(mu);
                if (--active == 0) done_cv.notify_all();
            }
        }
    }
    // f(0) .. f(n-1), each exactly once, on up to `threads` threads (the caller included); returns when all are done
    void run(unsigned n, unsigned threads, const std::function<void(unsigned)> &f) {
        if (n == 0) return;
        if (n == 1 || threads <= 1) { for (unsigned t = 0; t < n; t++) f(t); return; }
        std::lock_guard<std::mutex> g(run_mu);
        const unsigned want = std::min(n, threads) - 1;
        while (th.size() < want) { const unsigned idx = (unsigned)th.size(); th.emplace_back([this, idx] { worker(idx); }); }
        {
            std::lock_guard<std::mutex> lk(mu);
            fn = &f; n_tasks = n; remaining = n; next = 0; limit = want; gen++;
        }
        cv.notify_all();
        drain(f, n);
        std::unique_lock<std::mutex> lk(mu);
        // no worker may still be inside drain() when `f` goes out of scope or `next` is reset for the next generation
        done_cv.wait(lk, [&] { return remaining == 0 && active == 0; });
        fn = nullptr;
    }
};
Pool &pool() { static Pool P; return P; }

thread_local std::string g_decode_error;   // per thread: concurrent kd_decode_open calls do not share it

// Growable array WITHOUT value-initialisation: resize() of a fresh array does not touch the pages, so the
// worker threads that fill it take the page faults in parallel (a zero-filling std::vector::resize of a few
// hundred MB on one thread was most of the decode time).
template <class T>
struct Arr {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    Arr() = default;
    Arr(const Arr &) = delete;
    Arr &operator=(const Arr &) = delete;
    ~Arr() { free(p); }
    bool resize(size_t m) {
        if (m > cap) {
            size_t c = std::max(m, cap + cap / 2 + 16);
            T *q = (T *)realloc(p, c * sizeof(T));
            if (!q) return false;
            p = q; cap = c;
        }
        n = m;
        return true;
    }
    void push_back(const T &v) { resize(n + 1); p[n - 1] = v; }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

struct File {
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    Arr<uint32_t> contig, flag, seq_len, n_cig, cigar;
    Arr<int32_t> pos0;
    Arr<uint64_t> seq_off, cig_off;
    Arr<uint8_t> seq4;
    uint64_t n_records = 0;
    bool append = false;       // parse_bam_records adds its records BEHIND what the arrays hold (kd_decode_open: chunk after chunk)
    kd_batch view;
};

inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | p[1] << 8); }

// Everything the descriptor still holds, up to its end: also for an input without a size (a pipe, /dev/stdin), which can be
// opened and read only ONCE.
bool read_all(int fd, Arr<uint8_t> &out) {
    size_t have = 0;
    if (!out.resize((size_t)1 << 20)) return false;
    for (;;) {
        if (have == out.size() && !out.resize(out.size() * 2)) return false;
        const ssize_t got = ::read(fd, out.data() + have, out.size() - have);
        if (got < 0) { if (errno == EINTR) continue; return false; }
        if (got == 0) break;
        have += (size_t)got;
    }
    return out.resize(have);
}

struct Block { size_t in_off, in_len, out_off, out_len; };

// The gzip member at p (avail >= 18 bytes left in the file, magic / FEXTRA already checked): the length of its extra field
// and its total size from the BC subfield (SAMv1 4.1).  Everything the header promises must lie inside the file BEFORE it is
// read: the file is memory-mapped, an xlen that points past its end is a SIGSEGV, not a short read.
inline bool bgzf_block_size(const uint8_t *p, size_t avail, size_t *xlen_out, size_t *bsize_out) {
    const size_t xlen = rd16(p + 10);
    if (12 + xlen + 8 > avail) return false;            // extra field + the 8-byte trailer do not fit
    size_t x = 12, bsize = 0;
    while (x + 4 <= 12 + xlen) {
        const size_t slen = rd16(p + x + 2);
        if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2) {
            if (x + 6 > 12 + xlen) return false;         // the BC payload would lie behind the extra field
            bsize = (size_t)rd16(p + x + 4) + 1;
        }
        x += 4 + slen;
    }
    if (!bsize || bsize > avail || bsize < 12 + xlen + 8) return false;
    *xlen_out = xlen; *bsize_out = bsize;
    return true;
}

// Split a BGZF file into its blocks using the BC extra subfield (SAMv1 4.1).
bool scan_bgzf(const Arr<uint8_t> &raw, std::vector<Block> &blocks, size_t &total) {
    size_t o = 0;
    total = 0;
    while (o + 18 <= raw.size()) {
        const uint8_t *p = raw.data() + o;
        if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return false;
        size_t xlen = 0, bsize = 0;
        if (!bgzf_block_size(p, raw.size() - o, &xlen, &bsize)) return false;
        const size_t isize = rd32(p + bsize - 4);
        blocks.push_back({o + 12 + xlen, bsize - (12 + xlen) - 8, total, isize});
        total += isize;
        o += bsize;
    }
    return o == raw.size();
}

// The same, a stretch at a time (kd_stream: the file is mapped, not read, and scanned as the chunks are taken: touching one
// header per 64 KiB block of a 300 MB file up front is tens of thousands of page faults before the first record is decoded).
// Appends blocks from offset *o until at least `want_out` more uncompressed bytes are covered or the file ends; false = malformed.
bool scan_bgzf_some(const uint8_t *raw, size_t n, size_t *o, std::vector<Block> &blocks, size_t *total, size_t want_out) {
    const size_t stop_at = *total + want_out;
    while (*o + 18 <= n && *total < stop_at) {
        const uint8_t *p = raw + *o;
        if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return false;
        size_t xlen = 0, bsize = 0;
        if (!bgzf_block_size(p, n - *o, &xlen, &bsize)) return false;
        const size_t isize = rd32(p + bsize - 4);
        blocks.push_back({*o + 12 + xlen, bsize - (12 + xlen) - 8, *total, isize});
        *total += isize;
        *o += bsize;
    }
    return *o + 18 <= n || *o == n;     // stopped early, or consumed the file exactly
}

// one BGZF block: kd_inflate.h (whole-buffer raw DEFLATE decoder; zlib's streaming inflate was 79 % of the decoder's CPU time),
// then the block's CRC-32 against its trailer -- in[in_len .. in_len + 4), in front of ISIZE; scan_bgzf* made sure it lies inside
// the file -- as htslib's bgzf_read does for the reference (kd_crc32.h: carry-less-multiply folding, a few percent of the
// inflater's time).  KD_BGZF_NO_CRC=1 skips the check (measurement).
static bool bgzf_check_crc() { static const bool on = !getenv("KD_BGZF_NO_CRC"); return on; }
bool inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
    if (!out_len) return true;
    if (!kdz::inflate_raw(in, in_len, out, out_len)) return false;
    return !bgzf_check_crc() || kdz::crc32_buf(out, out_len) == rd32(in + in_len);
}

// generic (non-BGZF) gzip: single stream, possibly several members
bool inflate_generic(const Arr<uint8_t> &raw, Arr<uint8_t> &out) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 32) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t *>(raw.data());
    zs.avail_in = (uInt)raw.size();
    out.resize(raw.size() * 4 + 65536);
    size_t have = 0;
    for (;;) {
        if (have == out.size()) out.resize(out.size() * 2);
        zs.next_out = out.data() + have;
        zs.avail_out = (uInt)std::min<size_t>(out.size() - have, 1u << 30);
        const int rc = inflate(&zs, Z_NO_FLUSH);
        have = zs.total_out;
        if (rc == Z_STREAM_END) {
            if (zs.avail_in == 0) break;
            if (inflateReset(&zs) != Z_OK) { inflateEnd(&zs); return false; }
        } else if (rc != Z_OK) { inflateEnd(&zs); return false; }
    }
    inflateEnd(&zs);
    out.resize(have);
    return true;
}

bool decompress(const Arr<uint8_t> &raw, Arr<uint8_t> &out, int n_threads) {
    std::vector<Block> blocks;
    size_t total = 0;
    if (!scan_bgzf(raw, blocks, total)) return inflate_generic(raw, out);
    out.resize(total);
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    auto work = [&]() {
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= blocks.size()) break;
            const Block &B = blocks[b];
            if (!inflate_raw(raw.data() + B.in_off, B.in_len, out.data() + B.out_off, B.out_len)) ok = false;
        }
    };
    unsigned nt = n_threads > 0 ? (unsigned)n_threads : hw_threads();
    nt = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, blocks.size()));
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    return ok;
}

void finish_view(File &f) {
    // slack so vector loads at the tail stay in bounds
    for (int k = 0; k < 8; k++) f.seq4.push_back(0);
    for (int k = 0; k < 2; k++) f.cigar.push_back(0);
    kd_batch &v = f.view;
    v.n_reads = f.contig.size();
    v.contig = f.contig.data(); v.pos0 = f.pos0.data(); v.flag = f.flag.data(); v.seq_off = f.seq_off.data();
    v.seq_len = f.seq_len.data(); v.cig_off = f.cig_off.data(); v.n_cig = f.n_cig.data();
    v.seq4 = f.seq4.data(); v.seq4_bytes = f.seq4.size() - 8;
    v.cigar = f.cigar.data(); v.cigar_words = f.cigar.size() - 2;
}

// BAM header of the uncompressed stream d[0, n): reference names / lengths -> f, *o = offset of the first record.
// KD_E_IO + g_decode_error on a malformed header; `need_more` is set when the header is merely cut off at n.
int parse_bam_header(const uint8_t *d, size_t n, std::vector<std::string> &names, std::vector<uint32_t> &lens, size_t *o_out,
                     bool *need_more) {
    *need_more = false;
    if (n < 12) { *need_more = true; g_decode_error = "truncated BAM header"; return KD_E_IO; }
    if (memcmp(d, "BAM\1", 4) != 0) { g_decode_error = "not a BAM stream"; return KD_E_IO; }
    size_t o = 8 + (size_t)rd32(d + 4);
    if (o + 4 > n) { *need_more = true; g_decode_error = "truncated BAM header"; return KD_E_IO; }
    const uint32_t n_ref = rd32(d + o);
    o += 4;
    for (uint32_t r = 0; r < n_ref; r++) {
        if (o + 4 > n) { *need_more = true; g_decode_error = "truncated BAM reference list"; return KD_E_IO; }
        const uint32_t l_name = rd32(d + o);
        if (!l_name) { g_decode_error = "malformed BAM reference list"; return KD_E_IO; }
        if (o + 8 + l_name > n) { *need_more = true; g_decode_error = "truncated BAM reference list"; return KD_E_IO; }
        names.emplace_back((const char *)d + o + 4, l_name - 1);
        lens.push_back(rd32(d + o + 4 + l_name));
        o += 8 + l_name;
    }
    *o_out = o;
    return KD_OK;
}

// The records of d[o, n) -> the SoA arrays of f (replaced, not appended).  final: the stream ends at n, a cut-off record
// is an error; otherwise the walk stops in front of it and *consumed tells where (the caller carries the rest over into
// the next chunk).
// Ranges of the record stream that are PRODUCED by the worker that then walks them (kd_stream: a range = a run of whole BGZF
// blocks, inflated by the worker itself, so that the block_size chain is followed through memory the same core just wrote --
// walking ranges other cores had inflated cost 240 ns per record on the GPU boxes' hosts, more CPU time than the inflate).
struct RangePlan {
    std::vector<size_t> bound;               // range t = bytes [bound[t], bound[t + 1]) of d; bound[0] = first record, bound.back() = d.size()
    std::function<bool(unsigned)> make;      // make(t): bring the bytes of range t into d; false = failed
};

// Does d[q, n) begin with a well-formed BAM record (block_size, refID, pos, l_read_name, the lengths adding up, the read name
// NUL-terminated)?  -> the offset of the record behind it, 0 if not, or KD_REC_UNKNOWN when too little of the record lies in front of n
// to tell.  The speculative record starts of the parallel walk (below) and of a span of a file (kd_decode_open_span) use it.
constexpr size_t KD_REC_UNKNOWN = ~(size_t)0;
constexpr int KD_SPEC_CHAIN = 16;
inline size_t bam_record_plausible(const uint8_t *d, size_t n, size_t q, uint32_t n_ref) {
    if (q + 36 > n) return KD_REC_UNKNOWN;
    const uint32_t bs = rd32(d + q);
    if (bs < 32) return 0;
    const uint8_t *r = d + q + 4;
    const int32_t refid = (int32_t)rd32(r), pos = (int32_t)rd32(r + 4);
    const uint32_t l_rn = r[8], n_cig = rd16(r + 12), l_seq = rd32(r + 16);
    if (refid < -1 || (refid >= 0 && (uint32_t)refid >= n_ref) || pos < -1 || l_rn == 0) return 0;
    if (32 + (size_t)l_rn + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq > bs) return 0;
    if (q + 4 + 32 + (size_t)l_rn > n) return KD_REC_UNKNOWN;
    if (r[32 + l_rn - 1] != 0) return 0;        // read name is NUL-terminated
    return q + 4 + bs;
}

int parse_bam_records(const Arr<uint8_t> &d, size_t o, uint32_t n_ref, File &f, int n_threads, bool final, size_t *consumed,
                      const RangePlan *plan = nullptr, size_t n_limit = 0) {
    const size_t n = n_limit ? n_limit : d.size();   // (n_limit: the stream of records ends inside the buffer -- a span of a file)
    // pass 1 (touches 4 + 20 bytes per record): follow the block_size chain, record where every kept record starts and
    // the running totals of packed-base bytes / CIGAR words.  The chain is sequential by nature (a record's length
    // says where the next one begins), and at ~75 ns per record it was most of the decode time, so it is walked IN
    // PARALLEL over ranges of the stream: the worker of range t looks for its first record speculatively -- the first
    // offset from which KD_SPEC_CHAIN consecutive records look well-formed -- and walks from there; afterwards the
    // hand-offs are verified (the walk of range t-1 must END exactly on the start range t guessed), and any range whose
    // guess was wrong is re-walked from the true position.  The result is identical to the sequential walk.
    // record body offset, packed-base bytes, CIGAR words, and -- for a read with more than 65535 CIGAR operations, which
    // BAM stores as the placeholder <l_seq>S<ref_len>N with the real CIGAR in the CG:B,I tag (SAMv1 4.2.2) -- the offset
    // of the tag's array (0 = the in-record CIGAR)
    struct Rec { uint64_t at; uint32_t sb, nc; uint64_t cg_at; };
    // the CG:B,I array of the record body r (bs bytes), if its CIGAR is the placeholder: -> element count, *at = offset
    auto real_cigar = [&](const uint8_t *r, uint32_t bs, uint32_t l_rn, uint32_t n_cig, uint32_t l_seq, uint64_t *at) -> uint32_t {
        *at = 0;
        if (n_cig != 2) return n_cig;
        const uint8_t *cg = r + 32 + l_rn;
        const uint32_t c0 = rd32(cg), c1 = rd32(cg + 4);
        if ((c0 & 15u) != 4u || (c0 >> 4) != l_seq || (c1 & 15u) != 3u) return n_cig;
        size_t a = 32 + (size_t)l_rn + 8 + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
        while (a + 3 <= bs) {
            const uint8_t t0 = r[a], t1 = r[a + 1], ty = r[a + 2];
            a += 3;
            size_t len = 0;
            if (ty == 'A' || ty == 'c' || ty == 'C') len = 1;
            else if (ty == 's' || ty == 'S') len = 2;
            else if (ty == 'i' || ty == 'I' || ty == 'f') len = 4;
            else if (ty == 'Z' || ty == 'H') { while (a + len < bs && r[a + len]) len++; len++; }
            else if (ty == 'B') {
                if (a + 5 > bs) return n_cig;
                const uint8_t sub = r[a];
                const uint32_t cnt = rd32(r + a + 1);
                const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                if (t0 == 'C' && t1 == 'G' && sub == 'I' && a + 5 + 4 * (size_t)cnt <= bs) {
                    *at = (uint64_t)(r - d.data()) + a + 5;
                    return cnt;
                }
                len = 5 + es * (size_t)cnt;
            } else return n_cig;   // unknown type: cannot skip it
            a += len;
        }
        return n_cig;
    };
    // While a worker walks its own range the bytes behind the range's end may not exist yet (lim < n): a record that starts
    // within KD_EDGE bytes of lim is left to the hand-off pass below, which runs when everything is there.
    constexpr size_t KD_EDGE = 4 + 32 + 255 + 8, KD_UNKNOWN = ~(size_t)0;
    auto plausible = [&](size_t q, size_t lim) -> size_t {      // 0, KD_UNKNOWN, or the offset of the record after the one at q
        if (lim < n && q + KD_EDGE > lim) return KD_UNKNOWN;
        if (q + 36 > n) return 0;
        const uint32_t bs = rd32(d.data() + q);
        if (bs < 32 || q + 4 + (size_t)bs > n) return 0;
        const uint8_t *r = d.data() + q + 4;
        const int32_t refid = (int32_t)rd32(r), pos = (int32_t)rd32(r + 4);
        const uint32_t l_rn = r[8], n_cig = rd16(r + 12), l_seq = rd32(r + 16);
        if (refid < -1 || (refid >= 0 && (uint32_t)refid >= n_ref) || pos < -1 || l_rn == 0) return 0;
        if (32 + (size_t)l_rn + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq > bs) return 0;
        if (r[32 + l_rn - 1] != 0) return 0;        // read name is NUL-terminated
        return q + 4 + bs;
    };
    // (sums[0] / sums[1]: packed-base bytes / CIGAR words of the kept records, for the prefix over the ranges)
    auto walk = [&](size_t p0, size_t p_end, size_t lim, std::vector<Rec> &out, uint64_t &n_rec, size_t &stop, std::string &err, size_t *sums) -> bool {
        size_t q = p0;
        sums[0] = 0; sums[1] = 0;
        while (q < p_end && q + 4 <= n) {
            if (lim < n && q + KD_EDGE > lim) break;         // header / name / CIGAR words may lie behind what exists: later
            const uint32_t bs = rd32(d.data() + q);
            if (bs < 32) { err = "malformed BAM record"; return false; }
            if (q + 4 + (size_t)bs > n) {
                if (!final) break;                 // cut off by the chunk boundary: the next chunk starts with it
                err = "truncated BAM record"; return false;
            }
            const uint8_t *r = d.data() + q + 4;
            // the chain of block_size fields is a pointer chase through freshly inflated memory: ask for the lines a few
            // records ahead (the records are back to back, so "ahead in bytes" is "ahead in the chain")
            __builtin_prefetch(r + 1024);
            __builtin_prefetch(r + 1088);
            const int32_t refid = (int32_t)rd32(r);
            n_rec++;
            if (refid >= 0) {
                if ((uint32_t)refid >= n_ref) { err = "BAM record with refID out of range"; return false; }
                const uint32_t l_rn = r[8], n_cig = rd16(r + 12), l_seq = rd32(r + 16);
                if (32 + (size_t)l_rn + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 > bs) { err = "malformed BAM record"; return false; }
                if (lim < n && n_cig == 2 && q + 4 + (size_t)bs > lim) { n_rec--; break; }   // (a placeholder CIGAR reads the whole record)
                uint64_t cg_at = 0;
                const uint32_t nc = real_cigar(r, bs, l_rn, n_cig, l_seq, &cg_at);
                out.push_back({q + 4, (uint32_t)(((size_t)l_seq + 1) / 2), nc, cg_at});
                sums[0] += ((size_t)l_seq + 1) / 2; sums[1] += nc;
            }
            q += 4 + bs;
        }
        stop = q;
        return true;
    };
    unsigned nt1 = n_threads > 0 ? (unsigned)n_threads : hw_threads();
    size_t min_range = 1u << 20;                                      // >= 1 MB of records per range ...
    if (const char *e = getenv("KD_DECODE_RANGE_BYTES")) min_range = std::max<size_t>(64, strtoull(e, nullptr, 10));   // ... tests: small
    nt1 = (unsigned)std::max<size_t>(1, std::min<size_t>(nt1, (n - std::min(o, n)) / min_range));
    if (plan) nt1 = (unsigned)(plan->bound.size() - 1);
    // (kept across calls by the calling thread: a stream parses ~70 chunks, and fresh vectors of this size come from mmap --
    //  page faults and munmap for every chunk)
    static thread_local std::vector<std::vector<Rec>> part_of_this_thread;
    std::vector<std::vector<Rec>> &part = part_of_this_thread;   // (the workers must see the CALLER's instance: captured by reference)
    if (part.size() < nt1) part.resize(nt1);
    for (unsigned t = 0; t < nt1; t++) part[t].clear();
    std::vector<uint64_t> part_nrec(nt1, 0);
    std::vector<size_t> part_sums(2 * (size_t)nt1, 0);
    std::vector<size_t> guess(nt1, 0), stop(nt1, 0);
    std::vector<std::string> perr(nt1);
    std::vector<char> pok(nt1, 1);
    const size_t span = (n - o + nt1 - 1) / nt1;
    auto range_begin = [&](unsigned t) { return plan ? plan->bound[t] : o + (size_t)t * span; };
    auto range_end = [&](unsigned t) { return plan ? plan->bound[t + 1] : (t + 1 == nt1 ? n : std::min(n, o + (size_t)(t + 1) * span)); };
    auto work1 = [&](unsigned t) {
        const size_t lim = plan ? range_end(t) : n;       // bytes that exist while this worker runs
        if (plan && !plan->make(t)) { pok[t] = 0; perr[t] = "BGZF inflate failed"; guess[t] = stop[t] = range_end(t); return; }
        size_t q = range_begin(t);
        if (t > 0) {   // speculative start: first offset in the range that begins a chain of well-formed records
            const size_t rend = range_end(t);
            size_t found = ~(size_t)0;
            for (; q < rend && found == ~(size_t)0; q++) {
                size_t z = q;
                int ok = 0;
                bool ran_out = false;      // the chain reached the end of what exists
                while (ok < KD_SPEC_CHAIN) {
                    const size_t nx = plausible(z, lim);
                    if (nx == KD_UNKNOWN) { ran_out = true; break; }
                    if (!nx) break;
                    ok++; z = nx;
                    if (z + 4 > n) { ran_out = true; break; }
                }
                if (ok == KD_SPEC_CHAIN || (ok > 0 && ran_out)) found = q;
                else if (ran_out) break;   // too close to the end to judge: the hand-off pass walks the rest of this range
            }
            if (found == ~(size_t)0) { guess[t] = rend; stop[t] = rend; return; }   // no (judgeable) record start in this range
            q = found;
        }
        guess[t] = q;
        pok[t] = walk(q, range_end(t), lim, part[t], part_nrec[t], stop[t], perr[t], &part_sums[2 * t]) ? 1 : 0;
    };
    static const bool trace_parse = getenv("KD_DECODE_TRACE") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    const unsigned nt_thr = n_threads > 0 ? (unsigned)n_threads : hw_threads();   // (ranges may outnumber the workers)
    pool().run(nt1, nt_thr, work1);
    const auto tp1 = std::chrono::steady_clock::now();
    // verify the hand-offs left to right; re-walk what a wrong guess (or an error seen from a wrong start) spoiled
    size_t p = o;
    for (unsigned t = 0; t < nt1; t++) {
        if (plan && !pok[t] && perr[t] == "BGZF inflate failed") { g_decode_error = perr[t]; return KD_E_IO; }
        if (t > 0 && p < guess[t]) {   // the records a worker left at the end of its range (everything exists now): they belong to range t - 1
            size_t gs[2] = {0, 0}, gstop = p;
            std::string gerr;
            if (!walk(p, guess[t], n, part[t - 1], part_nrec[t - 1], gstop, gerr, gs)) { g_decode_error = gerr; return KD_E_IO; }
            part_sums[2 * (t - 1)] += gs[0]; part_sums[2 * (t - 1) + 1] += gs[1];
            p = gstop;
        }
        if (guess[t] != p || (t > 0 && !pok[t])) {
            part[t].clear(); part_nrec[t] = 0; perr[t].clear(); part_sums[2 * t] = part_sums[2 * t + 1] = 0;
            pok[t] = p >= range_end(t) ? 1 : (walk(p, range_end(t), n, part[t], part_nrec[t], stop[t], perr[t], &part_sums[2 * t]) ? 1 : 0);
            if (p >= range_end(t)) stop[t] = p;
        }
        if (!pok[t]) { g_decode_error = perr[t]; return KD_E_IO; }
        p = stop[t];
    }
    if (plan && p < n) {   // (only when the LAST range had no judgeable start: its records are still to be walked)
        size_t gs[2] = {0, 0}, gstop = p;
        std::string gerr;
        if (!walk(p, n, n, part[nt1 - 1], part_nrec[nt1 - 1], gstop, gerr, gs)) { g_decode_error = gerr; return KD_E_IO; }
        part_sums[2 * (nt1 - 1)] += gs[0]; part_sums[2 * (nt1 - 1) + 1] += gs[1];
        p = gstop;
    }
    if (final && p != n) { g_decode_error = "truncated BAM record"; return KD_E_IO; }
    *consumed = p;
    // prefix over the ranges: first record index / packed-base byte / CIGAR word of each
    std::vector<size_t> k_at(nt1 + 1, 0), sq_at(nt1 + 1, 0), cg_at(nt1 + 1, 0);
    if (f.append) { k_at[0] = f.contig.size(); sq_at[0] = f.seq4.size(); cg_at[0] = f.cigar.size(); }
    for (unsigned t = 0; t < nt1; t++) {
        k_at[t + 1] = k_at[t] + part[t].size(); sq_at[t + 1] = sq_at[t] + part_sums[2 * t]; cg_at[t + 1] = cg_at[t] + part_sums[2 * t + 1];
        f.n_records += part_nrec[t];
    }
    const size_t n_keep = k_at[nt1];
    f.contig.resize(n_keep); f.pos0.resize(n_keep); f.flag.resize(n_keep); f.seq_off.resize(n_keep);
    f.seq_len.resize(n_keep); f.cig_off.resize(n_keep); f.n_cig.resize(n_keep);
    f.seq4.resize(sq_at[nt1]); f.cigar.resize(cg_at[nt1]);
    // pass 2 (parallel, one worker per range): every record writes its own slots of the SoA arrays
    auto fill = [&](unsigned t) {
        size_t k = k_at[t], so = sq_at[t], co = cg_at[t];
        const std::vector<Rec> &pt = part[t];
        for (size_t ri = 0; ri < pt.size(); ri++) {
            const Rec &rc = pt[ri];
            const uint8_t *r = d.data() + rc.at;
            if (ri + 6 < pt.size()) { const uint8_t *nx = d.data() + pt[ri + 6].at; __builtin_prefetch(nx); __builtin_prefetch(nx + 64); }
            const uint32_t l_rn = r[8], n_cig = rd16(r + 12), l_seq = rd32(r + 16);
            f.contig[k] = rd32(r);
            f.pos0[k] = (int32_t)rd32(r + 4);
            f.flag[k] = rd16(r + 14);
            f.seq_len[k] = l_seq;
            f.n_cig[k] = rc.nc;
            f.cig_off[k] = co;
            f.seq_off[k] = so;
            const uint8_t *cg = r + 32 + l_rn;
            const uint8_t *cgs = rc.cg_at ? d.data() + rc.cg_at : cg;   // CG:B,I tag or the in-record CIGAR
            for (uint32_t c = 0; c < rc.nc; c++) f.cigar[co + c] = rd32(cgs + 4 * c);
            const size_t sb = ((size_t)l_seq + 1) / 2;
            memcpy(f.seq4.data() + so, cg + 4 * (size_t)n_cig, sb);
            if (l_seq & 1) f.seq4[so + sb - 1] &= 0xf0;
            k++; so += sb; co += rc.nc;
        }
    };
    const auto tp2 = std::chrono::steady_clock::now();
    pool().run(nt1, nt_thr, fill);
    if (trace_parse) {
        const auto tp3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "kd parse: %u ranges: %s %.2f ms, hand-offs + prefix + resize %.2f ms, fill %.2f ms\n", nt1,
                plan ? "inflate + walk" : "walk", ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3));
    }
    return KD_OK;
}

// Records of one range of a SAM file (parallel parse), or of the whole file.
struct SamPart {
    Arr<uint32_t> contig, flag, seq_len, n_cig, cigar;
    Arr<int32_t> pos0;
    Arr<uint64_t> seq_off, cig_off;   // relative to the part
    Arr<uint8_t> seq4;
    uint64_t n_records = 0;
    std::string err;
    int err_code = KD_E_IO;
};

// character -> BAM base code / CIGAR op code; built once (C++11 static initialisation is thread-safe)
struct SamTables {
    int8_t nib[256], op[256];
    uint8_t known[256];
    SamTables() {
        memset(nib, 0, sizeof nib);     // characters outside the BAM alphabet -> '=' (0): a KeyError in M / clip context
        memset(op, 15, sizeof op);      // unknown CIGAR letters are ignored by the reference's if/elif chain
        memset(known, 0, sizeof known);
        const char *nibs = "=ACMGRSVTWYHKDBN";
        for (int i = 0; i < 16; i++) {
            nib[(uint8_t)nibs[i]] = (int8_t)i; nib[(uint8_t)tolower(nibs[i])] = (int8_t)i;
            known[(uint8_t)nibs[i]] = 1; known[(uint8_t)tolower(nibs[i])] = 1;
        }
        const char *ops = "MIDNSHP=X";
        for (int i = 0; i < 9; i++) op[(uint8_t)ops[i]] = (int8_t)i;
    }
};
const SamTables &sam_tables() { static const SamTables TB; return TB; }

typedef std::unordered_map<std::string, uint32_t> SamIds;

// '@' line: only @SQ matters (kindel.py:138-141)
bool sam_header_line(const char *p, const char *e, SamIds &ids, std::vector<std::string> &names, std::vector<uint32_t> &lens) {
    if (e - p >= 3 && p[1] == 'S' && p[2] == 'Q') {
        std::string name;
        long ln = -1;
        const char *q = p;
        while (q < e) {
            const char *t = (const char *)memchr(q, '\t', (size_t)(e - q));
            const char *fe = t ? t : e;
            if (fe - q > 3 && q[0] == 'S' && q[1] == 'N' && q[2] == ':') name.assign(q + 3, fe);
            if (fe - q > 3 && q[0] == 'L' && q[1] == 'N' && q[2] == ':') ln = strtol(std::string(q + 3, fe).c_str(), nullptr, 10);
            q = t ? t + 1 : e;
        }
        if (name.empty() || ln < 0) { g_decode_error = "@SQ line without SN/LN"; return false; }
        ids[name] = (uint32_t)names.size();
        names.push_back(name);
        lens.push_back((uint32_t)ln);
    }
    return true;
}

// alignment line [p, e) -> one record of `o` (or dropped: RNAME '*', kindel.py:147-148); false = error in o.err
bool sam_record_line(const char *p, const char *e, const SamIds &ids, SamPart &o) {
    const SamTables &TB = sam_tables();
    const int8_t *nibtab = TB.nib, *optab = TB.op;
    const char *fld[11];
    int nf = 0;
    const char *q = p;
    while (nf < 11 && q <= e) {
        fld[nf++] = q;
        const char *t = (const char *)memchr(q, '\t', (size_t)(e - q));
        if (!t) break;
        q = t + 1;
    }
    if (nf < 10) { o.err = "SAM record with fewer than 10 fields"; return false; }
    auto flen = [&](int i) { return (size_t)((i + 1 < nf ? fld[i + 1] - 1 : e) - fld[i]); };
    o.n_records++;
    std::string rname(fld[2], flen(2));
    if (rname == "*") return true;
    auto it = ids.find(rname);
    if (it == ids.end()) { o.err = rname; o.err_code = KD_E_NOREF; return false; }   // refs_lens[ref_id], kindel.py:151
    o.contig.push_back(it->second);
    o.flag.push_back((uint32_t)strtoul(std::string(fld[1], flen(1)).c_str(), nullptr, 10));
    o.pos0.push_back((int32_t)(strtol(std::string(fld[3], flen(3)).c_str(), nullptr, 10) - 1));
    o.cig_off.push_back(o.cigar.size());
    uint32_t nc = 0;
    const char *c = fld[5], *ce = c + flen(5);
    if (!(ce - c == 1 && *c == '*')) {
        uint64_t num = 0;
        for (; c < ce; c++) {
            if (*c >= '0' && *c <= '9') num = num * 10 + (uint64_t)(*c - '0');
            else {
                if (num >= (1ULL << 28)) { o.err = "CIGAR length too large"; return false; }
                o.cigar.push_back((uint32_t)(num << 4) | (uint32_t)(uint8_t)optab[(uint8_t)*c]);
                num = 0; nc++;
            }
        }
    }
    o.n_cig.push_back(nc);
    const char *sq = fld[9];
    size_t sl = flen(9);
    if (sl == 1 && *sq == '*') sl = 0;
    o.seq_off.push_back(o.seq4.size());
    o.seq_len.push_back((uint32_t)sl);
    bool all_known = true;
    for (size_t i = 0; i < sl; i += 2) {
        const uint8_t hi = (uint8_t)nibtab[(uint8_t)sq[i]];
        const uint8_t lo = i + 1 < sl ? (uint8_t)nibtab[(uint8_t)sq[i + 1]] : 0;
        all_known = all_known && TB.known[(uint8_t)sq[i]] && (i + 1 >= sl || TB.known[(uint8_t)sq[i + 1]]);
        o.seq4.push_back((uint8_t)(hi << 4 | lo));
    }
    if (!all_known) {
        // A character outside "=ACMGRSVTWYHKDBN" cannot be stored in 4 bits.  Inside M / clip context it is a KeyError like
        // any non-ACGTN base (code 0 raises it); inside an INSERTION the reference keeps the text verbatim
        // (kindel.py:55-58, no alphabet check), which this encoding cannot reproduce: refuse loudly.
        const uint32_t *cw = o.cigar.data() + o.cig_off[o.cig_off.size() - 1];
        size_t qpos = 0;
        for (uint32_t k = 0; k < nc; k++) {
            const size_t len = cw[k] >> 4;
            const uint32_t op = cw[k] & 15u;
            if (op == 1)
                for (size_t x = qpos; x < qpos + len && x < sl; x++)
                    if (!TB.known[(uint8_t)sq[x]]) {
                        o.err = std::string("insertion contains '") + sq[x] + "', a character outside the BAM base alphabet (=ACMGRSVTWYHKDBN)";
                        return false;
                    }
            if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) qpos += len;
        }
    }
    return true;
}

// The leading header block of SAM text [base, end): fills ids / names / lens, -> the first alignment line (or end)
int parse_sam_header(const char *base, const char *end, SamIds &ids, std::vector<std::string> &names, std::vector<uint32_t> &lens,
                     const char **rec0) {
    const char *p = base;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        const char *e = le;
        if (e > p && e[-1] == '\r') e--;
        if (e > p) {
            if (*p != '@') break;
            if (!sam_header_line(p, e, ids, names, lens)) return KD_E_IO;
        }
        if (!nl) { p = end; break; }
        p = nl + 1;
    }
    *rec0 = p;
    return KD_OK;
}

// The alignment lines of [rec0, end) (whole lines) -> the SoA arrays of f (replaced), parsed in parallel line-aligned ranges
int parse_sam_records(const char *rec0, const char *end, const SamIds &ids, File &f, int n_threads) {
    auto lines = [&](const char *p, const char *pe, SamPart &o) -> bool {
        while (p < pe) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(pe - p));
            const char *le = nl ? nl : pe;
            const char *e = le;
            if (e > p && e[-1] == '\r') e--;
            if (e > p) {
                if (*p == '@') { o.err = "header line after the first alignment line"; return false; }
                if (!sam_record_line(p, e, ids, o)) return false;
            }
            if (!nl) break;
            p = nl + 1;
        }
        return true;
    };
    size_t min_range = 4u << 20;
    if (const char *ev = getenv("KD_DECODE_RANGE_BYTES")) min_range = std::max<size_t>(64, strtoull(ev, nullptr, 10));
    unsigned nt = n_threads > 0 ? (unsigned)n_threads : hw_threads();
    nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, (size_t)(end - rec0) / min_range));
    std::vector<const char *> cut(nt + 1, end);
    cut[0] = rec0;
    for (unsigned t = 1; t < nt; t++) {   // range starts on the line after the proportional split point
        const char *q = rec0 + (size_t)(end - rec0) / nt * t;
        if (q < cut[t - 1]) q = cut[t - 1];
        const char *nl = q < end ? (const char *)memchr(q, '\n', (size_t)(end - q)) : nullptr;
        cut[t] = nl ? nl + 1 : end;
    }
    std::vector<SamPart> part(nt);
    std::vector<char> ok(nt, 1);
    {
        std::vector<std::thread> th;
        auto work = [&](unsigned t) { ok[t] = lines(cut[t], cut[t + 1], part[t]) ? 1 : 0; };
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    for (unsigned t = 0; t < nt; t++)
        if (!ok[t]) { g_decode_error = part[t].err; return part[t].err_code; }   // the first failing range = the first failing line
    std::vector<size_t> k_at(nt + 1, 0), sq_at(nt + 1, 0), cg_at(nt + 1, 0);
    for (unsigned t = 0; t < nt; t++) {
        k_at[t + 1] = k_at[t] + part[t].contig.size(); sq_at[t + 1] = sq_at[t] + part[t].seq4.size();
        cg_at[t + 1] = cg_at[t] + part[t].cigar.size();
        f.n_records += part[t].n_records;
    }
    const size_t n_keep = k_at[nt];
    f.contig.resize(n_keep); f.pos0.resize(n_keep); f.flag.resize(n_keep); f.seq_off.resize(n_keep);
    f.seq_len.resize(n_keep); f.cig_off.resize(n_keep); f.n_cig.resize(n_keep);
    f.seq4.resize(sq_at[nt]); f.cigar.resize(cg_at[nt]);
    {
        auto merge = [&](unsigned t) {
            const SamPart &o = part[t];
            const size_t k0 = k_at[t], m = o.contig.size();
            if (m) {
                memcpy(f.contig.data() + k0, o.contig.data(), m * 4); memcpy(f.flag.data() + k0, o.flag.data(), m * 4);
                memcpy(f.seq_len.data() + k0, o.seq_len.data(), m * 4); memcpy(f.n_cig.data() + k0, o.n_cig.data(), m * 4);
                memcpy(f.pos0.data() + k0, o.pos0.data(), m * 4);
                for (size_t k = 0; k < m; k++) { f.seq_off[k0 + k] = o.seq_off[k] + sq_at[t]; f.cig_off[k0 + k] = o.cig_off[k] + cg_at[t]; }
            }
            if (o.seq4.size()) memcpy(f.seq4.data() + sq_at[t], o.seq4.data(), o.seq4.size());
            if (o.cigar.size()) memcpy(f.cigar.data() + cg_at[t], o.cigar.data(), o.cigar.size() * 4);
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(merge, t);
        merge(0);
        for (auto &x : th) x.join();
    }
    return KD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Chunked reading: the file as a sequence of batches of about `chunk_bytes` uncompressed bytes each (whole records).
// BGZF: the blocks of a chunk are inflated in parallel behind the carried-over tail of the previous chunk (the record the
// chunk boundary cut); SAM text: chunks end on line boundaries; any other gzip stream: one chunk.
// ---------------------------------------------------------------------------------------------------------------------
struct RawView {   // the compressed file: mapped read-only (the workers fault its pages in as they inflate), or read as a fallback
    const uint8_t *p = nullptr;
    size_t n = 0;
    void *map = nullptr;
    Arr<uint8_t> own;
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
    uint8_t operator[](size_t i) const { return p[i]; }
    bool open(const char *path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                (void)madvise(m, (size_t)st.st_size, MADV_WILLNEED);
                map = m; p = (const uint8_t *)m; n = (size_t)st.st_size;
                ::close(fd);
                return true;
            }
        }
        const bool ok = read_all(fd, own);   // (the SAME descriptor: a pipe's writer sees its reader leave only once)
        ::close(fd);
        if (!ok) return false;
        p = own.data(); n = own.size();
        return true;
    }
    ~RawView() { if (map) munmap(map, n); }
};

struct Stream {
    RawView raw;
    int n_threads = 0;
    size_t chunk_bytes = 0;
    bool is_text = false, bgzf = false, done = false;
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    uint64_t n_records = 0;
    // BGZF / BAM
    std::vector<Block> blocks;          // scanned so far
    size_t scan_off = 0, scan_total = 0;
    bool scan_done = false;
    size_t next_block = 0;
    Arr<uint8_t> buf, carry;   // uncompressed bytes of the current chunk (carry + blocks); the cut-off record of the last one
    uint32_t n_ref = 0;
    bool header_done = false;
    // a SPAN of the file (kd_decode_open_span): records from absolute uncompressed offset span_begin (inside block next_block)
    // up to span_end; blocks from span_blk_end on are not touched.  Defaults: the whole file.
    size_t span_skip = 0, span_end = ~(size_t)0, span_blk_end = ~(size_t)0;
    Arr<uint8_t> whole;        // non-BGZF gzip: the whole stream
    // SAM
    SamIds ids;
    const char *sam_pos = nullptr, *sam_end = nullptr;

    // blocks covering at least `want` uncompressed bytes behind next_block, or up to the end of the file
    bool scan_ahead(size_t want) {
        size_t have = 0;
        for (size_t b = next_block; b < blocks.size(); b++) have += blocks[b].out_len;
        while (!scan_done && have < want) {
            const size_t before = scan_total;
            if (!scan_bgzf_some(raw.data(), raw.size(), &scan_off, blocks, &scan_total, want - have + 1)) return false;
            have += scan_total - before;
            if (scan_off == raw.size()) scan_done = true;
            else if (scan_total == before && scan_off + 18 > raw.size()) return false;   // trailing garbage shorter than a block header
        }
        return true;
    }

    int open(const char *path, int threads, size_t chunk) {
        n_threads = threads; chunk_bytes = chunk ? chunk : (size_t)64 << 20;
        if (!raw.open(path)) { g_decode_error = std::string("cannot read ") + path; return KD_E_IO; }
        if (raw.size() >= 2 && raw[0] == 0x1f && raw[1] == 0x8b) {
            // BGZF if the first block says so (a later block that does not parse is then a corrupt file, not another format)
            std::vector<Block> probe;
            size_t po = 0, pt = 0;
            bgzf = scan_bgzf_some(raw.data(), raw.size(), &po, probe, &pt, 1) && !probe.empty();
            if (!bgzf) {
                Arr<uint8_t> all;     // (inflate_generic wants the bytes in an Arr)
                if (!all.resize(raw.size())) { g_decode_error = "out of memory"; return KD_E_NOMEM; }
                memcpy(all.data(), raw.data(), raw.size());
                if (!inflate_generic(all, whole)) { g_decode_error = "gzip inflate failed"; return KD_E_IO; }
            }
            // the header may span several blocks: inflate until it parses
            return bgzf ? KD_OK : header_from(whole.data(), whole.size(), true);
        }
        is_text = true;
        sam_end = (const char *)raw.data() + raw.size();
        return parse_sam_header((const char *)raw.data(), sam_end, ids, names, lens, &sam_pos);
    }
    // BGZF: the header sits in the first blocks; read it at open so that the contig table is known before the first batch
    // (next() parses it again from the first chunk, and skips it)
    int preload_header() {
        if (is_text || header_done) return KD_OK;
        int rc = KD_OK;
        size_t b1 = 0, want = 1;
        while (!header_done) {
            if (!scan_ahead(want)) { rc = KD_E_IO; g_decode_error = "corrupt BGZF block header"; break; }   // (the blocks are scanned as needed)
            if (b1 >= blocks.size()) break;
            want += blocks[b1++].out_len + 1;
            if (!inflate_blocks(0, b1, 0)) { rc = KD_E_IO; g_decode_error = "BGZF inflate failed"; break; }
            const int hr = header_from(buf.data(), buf.size(), scan_done && b1 >= blocks.size());
            if (hr != 1 && hr != KD_OK) { rc = hr; break; }
        }
        if (!rc && !header_done) { rc = KD_E_IO; g_decode_error = "truncated BAM header"; }
        header_done = false;
        return rc;
    }
    size_t hdr_end = 0;
    int header_from(const uint8_t *d, size_t n, bool final) {
        bool more = false;
        names.clear(); lens.clear();
        int rc = parse_bam_header(d, n, names, lens, &hdr_end, &more);
        if (rc && more && !final) return 1;   // need more bytes
        if (rc) return rc;
        n_ref = (uint32_t)names.size();
        header_done = true;
        return KD_OK;
    }
    // inflate blocks [b0, b1) behind the first `keep` bytes of buf
    bool inflate_blocks(size_t b0, size_t b1, size_t keep) {
        size_t add = 0;
        for (size_t b = b0; b < b1; b++) add += blocks[b].out_len;
        if (!buf.resize(keep + add)) return false;
        std::vector<size_t> at(b1 - b0);
        size_t o = keep;
        for (size_t b = b0; b < b1; b++) { at[b - b0] = o; o += blocks[b].out_len; }
        std::atomic<bool> ok{true};
        unsigned nt = n_threads > 0 ? (unsigned)n_threads : hw_threads();
        pool().run((unsigned)(b1 - b0), nt, [&](unsigned k) {
            const size_t b = b0 + k;
            if (!inflate_raw(raw.data() + blocks[b].in_off, blocks[b].in_len, buf.data() + at[k], blocks[b].out_len)) ok = false;
        });
        return ok;
    }
    // next batch into f (arrays replaced); *got = false at the end of the file
    int next(File &f, bool *got) {
        *got = false;
        if (!f.append) f.n_records = 0;
        if (done) return KD_OK;
        if (is_text) {
            if (sam_pos >= sam_end) { done = true; return KD_OK; }
            const char *e = sam_pos + std::min<size_t>(chunk_bytes, (size_t)(sam_end - sam_pos));
            if (e < sam_end) {
                const char *nl = (const char *)memchr(e, '\n', (size_t)(sam_end - e));
                e = nl ? nl + 1 : sam_end;
            }
            int rc = parse_sam_records(sam_pos, e, ids, f, n_threads);
            if (rc) return rc;
            sam_pos = e;
            n_records += f.n_records;
            *got = true;
            return KD_OK;
        }
        if (!bgzf) {   // one chunk
            if (!header_done) return KD_E_IO;
            size_t used = 0;
            int rc = parse_bam_records(whole, hdr_end, n_ref, f, n_threads, true, &used);
            if (rc) return rc;
            done = true; n_records += f.n_records; *got = true;
            return KD_OK;
        }
        for (;;) {
            const size_t kept_before = f.append ? f.contig.size() : 0;
            const uint64_t recs_before = f.append ? f.n_records : 0;
            if (!scan_ahead(chunk_bytes)) { g_decode_error = "corrupt BGZF block header"; return KD_E_IO; }
            const size_t blk_end = std::min(blocks.size(), span_blk_end);   // (a span stops in front of span_blk_end)
            const bool all_scanned = scan_done || span_blk_end <= blocks.size();
            if (next_block >= blk_end && all_scanned && carry.size() == 0 && header_done) { done = true; return KD_OK; }
            // blocks of this chunk
            size_t b1 = next_block, add = 0;
            const size_t b0c = next_block;
            while (b1 < blk_end && (add < chunk_bytes || b1 == next_block)) add += blocks[b1++].out_len;
            const size_t keep = carry.size();
            static const bool trace = getenv("KD_DECODE_TRACE") != nullptr;
            const auto tt0 = std::chrono::steady_clock::now();
            size_t used = 0;
            auto tt1 = tt0;
            bool final;
            // (tests / measurements: the two-phase path.  Never for a SPAN -- kd_decode_open_span: only the fused branch knows where a
            // span begins inside its first block and where it must end)
            const bool unfused = getenv("KD_DECODE_UNFUSED") != nullptr && span_end == ~(size_t)0 && span_skip == 0;
            if (header_done && !unfused) {
                // Records only: every worker inflates a run of whole blocks and walks the records in them straight away
                // (RangePlan); what a worker cannot judge at the end of its run is walked by the hand-off pass.
                if (!buf.resize(keep + add)) { g_decode_error = "out of memory"; return KD_E_NOMEM; }
                if (keep) memcpy(buf.data(), carry.data(), keep);
                const size_t nb = b1 - next_block;
                std::vector<size_t> at(nb + 1);
                at[0] = keep;
                for (size_t k = 0; k < nb; k++) at[k + 1] = at[k] + blocks[next_block + k].out_len;
                unsigned nt = n_threads > 0 ? (unsigned)n_threads : hw_threads();
                size_t min_range = 1u << 20;
                if (const char *e = getenv("KD_DECODE_RANGE_BYTES")) min_range = std::max<size_t>(64, strtoull(e, nullptr, 10));
                size_t per_thread = 2;          // ranges per worker: finer = better balance when the cores are shared, more hand-offs
                if (const char *e = getenv("KD_DECODE_RANGES_PER_THREAD")) per_thread = std::max<size_t>(1, strtoull(e, nullptr, 10));
                const size_t nr = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(per_thread * (size_t)nt, nb), add / min_range));
                RangePlan plan;
                plan.bound.resize(nr + 1);
                std::vector<size_t> first(nr + 1);
                for (size_t t = 0; t <= nr; t++) { first[t] = nb * t / nr; plan.bound[t] = at[first[t]]; }
                plan.bound[0] = 0;                      // the carried-over bytes belong to the first range
                for (size_t t = 0; t <= nr; t++) plan.bound[t] = std::max(plan.bound[t], span_skip);   // (a span begins inside its first block)
                plan.bound[nr] = at[nb];
                span_skip = 0;
                const size_t b0 = next_block;
                plan.make = [&, b0](unsigned t) -> bool {
                    for (size_t k = first[t]; k < first[t + 1]; k++)
                        if (!inflate_raw(raw.data() + blocks[b0 + k].in_off, blocks[b0 + k].in_len, buf.data() + at[k], blocks[b0 + k].out_len)) return false;
                    return true;
                };
                next_block = b1;
                final = all_scanned && next_block >= blk_end;
                // the last chunk of a span ends at span_end, which lies inside its last block: the walk must land exactly there
                size_t limit = 0;
                if (final && span_end != ~(size_t)0 && nb) {
                    limit = keep + (span_end - blocks[b0].out_off);
                    if (limit > buf.size()) limit = buf.size();
                    for (size_t t = 0; t <= nr; t++) plan.bound[t] = std::min(plan.bound[t], limit);
                }
                int rc = parse_bam_records(buf, plan.bound[0], n_ref, f, n_threads, final, &used, &plan, limit);
                if (rc) return rc;
                if (limit) used = buf.size();           // (what lies behind the span's end belongs to the next span)
            } else {
                if (!buf.resize(keep)) { g_decode_error = "out of memory"; return KD_E_NOMEM; }
                if (keep) memcpy(buf.data(), carry.data(), keep);
                if (!inflate_blocks(next_block, b1, keep)) { g_decode_error = "BGZF inflate failed"; return KD_E_IO; }
                tt1 = std::chrono::steady_clock::now();
                next_block = b1;
                final = all_scanned && next_block >= blk_end;
                size_t o = 0;
                if (!header_done) {
                    int rc = header_from(buf.data(), buf.size(), final);
                    if (rc == 1) { carry.resize(buf.size()); memcpy(carry.data(), buf.data(), buf.size()); continue; }   // header not complete yet
                    if (rc) return rc;
                    o = hdr_end;
                }
                int rc = parse_bam_records(buf, o, n_ref, f, n_threads, final, &used);
                if (rc) return rc;
            }
            if (trace) {
                const auto tt2 = std::chrono::steady_clock::now();
                fprintf(stderr, "kd_stream chunk: %zu blocks %zu bytes: inflate %.1f ms, parse %.1f ms, %zu records\n", b1 - b0c, buf.size(),
                        std::chrono::duration<double, std::milli>(tt1 - tt0).count(), std::chrono::duration<double, std::milli>(tt2 - tt1).count(), f.contig.size());
            }
            carry.resize(buf.size() - used);
            if (carry.size()) memcpy(carry.data(), buf.data() + used, carry.size());
            n_records += f.append ? f.n_records - recs_before : f.n_records;
            const size_t kept_now = f.contig.size() - kept_before;
            if (kept_now == 0 && !final) continue;   // a chunk smaller than one record: keep reading
            if (final) { done = true; }
            *got = kept_now > 0;
            return KD_OK;
        }
    }
};

}  // namespace

struct kd_file {
    File f;
};

extern "C" {

// The whole file as ONE batch: the stream's chunks (64 MiB of records each, workers inflating and walking their own runs of
// blocks) appended to one another -- no 4 GB buffer of inflated records in between.  KD_DECODE_ONE_CHUNK=1: the old way (tests).
int kd_decode_open(kd_file **out, const char *path, int n_threads) {
    if (!out || !path) return KD_E_ARG;
    *out = nullptr;
    const bool one_chunk = getenv("KD_DECODE_ONE_CHUNK") != nullptr;
    Stream st;
    size_t chunk = (size_t)64 << 20;
    if (const char *e = getenv("KD_DECODE_CHUNK_BYTES")) chunk = std::max<size_t>(1, strtoull(e, nullptr, 10));   // (tests: many chunks)
    int rc = st.open(path, n_threads, one_chunk ? ~(size_t)0 >> 1 : chunk);
    if (rc) return rc;
    std::vector<std::string> names; std::vector<uint32_t> lens;
    if (!one_chunk) {
        if ((rc = st.preload_header())) return rc;
        names = st.names; lens = st.lens;
    }
    kd_file *h = new kd_file();
    File &F = h->f;
    F.contig.resize(0); F.pos0.resize(0); F.flag.resize(0); F.seq_off.resize(0); F.seq_len.resize(0);
    F.cig_off.resize(0); F.n_cig.resize(0); F.seq4.resize(0); F.cigar.resize(0);
    if (st.is_text || !st.bgzf) st.chunk_bytes = ~(size_t)0 >> 1;   // SAM text / plain gzip: one chunk (their parsers replace the arrays)
    else F.append = !one_chunk;                                     // BGZF: every chunk's records land behind the previous ones, in place
    for (;;) {
        bool got = false;
        rc = st.next(F, &got);
        if (rc) { delete h; return rc; }
        if (!got || st.done || !F.append) break;
    }
    F.append = false;
    if (!st.is_text && !st.header_done) { delete h; g_decode_error = "truncated BAM header"; return KD_E_IO; }
    F.names = one_chunk || st.is_text || names.empty() ? st.names : names; F.lens = one_chunk || st.is_text || lens.empty() ? st.lens : lens;
    F.n_records = st.n_records;
    finish_view(F);
    *out = h;
    return KD_OK;
}

// ---- a span of a BGZF / BAM file: the unit of work of one rank in the multi-GPU ingest ---------------------------------
// kd_bgzf_index: the compressed offsets of the file's BGZF blocks (to cut it into byte shares).
int kd_bgzf_index(const char *path, uint64_t *n_blocks, uint64_t *in_off, uint64_t cap) {
    if (!path || !n_blocks) return KD_E_ARG;
    RawView raw;
    if (!raw.open(path)) { g_decode_error = std::string("cannot read ") + path; return KD_E_IO; }
    std::vector<Block> blocks;
    size_t o = 0, total = 0;
    if (raw.size() < 18 || raw[0] != 0x1f || raw[1] != 0x8b || !scan_bgzf_some(raw.data(), raw.size(), &o, blocks, &total, ~(size_t)0 >> 1) ||
        o != raw.size() || blocks.empty()) {
        g_decode_error = "not a BGZF file";
        return KD_E_IO;
    }
    *n_blocks = blocks.size();
    if (in_off)
        for (size_t b = 0; b < blocks.size() && b < cap; b++) in_off[b] = blocks[b].in_off;   // (offset of the block's DEFLATE data)
    return KD_OK;
}

// The records that BEGIN in blocks [block_lo, block_hi) of a BAM file -- precisely: from the first offset at or behind the
// start of block block_lo at which KD_SPEC_CHAIN consecutive well-formed records begin (block 0: the first record behind the
// header) up to the offset found in the same way for block block_hi (the end of the file if there is no such block).  The
// rule is a function of the file alone, so every rank that asks for a boundary gets the same one, and a walk that starts on
// a true record start follows the record chain: it must END exactly on the next boundary -- if a boundary were a fake (sixteen
// plausible records in a row that are not records) the walk would run over it and this call fails with KD_E_IO; the caller
// then falls back to reading the whole file.  info: [0] / [1] absolute uncompressed offsets of the span's first byte / end,
// [2] records walked (incl. RNAME '*'), [3] 1 if the span reaches the end of the file.
int kd_decode_open_span(kd_file **out, const char *path, int n_threads, uint64_t block_lo, uint64_t block_hi, uint64_t *info) {
    if (!out || !path) return KD_E_ARG;
    *out = nullptr;
    Stream st;
    int rc = st.open(path, n_threads, (size_t)64 << 20);
    if (rc) return rc;
    if (!st.bgzf) { g_decode_error = "kd_decode_open_span: not a BGZF-compressed BAM file"; return KD_E_IO; }
    if ((rc = st.preload_header())) return rc;
    const std::vector<std::string> names = st.names;
    const std::vector<uint32_t> lens = st.lens;
    const size_t hdr_end = st.hdr_end;
    if (!st.scan_ahead(~(size_t)0 >> 1) || !st.scan_done) { g_decode_error = "corrupt BGZF block header"; return KD_E_IO; }
    const size_t nb = st.blocks.size();
    const size_t total = nb ? st.blocks[nb - 1].out_off + st.blocks[nb - 1].out_len : 0;
    const uint32_t n_ref = (uint32_t)names.size();
    // first chain start at or behind the start of block b (absolute uncompressed offset), total if there is none
    auto boundary = [&](size_t b, size_t *at) -> int {
        if (b == 0) { *at = hdr_end; return KD_OK; }
        if (b >= nb) { *at = total; return KD_OK; }
        Arr<uint8_t> tmp;
        size_t b1 = b, have = 0, want = (size_t)1 << 18;
        for (;;) {
            while (b1 < nb && have < want) {
                if (!tmp.resize(have + st.blocks[b1].out_len)) { g_decode_error = "out of memory"; return KD_E_NOMEM; }
                if (!inflate_raw(st.raw.data() + st.blocks[b1].in_off, st.blocks[b1].in_len, tmp.data() + have, st.blocks[b1].out_len)) {
                    g_decode_error = "BGZF inflate failed"; return KD_E_IO;
                }
                have += st.blocks[b1].out_len; b1++;
            }
            const bool at_eof = b1 >= nb;
            bool need_more = false;
            for (size_t q = 0; q < have; q++) {
                size_t z = q;
                int ok = 0;
                bool ran_out = false;
                while (ok < KD_SPEC_CHAIN) {
                    if (z == have && at_eof) break;                       // the chain ends with the file
                    const size_t nx = bam_record_plausible(tmp.data(), have, z, n_ref);
                    if (nx == KD_REC_UNKNOWN || (nx && nx > have)) { ran_out = true; break; }
                    if (!nx) break;
                    ok++; z = nx;
                }
                if (ok == KD_SPEC_CHAIN || (ok > 0 && z == have && at_eof)) { *at = st.blocks[b].out_off + q; return KD_OK; }
                if (ran_out) {
                    if (at_eof) continue;        // a record that claims to run past the end of the file: not a record start
                    need_more = true; break;     // cannot judge q with what is inflated: inflate more and start over
                }
            }
            if (!need_more) { *at = total; return KD_OK; }
            want *= 4;
        }
    };
    size_t S = 0, E = 0;
    if ((rc = boundary((size_t)std::min<uint64_t>(block_lo, nb), &S)) || (rc = boundary((size_t)std::min<uint64_t>(block_hi, nb), &E))) return rc;
    kd_file *h = new kd_file();
    File &F = h->f;
    F.contig.resize(0); F.pos0.resize(0); F.flag.resize(0); F.seq_off.resize(0); F.seq_len.resize(0);
    F.cig_off.resize(0); F.n_cig.resize(0); F.seq4.resize(0); F.cigar.resize(0);
    F.names = names; F.lens = lens;
    uint64_t n_rec = 0;
    if (S < E) {
        size_t bs = 0;
        while (bs + 1 < nb && st.blocks[bs + 1].out_off <= S) bs++;     // the block S lies in
        size_t be = bs;
        while (be < nb && st.blocks[be].out_off < E) be++;              // first block that begins at or behind E
        st.next_block = bs; st.span_skip = S - st.blocks[bs].out_off; st.span_end = E; st.span_blk_end = be;
        st.header_done = true; st.n_ref = n_ref; st.names = names; st.lens = lens;
        st.carry.resize(0); st.done = false; st.n_records = 0;
        F.append = true;
        for (;;) {
            bool got = false;
            rc = st.next(F, &got);
            if (rc) { delete h; return rc; }
            if (st.done || (!got && st.next_block >= be)) break;
        }
        F.append = false;
        n_rec = st.n_records;
    }
    F.n_records = n_rec;
    finish_view(F);
    if (info) { info[0] = S; info[1] = E; info[2] = n_rec; info[3] = E >= total ? 1 : 0; }
    *out = h;
    return KD_OK;
}

const char *kd_decode_last_error(void) { return g_decode_error.c_str(); }
/* ---- the host's share of the device-side ingest (kd_ingest.h): file mapped, BGZF block table, BAM header ---- */
struct kd_bgzf_plan {
    RawView raw;
    struct Blk { uint64_t in_off, out_off; uint32_t in_len, out_len; };     // = GiBlock (kd_gpu_inflate.h)
    std::vector<Blk> blocks;
    uint64_t total = 0, hdr_end = 0;
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
};
int kd_bgzf_plan_open(kd_bgzf_plan **out, const char *path) {
    if (!out || !path) return KD_E_ARG;
    *out = nullptr;
    std::unique_ptr<kd_bgzf_plan> P(new kd_bgzf_plan());
    if (!P->raw.open(path)) { g_decode_error = std::string("cannot read ") + path; return KD_E_IO; }
    const uint8_t *raw = P->raw.data();
    const size_t n = P->raw.size();
    if (n < 18 || raw[0] != 0x1f || raw[1] != 0x8b) { g_decode_error = "not a BGZF file (SAM text or uncompressed: the host decoder reads those)"; return KD_E_UNSUPPORTED; }
    // The block scan below is a chain (every header says where the next one is): over a freshly mapped file it takes one page fault per
    // block, one after the other -- 36 ms for the 68 568 blocks of a 2 GB file, a sixth of the device-side ingest (round 6).  All
    // threads touch the mapping's pages first (a byte per 4 KiB, each its own slice): the faults are taken side by side, and neither the
    // scan nor the upload's copies into the pinned pieces meet one afterwards.
    if (P->raw.map && n >= ((size_t)64 << 20)) {
        const unsigned nt = std::max(1u, hw_threads());
        std::vector<std::thread> th;
        std::atomic<unsigned> sink{0};
        const size_t per = ((n / nt) + 4095) & ~(size_t)4095;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                unsigned acc = 0;
                for (size_t o = (size_t)t * per, e = std::min(n, o + per); o < e; o += 4096) acc += raw[o];
                sink.fetch_add(acc, std::memory_order_relaxed);
            });
        for (auto &x : th) x.join();
    }
    std::vector<Block> blocks;
    size_t o = 0, total = 0;
    if (!scan_bgzf_some(raw, n, &o, blocks, &total, ~(size_t)0 >> 1) || o != n) {
        g_decode_error = blocks.empty() ? "not a BGZF file (plain gzip: the host decoder reads it)" : "malformed or truncated BGZF block";
        return blocks.empty() ? KD_E_UNSUPPORTED : KD_E_IO;
    }
    P->total = total;
    // the BAM header: inflate blocks on the host until it is complete
    Arr<uint8_t> head;
    size_t have = 0, b = 0;
    for (;;) {
        bool need_more = false;
        size_t hdr = 0;
        P->names.clear(); P->lens.clear();
        const int rc = have ? parse_bam_header(head.data(), have, P->names, P->lens, &hdr, &need_more) : KD_E_IO;
        if (have && rc == KD_OK) { P->hdr_end = hdr; break; }
        if (have && !need_more) return KD_E_IO;
        if (b >= blocks.size()) { g_decode_error = "truncated BAM header"; return KD_E_IO; }
        if (!head.resize(have + blocks[b].out_len)) { g_decode_error = "out of memory"; return KD_E_NOMEM; }
        if (!inflate_raw(raw + blocks[b].in_off, blocks[b].in_len, head.data() + have, blocks[b].out_len)) { g_decode_error = "BGZF inflate failed"; return KD_E_IO; }
        have += blocks[b].out_len; b++;
    }
    for (const Block &B : blocks)
        if (B.out_len) P->blocks.push_back({(uint64_t)B.in_off, (uint64_t)B.out_off, (uint32_t)B.in_len, (uint32_t)B.out_len});
    *out = P.release();
    return KD_OK;
}
uint32_t kd_bgzf_plan_n_contigs(const kd_bgzf_plan *p) { return p ? (uint32_t)p->lens.size() : 0; }
const char *kd_bgzf_plan_contig_name(const kd_bgzf_plan *p, uint32_t i) { return p && i < p->names.size() ? p->names[i].c_str() : ""; }
uint32_t kd_bgzf_plan_contig_len(const kd_bgzf_plan *p, uint32_t i) { return p && i < p->lens.size() ? p->lens[i] : 0; }
int kd_bgzf_plan_view(const kd_bgzf_plan *p, const uint8_t **file, uint64_t *file_bytes, const void **blocks, uint32_t *n_blocks,
                      uint64_t *total_out, uint64_t *hdr_end) {
    if (!p) return KD_E_ARG;
    if (file) *file = p->raw.data();
    if (file_bytes) *file_bytes = p->raw.size();
    if (blocks) *blocks = p->blocks.data();
    if (n_blocks) *n_blocks = (uint32_t)p->blocks.size();
    if (total_out) *total_out = p->total;
    if (hdr_end) *hdr_end = p->hdr_end;
    return p->blocks.size() > 0xffffffffULL ? KD_E_ARG : KD_OK;
}
void kd_bgzf_plan_close(kd_bgzf_plan *p) { delete p; }

/* host threads the decoder uses by default: visible cores capped by the cgroup CPU quota */
uint32_t kd_host_threads(void) { return hw_threads(); }
int kd_host_inflate(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_len) {
    return kdz::inflate_raw(in, (size_t)in_len, out, (size_t)out_len) ? KD_OK : KD_E_IO;
}
uint32_t kd_host_crc32(const uint8_t *data, uint64_t len) { return kdz::crc32_buf(data, (size_t)len); }

// ---- chunked reading (kd_stream_*) ----
struct kd_stream {
    Stream st;
    File slot[2];
    int cur = 0;
    std::string err;
    std::vector<uint32_t> cmap;     // kd_stream_set_contig_map: file refID -> the caller's contig id (empty: identity)
};

int kd_stream_open(kd_stream **out, const char *path, int n_threads, uint64_t chunk_bytes) {
    if (!out || !path) return KD_E_ARG;
    *out = nullptr;
    kd_stream *h = new kd_stream();
    int rc = h->st.open(path, n_threads, (size_t)chunk_bytes);
    if (!rc) rc = h->st.preload_header();
    if (rc) { delete h; return rc; }
    *out = h;
    return KD_OK;
}
uint32_t kd_stream_n_contigs(const kd_stream *s) { return s ? (uint32_t)s->st.names.size() : 0; }
const char *kd_stream_contig_name(const kd_stream *s, uint32_t i) { return (s && i < s->st.names.size()) ? s->st.names[i].c_str() : nullptr; }
uint32_t kd_stream_contig_len(const kd_stream *s, uint32_t i) { return (s && i < s->st.lens.size()) ? s->st.lens[i] : 0; }
uint64_t kd_stream_n_records(const kd_stream *s) { return s ? s->st.n_records : 0; }
int kd_stream_next(kd_stream *s, const kd_batch **batch) {
    if (!s || !batch) return KD_E_ARG;
    *batch = nullptr;
    File &f = s->slot[s->cur];
    s->cur ^= 1;
    std::vector<std::string> names = s->st.names;     // the header of a BGZF stream is re-read with the first chunk
    std::vector<uint32_t> lens = s->st.lens;
    bool got = false;
    const int rc = s->st.next(f, &got);
    if (s->st.names.size() != names.size()) { s->st.names = names; s->st.lens = lens; }
    if (rc) { s->err = g_decode_error; return rc; }
    if (!got) return KD_OK;
    if (!s->cmap.empty()) {
        size_t w = 0;      // records kept so far (0xfffffffe: the records of this @SQ entry are dropped -- another pass's contigs)
        for (size_t i = 0; i < f.contig.size(); i++) {
            const uint32_t c = f.contig[i];
            const uint32_t m = c < s->cmap.size() ? s->cmap[c] : 0xffffffffu;
            if (m == 0xffffffffu) {
                s->err = "kd_stream_set_contig_map: a record lies on @SQ entry " + std::to_string(c) + ", which the map leaves out";
                return KD_E_ARG;
            }
            if (m == 0xfffffffeu) continue;
            f.contig[w] = m;
            if (w != i) {      // (the payload stays where it is: the offsets say where)
                f.pos0[w] = f.pos0[i]; f.flag[w] = f.flag[i]; f.seq_off[w] = f.seq_off[i]; f.seq_len[w] = f.seq_len[i];
                f.cig_off[w] = f.cig_off[i]; f.n_cig[w] = f.n_cig[i];
            }
            w++;
        }
        if (w != f.contig.size()) {
            f.contig.resize(w); f.pos0.resize(w); f.flag.resize(w); f.seq_off.resize(w); f.seq_len.resize(w); f.cig_off.resize(w); f.n_cig.resize(w);
        }
    }
    finish_view(f);
    *batch = &f.view;
    return KD_OK;
}
int kd_stream_set_contig_map(kd_stream *s, const uint32_t *map, uint32_t n) {
    if (!s || (n && !map)) return KD_E_ARG;
    if (n && n != s->st.names.size()) { s->err = "kd_stream_set_contig_map: the map must have one entry per @SQ line"; return KD_E_ARG; }
    s->cmap.assign(map, map + n);
    return KD_OK;
}
const char *kd_stream_last_error(const kd_stream *s) { return s ? s->err.c_str() : g_decode_error.c_str(); }
void kd_stream_close(kd_stream *s) { delete s; }
const kd_batch *kd_decode_batch(const kd_file *f) { return f ? &f->f.view : nullptr; }
uint32_t kd_decode_n_contigs(const kd_file *f) { return f ? (uint32_t)f->f.names.size() : 0; }
const char *kd_decode_contig_name(const kd_file *f, uint32_t i) { return (f && i < f->f.names.size()) ? f->f.names[i].c_str() : nullptr; }
uint32_t kd_decode_contig_len(const kd_file *f, uint32_t i) { return (f && i < f->f.lens.size()) ? f->f.lens[i] : 0; }
uint64_t kd_decode_n_records(const kd_file *f) { return f ? f->f.n_records : 0; }
void kd_decode_close(kd_file *f) { delete f; }

}  // extern "C"
