// tools/kd_bamwriter.cpp -- TEST / BENCH TOOL, not part of the product library: writes a host batch (kd_batch, host arrays) as a
// BGZF-compressed BAM, record bodies laid out and deflated in parallel -- the synthetic inputs of the end-to-end runs and of the
// decoder / device-side-ingest tests (the product only READS alignment files; until round 3 this function sat in libkindel_hip.so
// and its header).  Built by __graft_entry__.build() into tools/libkindel_tools.so; bound by kindel_amd/_native.py: write_bam().
//   g++ -O2 -std=c++17 -shared -fPIC -pthread tools/kd_bamwriter.cpp -lz -o tools/libkindel_tools.so
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/kindel_hip.h"      // kd_batch, the error codes
#include "kindel_tools.h"

namespace {
std::string g_tools_error;
struct RawBuf {                          // (malloc, not a vector: 4.5 GB for the full C3 file, no zero fill)
    uint8_t *p = nullptr;
    size_t n = 0;
    ~RawBuf() { free(p); }
    bool resize(size_t m) { uint8_t *q = (uint8_t *)realloc(p, m ? m : 1); if (!q) return false; p = q; n = m; return true; }
    uint8_t *data() { return p; }
    size_t size() const { return n; }
};
// visible cores capped by 1.5 x the cgroup CPU quota (the GPU boxes show 256 cores and grant 16)
unsigned hw_threads() {
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    long long quota = -1, period = 100000;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        if (fscanf(f, "%63s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    }
    if (quota > 0 && period > 0) hw = (unsigned)std::min<long long>(hw, std::max<long long>(1, (3 * quota + 2 * period - 1) / (2 * period)));
    return hw;
}
}  // namespace

extern "C" {

const char *kd_tools_last_error(void) { return g_tools_error.c_str(); }

int kd_write_bam(const char *path, const kd_batch *b, uint32_t n_contigs, const char *const *names, const uint32_t *lens,
                 const char *sort_order, int n_threads, int level) {
    if (!path || !b || (n_contigs && (!names || !lens))) return KD_E_ARG;
    const size_t n = (size_t)b->n_reads;
    unsigned nt = n_threads > 0 ? (unsigned)n_threads : hw_threads();
    std::string text = std::string("@HD\tVN:1.6\tSO:") + (sort_order ? sort_order : "unknown") + "\n";
    for (uint32_t c = 0; c < n_contigs; c++) text += std::string("@SQ\tSN:") + names[c] + "\tLN:" + std::to_string(lens[c]) + "\n";
    std::string head("BAM\1", 4);
    auto put32 = [](std::string &o, uint32_t v) { char t[4] = {(char)v, (char)(v >> 8), (char)(v >> 16), (char)(v >> 24)}; o.append(t, 4); };
    put32(head, (uint32_t)text.size()); head += text; put32(head, n_contigs);
    for (uint32_t c = 0; c < n_contigs; c++) { const std::string nm = names[c]; put32(head, (uint32_t)nm.size() + 1); head += nm; head.push_back(0); put32(head, lens[c]); }
    // record sizes -> offsets
    std::vector<uint64_t> off(n + 1);
    auto rec_size = [&](size_t i) -> uint64_t {
        const uint64_t sl = b->seq_len[i], nc = b->n_cig[i];
        const uint64_t cig = nc > 65535 ? 8 : 4 * nc, aux = nc > 65535 ? 8 + 4 * nc : 0;
        return 4 + 32 + 2 + cig + (sl + 1) / 2 + sl + aux;
    };
    {
        const size_t per = (n + nt - 1) / std::max(1u, nt);
        std::vector<uint64_t> part(nt + 1, 0);
        auto w1 = [&](unsigned t) { uint64_t a = 0; for (size_t i = t * per; i < std::min(n, (t + 1) * per); i++) { off[i] = a; a += rec_size(i); } part[t + 1] = a; };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(w1, t);
        w1(0);
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < nt; t++) part[t + 1] += part[t];
        auto w2 = [&](unsigned t) { for (size_t i = t * per; i < std::min(n, (t + 1) * per); i++) off[i] += part[t] + head.size(); };
        std::vector<std::thread> th2;
        for (unsigned t = 1; t < nt; t++) th2.emplace_back(w2, t);
        w2(0);
        for (auto &x : th2) x.join();
        off[n] = part[nt] + head.size();
    }
    RawBuf raw;
    if (!raw.resize((size_t)off[n])) { g_tools_error = "out of memory"; return KD_E_NOMEM; }
    memcpy(raw.data(), head.data(), head.size());
    const char *qual_mode = getenv("KD_WRITE_BAM_QUAL");
    const bool phred = qual_mode && !strcmp(qual_mode, "phred");
    {
        const size_t per = (n + nt - 1) / std::max(1u, nt);
        auto fill = [&](unsigned t) {
            for (size_t i = t * per; i < std::min(n, (t + 1) * per); i++) {
                uint8_t *r = raw.data() + off[i];
                const uint32_t sl = b->seq_len[i], nc = b->n_cig[i];
                const uint32_t *cg = b->cigar + b->cig_off[i];
                auto w32 = [](uint8_t *q, uint32_t v) { q[0] = (uint8_t)v; q[1] = (uint8_t)(v >> 8); q[2] = (uint8_t)(v >> 16); q[3] = (uint8_t)(v >> 24); };
                w32(r, (uint32_t)(off[i + 1] - off[i] - 4));
                w32(r + 4, b->contig[i]); w32(r + 8, (uint32_t)b->pos0[i]);
                r[12] = 2; r[13] = 60; r[14] = 0; r[15] = 0;
                const uint32_t ncw = nc > 65535 ? 2 : nc;
                r[16] = (uint8_t)ncw; r[17] = (uint8_t)(ncw >> 8);
                r[18] = (uint8_t)b->flag[i]; r[19] = (uint8_t)(b->flag[i] >> 8);
                w32(r + 20, sl); w32(r + 24, 0xffffffffu); w32(r + 28, 0xffffffffu); w32(r + 32, 0);
                r[36] = 'r'; r[37] = 0;
                uint8_t *q = r + 38;
                if (nc > 65535) {   // SAMv1 4.2.2: placeholder CIGAR, the real one in the CG:B,I tag
                    uint64_t ref_len = 0;
                    for (uint32_t k = 0; k < nc; k++) { const uint32_t op = cg[k] & 15u; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += cg[k] >> 4; }
                    w32(q, (sl << 4) | 4u); w32(q + 4, ((uint32_t)ref_len << 4) | 3u); q += 8;
                } else {
                    for (uint32_t k = 0; k < nc; k++) w32(q + 4 * k, cg[k]);
                    q += 4 * (size_t)nc;
                }
                memcpy(q, b->seq4 + b->seq_off[i], ((size_t)sl + 1) / 2); q += ((size_t)sl + 1) / 2;
                if (!phred) memset(q, 0xff, sl);     // qualities absent (SAM '*'): compresses to nothing
                else {                                // KD_WRITE_BAM_QUAL=phred: a skewed spread over Phred 2 .. 41, deterministic per (read, base) --
                    for (uint32_t j = 0; j < sl; j++) {   // a file that compresses like sequencer output (3 - 4 x), not 15 x
                        uint32_t hsh = (uint32_t)i * 0x9e3779b1u + j * 0x85ebca6bu;
                        hsh ^= hsh >> 15; hsh *= 0x2c1b3c6du; hsh ^= hsh >> 12; hsh *= 0x297a2d39u; hsh ^= hsh >> 15;
                        const uint32_t a = hsh & 63u, c2 = (hsh >> 6) & 63u;
                        q[j] = (uint8_t)(41u - (a * c2) / 104u);      // most bases near 41, a tail down to ~3
                    }
                }
                q += sl;
                if (nc > 65535) {
                    q[0] = 'C'; q[1] = 'G'; q[2] = 'B'; q[3] = 'I'; w32(q + 4, nc);
                    for (uint32_t k = 0; k < nc; k++) w32(q + 8 + 4 * (size_t)k, cg[k]);
                }
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(fill, t);
        fill(0);
        for (auto &x : th) x.join();
    }
    // BGZF blocks of 0xff00 uncompressed bytes, deflated in parallel, written in order
    const size_t BS = 0xff00, nb = (raw.size() + BS - 1) / BS;
    std::vector<std::vector<uint8_t>> blk(nb + 1);
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    auto deflate_block = [&](const uint8_t *src, size_t len, std::vector<uint8_t> &out) -> bool {
        out.resize(len + len / 8 + 64);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        zs.next_in = const_cast<uint8_t *>(src); zs.avail_in = (uInt)len;
        zs.next_out = out.data() + 18; zs.avail_out = (uInt)(out.size() - 26);
        const int rc = deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) return false;
        const size_t bsize = 18 + clen + 8;
        if (bsize > 65536) return false;
        static const uint8_t hdr[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
        memcpy(out.data(), hdr, 12);
        out[12] = 'B'; out[13] = 'C'; out[14] = 2; out[15] = 0; out[16] = (uint8_t)(bsize - 1); out[17] = (uint8_t)((bsize - 1) >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)len);
        uint8_t *t = out.data() + 18 + clen;
        t[0] = (uint8_t)crc; t[1] = (uint8_t)(crc >> 8); t[2] = (uint8_t)(crc >> 16); t[3] = (uint8_t)(crc >> 24);
        t[4] = (uint8_t)len; t[5] = (uint8_t)(len >> 8); t[6] = (uint8_t)(len >> 16); t[7] = (uint8_t)(len >> 24);
        out.resize(bsize);
        return true;
    };
    auto work = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k > nb) break;
            const size_t o = k * BS, len = k < nb ? std::min(BS, raw.size() - o) : 0;   // block nb: the empty EOF block
            if (!deflate_block(raw.data() + std::min(o, raw.size()), len, blk[k])) ok = false;
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    }
    if (!ok) { g_tools_error = "deflate failed"; return KD_E_IO; }
    FILE *f = fopen(path, "wb");
    if (!f) { g_tools_error = std::string("cannot write ") + path; return KD_E_IO; }
    bool wok = true;
    for (size_t k = 0; k <= nb && wok; k++) wok = fwrite(blk[k].data(), 1, blk[k].size(), f) == blk[k].size();
    wok = fclose(f) == 0 && wok;
    if (!wok) { g_tools_error = std::string("write error on ") + path; return KD_E_IO; }
    return KD_OK;
}

}  // extern "C"
