// EXPERIMENT RECORD / TEST TOOL (round 5):
//   g++ -O1 -g -fsanitize=thread -std=c++20 -pthread -Iinclude scripts/exp/decoder_tsan.cpp kindel_amd/csrc/kd_decode.cpp -lz -o /tmp/drv && /tmp/drv a.bam b.bam
// three files (short reads with 3 kB and 64 kB blocks, long reads), 1 / 3 / 8 / 24 threads: no report, equal digests.
// ThreadSanitizer driver for the host decoder's C-ABI (kd_decode.cpp): whole-file decode, span decode, streaming with the
// consumer reading batch k while the next kd_stream_next decodes batch k+1 (what kd_push_stream does).
#include "kindel_hip.h"
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <cstring>
static uint64_t digest(const kd_batch *b) {
    uint64_t h = 1469598103934665603ULL;
    for (uint64_t i = 0; i < b->n_reads; i++) { h = (h ^ b->contig[i]) * 1099511628211ULL; h = (h ^ (uint32_t)b->pos0[i]) * 1099511628211ULL; h = (h ^ b->seq_len[i]) * 1099511628211ULL; }
    return h;
}
int main(int argc, char **argv) {
    for (int a = 1; a < argc; a++) {
        const char *path = argv[a];
        for (int nt : {1, 3, 8, 24}) {
            kd_file *f = nullptr;
            if (kd_decode_open(&f, path, nt)) { printf("open failed %s: %s\n", path, kd_decode_last_error()); return 1; }
            const kd_batch *b = kd_decode_batch(f);
            uint64_t n = b->n_reads, d = digest(b);
            kd_decode_close(f);
            // spans: the file cut into 5 block ranges, decoded by 5 threads AT ONCE (ranks of one host would be processes; this is harsher)
            uint64_t nb = 0; kd_bgzf_index(path, &nb, nullptr, 0);
            std::vector<uint64_t> cnt(5, 0); std::vector<std::thread> th;
            for (int r = 0; r < 5; r++) th.emplace_back([&, r] {
                kd_file *g = nullptr; uint64_t info[4];
                if (kd_decode_open_span(&g, path, nt, nb * r / 5, nb * (r + 1) / 5, info)) { cnt[r] = ~0ULL; return; }
                cnt[r] = kd_decode_batch(g)->n_reads; kd_decode_close(g); });
            for (auto &t : th) t.join();
            uint64_t tot = 0; for (auto c : cnt) tot += c;
            // stream: consumer thread digests batch k while the main thread asks for batch k+1
            kd_stream *s = nullptr;
            if (kd_stream_open(&s, path, nt, 200000)) { printf("stream open failed\n"); return 1; }
            uint64_t sn = 0; const kd_batch *cur = nullptr; std::thread cons; uint64_t dsum = 0;
            for (;;) {
                const kd_batch *nx = nullptr;
                if (kd_stream_next(s, &nx)) { printf("stream next failed: %s\n", kd_stream_last_error(s)); return 1; }
                if (cons.joinable()) cons.join();
                if (!nx) break;
                sn += nx->n_reads; cur = nx;
                cons = std::thread([cur, &dsum] { dsum += digest(cur); });
            }
            if (cons.joinable()) cons.join();
            kd_stream_close(s);
            printf("%s threads %d: file %llu reads (digest %llx), spans %llu, stream %llu\n", path, nt, (unsigned long long)n, (unsigned long long)d, (unsigned long long)tot, (unsigned long long)sn);
            if (tot != n || sn != n) { printf("MISMATCH\n"); return 2; }
        }
    }
    return 0;
}
