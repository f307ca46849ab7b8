// scripts/exp/inflate_bench.cpp -- MEASUREMENT (not part of the product): the host inflater (kindel_amd/csrc/kd_inflate.h) alone over
// the BGZF blocks of a BAM file, one core, next to zlib's inflate on the same blocks (every block compared with zlib's bytes first).
// Round 5 on a 2.1 GHz Xeon core, 54 MB of BAM records with Phred-like qualities written at deflate level 1 (18.5 M symbols: 9.7 M
// literal look-ups, 8.8 M matches of 4.8 bytes on average = 78 % of the bytes): 295 - 316 MB/s (clang / gcc; run-to-run noise +-4 %)
// against zlib's 190 - 215.  Tried and measured here, none adopted (all within the noise or slower): the fast loop's cursors as
// locals, primary-table entries carrying TWO literals (16 % of the literal look-ups pair up: a match follows a literal too often),
// BMI2 shifts (+6 % at best), two blocks decoded in one loop for instruction-level parallelism (+3 %), a predicated step that
// treats literals and matches alike (218 MB/s, ~75 micro-ops per symbol).  Without the match copies the loop runs no faster:
// what bounds it is the literal-or-match branch, mispredicted on about every other symbol of such data.
//   g++ -O3 -std=c++17 -march=native scripts/exp/inflate_bench.cpp -lz -o /tmp/inflate_bench && /tmp/inflate_bench file.bam [reps]
#include <zlib.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef KD_INFLATE_HEADER
#define KD_INFLATE_HEADER "../../kindel_amd/csrc/kd_inflate.h"
#endif
#include KD_INFLATE_HEADER

struct Blk { size_t in_off, in_len, out_len; };
int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> raw((size_t)n + 64, 0);
    if (fread(raw.data(), 1, (size_t)n, f) != (size_t)n) return 1;
    fclose(f);
    std::vector<Blk> blocks;
    size_t o = 0, total = 0, max_out = 0;
    while (o + 18 <= (size_t)n) {
        const size_t xlen = raw[o + 10] | raw[o + 11] << 8;
        const size_t bsize = (raw[o + 16] | raw[o + 17] << 8) + 1;
        const size_t isize = raw[o + bsize - 4] | raw[o + bsize - 3] << 8 | raw[o + bsize - 2] << 16 | (size_t)raw[o + bsize - 1] << 24;
        blocks.push_back({o + 12 + xlen, bsize - xlen - 20, isize});
        total += isize; max_out = isize > max_out ? isize : max_out;
        o += bsize;
    }
    std::vector<uint8_t> out(max_out + 64), ref(max_out + 64);
    // correctness against zlib first
    for (const Blk &b : blocks) {
        if (!b.out_len) continue;
        z_stream z{}; inflateInit2(&z, -15);
        z.next_in = raw.data() + b.in_off; z.avail_in = (uInt)b.in_len; z.next_out = ref.data(); z.avail_out = (uInt)b.out_len;
        const int rc = inflate(&z, Z_FINISH); inflateEnd(&z);
        if (rc != Z_STREAM_END) { printf("zlib failed\n"); return 2; }
        if (!kdz::inflate_raw(raw.data() + b.in_off, b.in_len, out.data(), b.out_len) || memcmp(out.data(), ref.data(), b.out_len)) { printf("MISMATCH\n"); return 3; }
    }
    double best = 1e9, bestz = 1e9;
    for (int r = 0; r < reps; r++) {
        auto t0 = std::chrono::steady_clock::now();
        for (const Blk &b : blocks) if (b.out_len && !kdz::inflate_raw(raw.data() + b.in_off, b.in_len, out.data(), b.out_len)) return 4;
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        best = dt < best ? dt : best;
    }
    for (int r = 0; r < 2; r++) {
        auto t0 = std::chrono::steady_clock::now();
        for (const Blk &b : blocks) {
            if (!b.out_len) continue;
            z_stream z{}; inflateInit2(&z, -15);
            z.next_in = raw.data() + b.in_off; z.avail_in = (uInt)b.in_len; z.next_out = ref.data(); z.avail_out = (uInt)b.out_len;
            inflate(&z, Z_FINISH); inflateEnd(&z);
        }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        bestz = dt < bestz ? dt : bestz;
    }
        printf("%zu blocks, %.1f MB inflated from %.1f MB: kdz %.1f MB/s, zlib %.1f MB/s\n", blocks.size(), total / 1e6, n / 1e6, total / 1e6 / best, total / 1e6 / bestz);
    return 0;
}
