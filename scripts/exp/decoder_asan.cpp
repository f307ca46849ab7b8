// EXPERIMENT RECORD / TEST TOOL (round 5): every decoder entry point (whole file, stream, span, BGZF plan) over a directory of
// mutated BAM files (scripts/exp/bam_mutations.py) under AddressSanitizer + UBSan:
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++20 -pthread -Iinclude scripts/exp/decoder_asan.cpp kindel_amd/csrc/kd_decode.cpp -lz -o /tmp/dec
//   python scripts/exp/bam_mutations.py 1500 1 /tmp/mut SOME.bam && ASAN_OPTIONS=detect_leaks=0 /tmp/dec /tmp/mut
// 1 500 files: 685 decoded, 815 refused, no report.
#include "kindel_hip.h"
#include <cstdio>
#include <cstdlib>
#include <dirent.h>
#include <string>
#include <vector>
#include <algorithm>
int main(int argc, char **argv) {
    std::vector<std::string> files; DIR *d = opendir(argv[1]); while (dirent *e = readdir(d)) if (e->d_name[0] == 'm') files.push_back(std::string(argv[1]) + "/" + e->d_name); closedir(d);
    std::sort(files.begin(), files.end());
    int ok = 0, bad = 0, k = 0;
    for (auto &p : files) {
        k++;
        kd_file *h = nullptr;
        if (kd_decode_open(&h, p.c_str(), 1 + k % 5) == 0) { ok++; kd_decode_close(h); } else bad++;
        kd_stream *s = nullptr;
        if (kd_stream_open(&s, p.c_str(), 1 + k % 4, 30000 + 7919 * (k % 13)) == 0) { for (;;) { const kd_batch *b = nullptr; if (kd_stream_next(s, &b) || !b) break; } kd_stream_close(s); }
        uint64_t nb = 0;
        if (kd_bgzf_index(p.c_str(), &nb, nullptr, 0) == 0 && nb > 2) { uint64_t info[4]; kd_file *g = nullptr; if (kd_decode_open_span(&g, p.c_str(), 2, nb / 3, 2 * nb / 3, info) == 0) kd_decode_close(g); }
        kd_bgzf_plan *pl = nullptr; if (kd_bgzf_plan_open(&pl, p.c_str()) == 0) kd_bgzf_plan_close(pl);
    }
    printf("files %zu: decoded %d, refused %d\n", files.size(), ok, bad);
}
