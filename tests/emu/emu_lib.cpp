// emu_lib.cpp -- TEST INFRASTRUCTURE ONLY: builds tests/emu/libkindel_emu.so, which exports
// the same C-ABI as libkindel_hip.so but executes kindel_amd/csrc/kd_kernels.h on the CPU
// through tests/emu/hip_emu.h.  Used by `-m "not gpu"` tests to check kernel *logic*
// (indexing, quirks, window planning, hash multiset, scan) against the oracle in a
// container without a GPU.  Never imported by kindel_amd/, never used for measurements.
//   g++ -O1 -std=c++20 -shared -fPIC -pthread emu_lib.cpp ../../kindel_amd/csrc/kd_decode.cpp -lz
#include "hip_emu.h"

#include <string>
#include <vector>

#include "../../kindel_amd/csrc/kd_engine.h"

struct EmuRt {
    std::string e;
    const char *err() const { return e.c_str(); }
    int init(int, void *) { return 0; }
    void shutdown() {}
    int n_cus() const { if (const char *e = getenv("KD_EMU_CUS")) return std::max(1, atoi(e)); return 2; }   // (256 = the launch geometry of an MI355X: idle workgroups, long grids)
    size_t free_bytes() { if (const char *e = getenv("KD_EMU_FREE_BYTES")) return (size_t)strtoull(e, nullptr, 10); return (size_t)1 << 40; }
    // KD_EMU_EXACT=1 (with the AddressSanitizer build, scripts/exp/asan_emu.sh): every "device" allocation has exactly the size asked
    // for and the engine asks for exactly what it needs (kd_engine.h: ensure, push_device) -- the CPU counterpart of the product's
    // KD_GUARD: an access past a buffer's end lands in the sanitizer's red zone instead of in head-room
    bool exact = getenv("KD_EMU_EXACT") != nullptr;
    // KD_EMU_FILL=<byte>: every allocation starts out filled with it -- what hipMalloc hands out in a long-lived process is not zero pages
    // either; a kernel that reads what nobody wrote shows in the results (tests/test_emu_kernels.py: test_poisoned_device_memory_...)
    void *alloc(size_t bytes, const char * = "") {
        // KD_EMU_ALLOC_CAP=<bytes>: a larger single allocation fails, as hipMalloc does for tables that do not fit the GPU (tests of the
        // out-of-memory paths: kindel.bam_to_consensus lays a reference that does not fit out in groups of contigs)
        if (const char *e = getenv("KD_EMU_ALLOC_CAP")) if (bytes > (size_t)strtoull(e, nullptr, 10)) return nullptr;
        const size_t n = exact ? (bytes ? bytes : 1) : ((bytes + 255) & ~size_t(255));
        void *p = exact ? malloc(n) : aligned_alloc(256, n ? n : 256);
        if (p) if (const char *e = getenv("KD_EMU_FILL")) ::memset(p, atoi(e), n);
        return p;
    }
    bool exact_sizes() const { return exact; }
    void free(void *p) { ::free(p); }
    int memset(void *p, int v, size_t n) { ::memset(p, v, n); return 0; }
    int memset2d(void *p, size_t pitch, int v, size_t width, size_t height) {
        for (size_t r = 0; r < height; r++) ::memset((char *)p + r * pitch, v, width);
        return 0;
    }
    int h2d(void *d, const void *h, size_t n) { ::memcpy(d, h, n); return 0; }
    int d2d(void *d, const void *s, size_t n) { ::memcpy(d, s, n); return 0; }
    int d2h(void *h, const void *d, size_t n) { ::memcpy(h, d, n); return 0; }
    int d2h_small(void *h, const void *d, size_t n) { ::memcpy(h, d, n); return 0; }
    unsigned char small[KDS_COUNT * 8 > 16384 ? (size_t)KDS_COUNT * 8 : 16384];
    int d2h_small_begin(const void *d, size_t n) { if (n > sizeof small) return 1; ::memcpy(small, d, n); return 0; }
    int d2h_small_end(void *h, size_t n) { ::memcpy(h, small, n); return 0; }
    template <class F>
    int upload(void *dst, const uint8_t *src, size_t n, F &&after) {     // (pieces as on the GPU, small ones on request: KD_UPLOAD_CHUNK)
        size_t chunk = (size_t)32 << 20;
        if (const char *e = getenv("KD_UPLOAD_CHUNK")) chunk = (size_t)std::max(1, atoi(e));
        for (size_t o = 0; o < n; o += chunk) {
            const size_t len = std::min(chunk, n - o);
            ::memcpy((uint8_t *)dst + o, src + o, len);
            if (after(o + len)) return 1;
        }
        return 0;
    }
    int sync() { return 0; }
    std::vector<unsigned char> stage_buf;
    void *stage(size_t bytes) { if (stage_buf.size() < bytes) stage_buf.resize(bytes); return stage_buf.data(); }
    int d2h_async(void *h, const void *d, size_t n) { ::memcpy(h, d, n); return 0; }
    void *device_view(void *host) { return getenv("KD_EMU_NO_PINNED") ? nullptr : host; }      // (every host buffer is "pinned" here: the zero-copy path of kd_step / kd_finish runs in the CPU tests)
    template <class K, class... A>
    int launch(const char *name, K k, unsigned grid, unsigned block, size_t shmem, A... args) {
        static const bool trace = getenv("KD_EMU_TRACE") != nullptr;
        if (trace) fprintf(stderr, "emu launch %s grid %u block %u\n", name, grid, block);
        emu::launch(k, dim3(grid), dim3(block), shmem, args...);
        return 0;
    }
    // (side streams: everything runs at once, in order, here)
    static constexpr int N_SIDE = 8;
    int side_fork() { return 0; }
    int side_after_upload(int) { return 0; }
    int side_join() { return 0; }
    int memset_side(int, void *p, int v, size_t n) { ::memset(p, v, n); return 0; }
    template <class K, class... A>
    int launch_side(int, const char *name, K k, unsigned grid, unsigned block, size_t shmem, A... args) { return launch(name, k, grid, block, shmem, args...); }
    void profile_enable(int) {}
    int profile_get(uint32_t *n_rows, char *, uint64_t *, double *) { *n_rows = 0; return 0; }
    void profile_reset() {}
};

#define KD_RT EmuRt
#include "../../kindel_amd/csrc/kd_abi.inl"

// ---- test hook: the product's kd_scan_cigar (32-bit, branch-free) next to its 64-bit reference statement (scan_ref.h) ----
#include "scan_ref.h"
// out[0..7] = product: cls, cold, lead, span, n_ins, ins_bases, aligned, walked; out[8..15] = reference
extern "C" void kd_emu_scan_compare(const uint32_t *cigar, uint32_t nc, int32_t pos0, uint32_t sl, uint32_t L, uint64_t *out) {
    uint32_t pre[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 4 && k < nc; k++) pre[k] = cigar[k];
    KdScan a;      // as k_prep does it: the sums-only scan where its premise holds, the exact one otherwise
    const bool inside = kd_scan_cigar_inside(cigar, nc, pos0, sl, L, pre, a);
    const KdScan exact = kd_scan_cigar(cigar, nc, pos0, sl, L, pre);
    if (inside) {  // ... and wherever the premise holds the exact scan must say the same, field by field
        if (exact.cls != a.cls || exact.cold != a.cold || exact.lead != a.lead || exact.span != a.span || exact.n_ins != a.n_ins ||
            exact.ins_bases != a.ins_bases || exact.aligned != a.aligned || exact.walked != a.walked) a.cls = 0xdeadu;
    } else a = exact;
    out[16] = inside ? 1 : 0;
    const KdScanRef b = kd_scan_cigar_ref(cigar, nc, pos0, sl, L);
    out[0] = a.cls; out[1] = a.cold; out[2] = a.lead; out[3] = a.span; out[4] = a.n_ins; out[5] = a.ins_bases; out[6] = a.aligned; out[7] = a.walked;
    out[8] = b.cls; out[9] = b.cold; out[10] = b.lead; out[11] = b.span; out[12] = b.n_ins; out[13] = b.ins_bases; out[14] = b.aligned; out[15] = b.walked;
}
