// kindel_hip.hip -- the product: libkindel_hip.so for AMD Instinct MI355X (gfx950, CDNA4).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC kindel_hip.hip kd_decode.cpp -lz
// HIP runtime policy for KdEngine + the C-ABI of include/kindel_hip.h.  There is no CPU
// fallback: without a usable GPU kd_create() fails with KD_E_HIP.
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <thread>
#include <vector>

#include "kd_engine.h"

struct HipRt {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int dev = 0, cus = 256;
    std::string e;
    int prof = 0;   // 0 off, 1 every launch, 2 only the launches named prof_only
    struct Pending { std::string name; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::map<std::string, std::pair<uint64_t, double>> rows;

    bool bad(hipError_t rc) {
        if (rc == hipSuccess) return false;
        e = hipGetErrorString(rc);
        (void)hipGetLastError();  // clear the sticky per-thread error so it cannot leak into the next call
        return true;
    }
    const char *err() const { return e.c_str(); }

    int trace = 0;          // KD_LAUNCH_TRACE (knob, read when the context is created: fault localisation -- 1: name, geometry, then wait for the kernel; 2: print only)
    int init(int device, void *s) {
        int n = 0;
        trace = getenv("KD_LAUNCH_TRACE") ? std::max(1, atoi(getenv("KD_LAUNCH_TRACE"))) : 0;
        if (bad(hipGetDeviceCount(&n))) return 1;
        if (device < 0 || device >= n) { e = "no such HIP device (" + std::to_string(n) + " visible)"; return 1; }
        dev = device;
        if (bad(hipSetDevice(dev))) return 1;
        hipDeviceProp_t p;
        if (bad(hipGetDeviceProperties(&p, dev))) return 1;
        cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
        if (s) { stream = (hipStream_t)s; own_stream = false; }
        else { if (bad(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking))) return 1; own_stream = true; }
        return 0;
    }
    void shutdown() {
        profile_reset();
        graph_drop();
        if (pin) { (void)hipHostFree(pin); pin = nullptr; }
        if (stage_buf) { (void)hipHostFree(stage_buf); stage_buf = nullptr; stage_cap = 0; }
        if (graph_pin) { (void)hipHostFree(graph_pin); graph_pin = nullptr; graph_pin_cap = 0; }
        for (int k = 0; k < 2; k++) { if (up_pin[k]) { (void)hipHostFree(up_pin[k]); up_pin[k] = nullptr; } if (up_ev[k]) { (void)hipEventDestroy(up_ev[k]); up_ev[k] = nullptr; } }
        if (copy_stream) { (void)hipStreamDestroy(copy_stream); copy_stream = nullptr; }
        if (ev_copy) { (void)hipEventDestroy(ev_copy); ev_copy = nullptr; }
        if (own_stream && stream) { (void)hipSetDevice(dev); (void)hipStreamDestroy(stream); }
        stream = nullptr;
    }
    int n_cus() const { return cus; }
    size_t free_bytes() { size_t f = 0, t = 0; if (bad(hipSetDevice(dev)) || bad(hipMemGetInfo(&f, &t))) return 0; return f; }
    void *alloc(size_t bytes) {
        void *p = nullptr;
        if (bad(hipSetDevice(dev))) return nullptr;
        if (bad(hipMalloc(&p, bytes ? bytes : 1))) return nullptr;
        return p;
    }
    void free(void *p) { (void)hipSetDevice(dev); (void)hipFree(p); }
    int memset(void *p, int v, size_t n) { return n ? bad(hipMemsetAsync(p, v, n, stream)) : 0; }
    int memset2d(void *p, size_t pitch, int v, size_t width, size_t height) {
        return (width && height) ? bad(hipMemset2DAsync(p, pitch, v, width, height, stream)) : 0;
    }
    int h2d(void *d, const void *h, size_t n) { return n ? bad(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, stream)) : 0; }
    int d2d(void *d, const void *s, size_t n) { return n ? bad(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, stream)) : 0; }
    int d2h(void *h, const void *d, size_t n) {
        if (n && bad(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, stream))) return 1;
        return bad(hipStreamSynchronize(stream));
    }
    int sync() { return bad(hipStreamSynchronize(stream)); }
    int d2h_async(void *h, const void *d, size_t n) { return n ? bad(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, stream)) : 0; }
    // ---- one-launch step (kd_step): the dispatch chain of a step captured into a hipGraph and replayed ----
    hipGraphExec_t graph_exec = nullptr;
    bool capturing = false;
    bool graph_supported() const { return true; }
    bool has_graph() const { return graph_exec != nullptr; }
    void graph_drop() { if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; } }
    int capture_begin() {
        graph_drop();
        if (bad(hipSetDevice(dev)) || bad(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed))) return 1;
        capturing = true;
        return 0;
    }
    int capture_end(bool keep) {
        hipGraph_t g = nullptr;
        capturing = false;
        if (bad(hipStreamEndCapture(stream, &g)) || !g) return 1;
        int rc = 0;
        if (keep && bad(hipGraphInstantiate(&graph_exec, g, nullptr, nullptr, 0))) { graph_exec = nullptr; rc = 1; }
        (void)hipGraphDestroy(g);
        return rc;
    }
    int graph_launch() { return bad(hipSetDevice(dev)) || bad(hipGraphLaunch(graph_exec, stream)); }
    // ---- a large host buffer (pageable: a mapped file) -> device, PIPELINED: worker threads copy 32 MiB pieces into two pinned
    // buffers, a copy stream moves them on (hipMemcpyAsync from pageable memory stages through one thread of the runtime: 8 GB/s
    // measured on the GPU node, 40 % of the device-side ingest), and after every piece `after(bytes_there)` may launch work on the
    // compute stream that needs only the bytes [0, bytes_there): it is ordered behind that piece's copy by an event.
    hipStream_t copy_stream = nullptr;
    void *up_pin[2] = {nullptr, nullptr};
    hipEvent_t up_ev[2] = {nullptr, nullptr};
    static constexpr size_t UP_CHUNK = (size_t)32 << 20;
    template <class F>
    int upload(void *dst, const uint8_t *src, size_t n, F &&after) {
        if (bad(hipSetDevice(dev))) return 1;
        if (!copy_stream && bad(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking))) return 1;
        for (int k = 0; k < 2; k++) {
            if (!up_pin[k] && bad(hipHostMalloc(&up_pin[k], UP_CHUNK, hipHostMallocDefault))) return 1;
            if (!up_ev[k] && bad(hipEventCreateWithFlags(&up_ev[k], hipEventDisableTiming))) return 1;
        }
        unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        if (const char *e = getenv("KD_UPLOAD_THREADS")) nt = (unsigned)std::max(1, atoi(e));
        size_t c = 0;
        for (size_t o = 0; o < n; o += UP_CHUNK, c++) {
            const size_t len = std::min(UP_CHUNK, n - o);
            const int k = (int)(c & 1);
            if (c >= 2 && bad(hipEventSynchronize(up_ev[k]))) return 1;      // the copy that read this buffer two pieces ago is done
            const size_t per = (len / nt + 4095) & ~(size_t)4095;
            std::vector<std::thread> th;
            for (unsigned t = 1; t < nt && (size_t)t * per < len; t++)
                th.emplace_back([=] { memcpy((uint8_t *)up_pin[k] + (size_t)t * per, src + o + (size_t)t * per, std::min(per, len - (size_t)t * per)); });
            memcpy(up_pin[k], src + o, std::min(per, len));
            for (auto &x : th) x.join();
            if (bad(hipMemcpyAsync((uint8_t *)dst + o, up_pin[k], len, hipMemcpyHostToDevice, copy_stream)) || bad(hipEventRecord(up_ev[k], copy_stream)) ||
                bad(hipStreamWaitEvent(stream, up_ev[k], 0)))
                return 1;
            if (after(o + len)) return 1;
        }
        return 0;
    }

    // a pinned host buffer for a step's closing round trip (status words + run metadata), grown on demand
    void *stage_buf = nullptr;
    size_t stage_cap = 0;
    void *stage(size_t bytes) {
        if (bytes <= stage_cap) return stage_buf;
        if (bad(hipSetDevice(dev))) return nullptr;
        if (stage_buf) { (void)hipStreamSynchronize(stream); (void)hipHostFree(stage_buf); stage_buf = nullptr; stage_cap = 0; }
        const size_t n = (bytes + 65535) & ~(size_t)65535;
        if (bad(hipHostMalloc(&stage_buf, n, hipHostMallocDefault))) { stage_buf = nullptr; return nullptr; }
        stage_cap = n;
        return stage_buf;
    }

    // a second pinned buffer, for what a captured graph copies back on every replay (the run's metadata): a memcpy node whose
    // destination is PAGEABLE host memory goes through the runtime's staging, which other users of the runtime (a torch .cpu()
    // between two replays) can pull from under it -- HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION on the next launch, seen on MI355X
    void *graph_pin = nullptr;
    size_t graph_pin_cap = 0;
    void *graph_stage(size_t bytes) {
        if (bytes <= graph_pin_cap) return graph_pin;
        if (bad(hipSetDevice(dev))) return nullptr;
        if (graph_pin) { (void)hipStreamSynchronize(stream); (void)hipHostFree(graph_pin); graph_pin = nullptr; graph_pin_cap = 0; }
        const size_t n = (bytes + 65535) & ~(size_t)65535;
        if (bad(hipHostMalloc(&graph_pin, n, hipHostMallocDefault))) { graph_pin = nullptr; return nullptr; }
        graph_pin_cap = n;
        return graph_pin;
    }

    // is p page-locked host memory the runtime knows (hipHostMalloc / hipHostRegister; torch's pin_memory)?  A graph's copy node
    // may only point at such memory (above).
    bool is_pinned(const void *p) {
        if (!p) return true;
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    }

    // small readbacks (the status words, one cache line each: 4 KiB, 5 KiB in a profiling build) go through a pinned bounce buffer:
    // no pageable staging in the runtime
    static constexpr size_t KD_SMALL_COPY = KDS_COUNT * 8 > 16384 ? (size_t)KDS_COUNT * 8 : 16384;      // (the status words are the largest of the small read-backs, whatever their spacing)
    void *pin = nullptr;
    int d2h_small(void *h, const void *d, size_t n) {
        if (!pin && bad(hipHostMalloc(&pin, KD_SMALL_COPY, hipHostMallocDefault))) return 1;
        if (n > KD_SMALL_COPY) return d2h(h, d, n);
        if (bad(hipMemcpyAsync(pin, d, n, hipMemcpyDeviceToHost, stream))) return 1;
        if (bad(hipStreamSynchronize(stream))) return 1;
        memcpy(h, pin, n);
        return 0;
    }

    // the same read-back in two halves: work queued between begin and end runs while the host waits for the copy
    hipEvent_t ev_copy = nullptr;
    int d2h_small_begin(const void *d, size_t n) {
        if (n > KD_SMALL_COPY) return 1;
        if (!pin && bad(hipHostMalloc(&pin, KD_SMALL_COPY, hipHostMallocDefault))) return 1;
        if (!ev_copy && bad(hipEventCreateWithFlags(&ev_copy, hipEventDisableTiming))) return 1;
        return bad(hipMemcpyAsync(pin, d, n, hipMemcpyDeviceToHost, stream)) || bad(hipEventRecord(ev_copy, stream));
    }
    int d2h_small_end(void *h, size_t n) {
        if (bad(hipEventSynchronize(ev_copy))) return 1;
        memcpy(h, pin, n);
        return 0;
    }

    template <class K, class... A>
    int launch(const char *name, K k, unsigned grid, unsigned block, size_t shmem, A... args) {
        if (bad(hipSetDevice(dev))) return 1;
        if (shmem > 48 * 1024 && !capturing &&
            bad(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)))
            return 1;
        Pending p;
        const bool timed = !capturing && (prof == 1 || (prof == 2 && (!strcmp(name, "k_window") || !strcmp(name, "k_strip"))));
        if (timed) {
            if (bad(hipEventCreate(&p.a)) || bad(hipEventCreate(&p.b))) return 1;
            if (bad(hipEventRecord(p.a, stream))) return 1;
        }
        if (trace && !capturing) fprintf(stderr, "[kd] %s grid %u block %u lds %zu\n", name, grid, block, shmem);
        k<<<dim3(grid), dim3(block), shmem, stream>>>(args...);
        if (bad(hipGetLastError())) return 1;
        if (trace == 1 && !capturing && bad(hipStreamSynchronize(stream))) return 1;
        if (timed) {
            if (bad(hipEventRecord(p.b, stream))) return 1;
            p.name = name;
            pending.push_back(p);
        }
        return 0;
    }

    void profile_enable(int mode) { prof = mode; }
    int drain() {
        if (pending.empty()) return 0;
        if (bad(hipStreamSynchronize(stream))) return 1;
        for (auto &p : pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                auto &r = rows[p.name];
                r.first++; r.second += ms;
            }
            (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b);
        }
        pending.clear();
        return 0;
    }
    int profile_get(uint32_t *n_rows, char *names, uint64_t *launches, double *ms) {
        if (drain()) return 1;
        if (names) {
            uint32_t i = 0;
            for (auto &kv : rows) {
                if (i >= *n_rows) break;
                snprintf(names + (size_t)i * 64, 64, "%s", kv.first.c_str());
                launches[i] = kv.second.first; ms[i] = kv.second.second;
                i++;
            }
        }
        *n_rows = (uint32_t)rows.size();
        return 0;
    }
    void profile_reset() { (void)drain(); rows.clear(); }
};

#define KD_RT HipRt
#include "kd_abi.inl"
