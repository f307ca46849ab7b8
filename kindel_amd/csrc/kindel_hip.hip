// kindel_hip.hip -- the product: libkindel_hip.so for AMD Instinct MI355X (gfx950, CDNA4).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC kindel_hip.hip kd_decode.cpp -lz
// HIP runtime policy for KdEngine + the C-ABI of include/kindel_hip.h.  There is no CPU
// fallback: without a usable GPU kd_create() fails with KD_E_HIP.
#include <hip/hip_runtime.h>

#include <csignal>
#include <map>
#include <atomic>
#include <chrono>
#include <mutex>
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
        guard = getenv("KD_GUARD") ? atoi(getenv("KD_GUARD")) : 0;
        if (guard) {
            static std::once_flag once;
            std::call_once(once, [] { signal(SIGABRT, guard_dump); fprintf(stderr, "[kd guard] on: fenced device allocations, a wait behind every launch\n"); });
        }
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
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        ev_pool.clear();
        if (pin) { (void)hipHostFree(pin); pin = nullptr; }
        if (stage_buf) { (void)hipHostFree(stage_buf); stage_buf = nullptr; stage_cap = 0; }
        if (up_ring) { (void)hipHostFree(up_ring); up_ring = nullptr; }
        for (int k = 0; k < UP_SLOTS; k++) if (up_ev[k]) { (void)hipEventDestroy(up_ev[k]); up_ev[k] = nullptr; }
        if (copy_stream) { (void)hipStreamDestroy(copy_stream); copy_stream = nullptr; }
        for (int i = 0; i < N_SIDE; i++) { if (side[i]) { (void)hipStreamDestroy(side[i]); side[i] = nullptr; } if (side_ev[i]) { (void)hipEventDestroy(side_ev[i]); side_ev[i] = nullptr; } }
        if (fork_ev) { (void)hipEventDestroy(fork_ev); fork_ev = nullptr; }
        if (ev_copy) { (void)hipEventDestroy(ev_copy); ev_copy = nullptr; }
        if (own_stream && stream) { (void)hipSetDevice(dev); (void)hipStreamDestroy(stream); }
        stream = nullptr;
    }
    int n_cus() const { return cus; }
    size_t free_bytes() { size_t f = 0, t = 0; if (bad(hipSetDevice(dev)) || bad(hipMemGetInfo(&f, &t))) return 0; return f; }
    // ---- KD_GUARD: a device-side "electric fence" (debug knob, round 6; off in the product's default run) ----
    // KD_GUARD=1: every device allocation is mapped by itself (hipMemAddressReserve / hipMemCreate / hipMemMap) so that its LAST
    // byte (16-byte granularity) is the last mapped byte of its address range: the first access past its end -- read or write --
    // is a GPU memory fault at once, whatever the neighbouring allocations are, at an address that names the buffer.  KD_GUARD=2:
    // the allocation's FIRST byte is the first mapped byte instead (underruns, negative offsets).  The slack at the other end is
    // filled with a canary that kd_free checks.  Every launch is followed by a wait (the last kernel named is the one that
    // faulted), buffers are allocated at their exact sizes (kd_engine.h: ensure), a caller's device batch is copied into fenced
    // buffers of exactly the sizes the header promises (kd_engine.h: push_device), and SIGABRT -- what the runtime raises on a
    // GPU fault -- prints the live allocations of every context in the process with their tags.
    int guard = 0;
    struct GuardRec { void *va; size_t va_bytes; void *map_at; size_t map_bytes; hipMemGenericAllocationHandle_t h; size_t bytes; char tag[24]; int dev; };
    static std::map<uintptr_t, GuardRec> &guard_table() { static std::map<uintptr_t, GuardRec> t; return t; }
    static std::mutex &guard_mutex() { static std::mutex m; return m; }
    static const char *&guard_last_kernel() { static const char *k = "(none)"; return k; }
    static void guard_dump(int) {
        fprintf(stderr, "[kd guard] SIGABRT; last kernel launched: %s; live device allocations:\n", guard_last_kernel());
        for (auto &kv : guard_table())
            fprintf(stderr, "[kd guard]   %-12s [%p, %p) %zu bytes (mapped [%p, %p))\n", kv.second.tag, (void *)kv.first, (void *)(kv.first + kv.second.bytes), kv.second.bytes,
                    kv.second.map_at, (void *)((uintptr_t)kv.second.map_at + kv.second.map_bytes));
        fflush(stderr);
        signal(SIGABRT, SIG_DFL);
        abort();
    }
    static constexpr size_t GUARD_ALIGN = 16, GUARD_CANARY = 4096;
    static constexpr size_t GUARD_BIG = (size_t)256 << 20, GUARD_BIG_CANARY = (size_t)64 << 10;
    void *guard_alloc(size_t bytes, const char *tag) {
        if (bytes > GUARD_BIG) {
            // a mapping of a gigabyte in 4 KiB pieces faulted inside the runtime's own fill on this stack (round 6: the full-size C4 batch,
            // profiles/r06_guard_campaign.txt): the few buffers of that size get hipMalloc + a canary on both sides instead of a fence
            GuardRec r = {};
            r.bytes = bytes; r.dev = dev; r.map_bytes = bytes + 2 * GUARD_BIG_CANARY;
            snprintf(r.tag, sizeof r.tag, "%s", tag ? tag : "");
            if (bad(hipMalloc(&r.map_at, r.map_bytes))) return nullptr;
            if (bad(hipMemset(r.map_at, 0xA5, r.map_bytes)) || bad(hipDeviceSynchronize())) return nullptr;
            void *user = (void *)((uintptr_t)r.map_at + GUARD_BIG_CANARY);
            std::lock_guard<std::mutex> lk(guard_mutex());
            guard_table()[(uintptr_t)user] = r;
            return user;
        }
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
        size_t gran = 0;
        if (bad(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) || !gran) { e = "KD_GUARD: hipMemGetAllocationGranularity: " + e; return nullptr; }
        const size_t need = (bytes + GUARD_ALIGN - 1) / GUARD_ALIGN * GUARD_ALIGN;
        GuardRec r = {};
        r.bytes = bytes; r.dev = dev;
        snprintf(r.tag, sizeof r.tag, "%s", tag ? tag : "");
        r.map_bytes = (need + GUARD_CANARY + gran - 1) / gran * gran;
        r.va_bytes = r.map_bytes + gran;      // one granule of the range stays unmapped: the fence
        if (bad(hipMemAddressReserve(&r.va, r.va_bytes, gran, nullptr, 0))) { e = "KD_GUARD: hipMemAddressReserve: " + e; return nullptr; }
        r.map_at = guard == 2 ? (void *)((uintptr_t)r.va + gran) : r.va;
        hipMemAccessDesc acc = {};
        acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        if (bad(hipMemCreate(&r.h, r.map_bytes, &prop, 0)) || bad(hipMemMap(r.map_at, r.map_bytes, 0, r.h, 0)) || bad(hipMemSetAccess(r.map_at, r.map_bytes, &acc, 1))) {
            e = "KD_GUARD: hipMemCreate / hipMemMap / hipMemSetAccess: " + e;
            return nullptr;
        }
        void *user = guard == 2 ? r.map_at : (void *)((uintptr_t)r.map_at + r.map_bytes - need);
        // the canary: everything mapped that is not the allocation
        if (bad(hipMemset(r.map_at, 0xA5, r.map_bytes)) || bad(hipDeviceSynchronize())) return nullptr;
        std::lock_guard<std::mutex> lk(guard_mutex());
        guard_table()[(uintptr_t)user] = r;
        return user;
    }
    void guard_free(void *p) {
        GuardRec r;
        {
            std::lock_guard<std::mutex> lk(guard_mutex());
            auto it = guard_table().find((uintptr_t)p);
            if (it == guard_table().end()) { fprintf(stderr, "[kd guard] free of an unknown pointer %p\n", p); abort(); }
            r = it->second;
            guard_table().erase(it);
        }
        (void)hipDeviceSynchronize();
        if (!r.va) {      // a big buffer (hipMalloc + canaries on both sides)
            std::vector<unsigned char> c(GUARD_BIG_CANARY);
            for (int side = 0; side < 2; side++) {
                const void *at = side ? (const void *)((uintptr_t)p + r.bytes) : r.map_at;
                if (hipMemcpy(c.data(), at, GUARD_BIG_CANARY, hipMemcpyDeviceToHost) != hipSuccess) continue;
                for (size_t k = 0; k < GUARD_BIG_CANARY; k++)
                    if (c[k] != 0xA5) {
                        fprintf(stderr, "[kd guard] CANARY of %s [%p, +%zu) damaged %s it, at byte %zu (value 0x%02x); last kernel launched: %s\n", r.tag, p, r.bytes,
                                side ? "behind" : "in front of", k, c[k], guard_last_kernel());
                        abort();
                    }
            }
            (void)hipFree(r.map_at);
            return;
        }
        // the canary next to the allocation (GUARD_CANARY bytes of it: the rest of a granule is not worth the copy)
        const size_t need = (r.bytes + GUARD_ALIGN - 1) / GUARD_ALIGN * GUARD_ALIGN;
        std::vector<unsigned char> c(GUARD_CANARY);
        const void *at = guard == 2 ? (const void *)((uintptr_t)p + need) : (const void *)((uintptr_t)p - GUARD_CANARY);
        if (hipMemcpy(c.data(), at, GUARD_CANARY, hipMemcpyDeviceToHost) == hipSuccess) {
            for (size_t k = 0; k < GUARD_CANARY; k++)
                if (c[k] != 0xA5) {
                    fprintf(stderr, "[kd guard] CANARY of %s [%p, +%zu) damaged %s it, at byte %zd (value 0x%02x); last kernel launched: %s\n", r.tag, p, r.bytes,
                            guard == 2 ? "behind" : "in front of", guard == 2 ? (ssize_t)k : (ssize_t)k - (ssize_t)GUARD_CANARY, c[k], guard_last_kernel());
                    abort();
                }
        }
        // The memory goes back, the ADDRESS RANGE does not (no hipMemAddressFree): on this stack (ROCm 7.2, MI355X) a range that is
        // reserved, mapped, unmapped, freed and handed out again shows kernels the wrong pages -- scripts/exp/vmm_tlb_check.hip: stale
        // reads in 299 of 300 rounds with hipMemMap, none with hipMalloc / hipFree (profiles/r06_guard_vmm_address_reuse.txt); KD_GUARD's
        // first version "caught" exactly that.  A range that stays reserved and unmapped also turns every use of a freed buffer into a
        // fault.  (47 bits of address space: a test run reserves a few hundred GB of it.)
        (void)hipMemUnmap(r.map_at, r.map_bytes); (void)hipMemRelease(r.h);
    }
    bool exact_sizes() const { return guard != 0; }
    void *alloc(size_t bytes, const char *tag = "") {
        void *p = nullptr;
        if (bad(hipSetDevice(dev))) return nullptr;
        if (guard) return guard_alloc(bytes ? bytes : 1, tag);
        if (bad(hipMalloc(&p, bytes ? bytes : 1))) return nullptr;
        return p;
    }
    void free(void *p) {
        (void)hipSetDevice(dev);
        if (guard) { guard_free(p); return; }
        (void)hipFree(p);
    }
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
    // ---- a large host buffer (pageable: a mapped file) -> device, PIPELINED: worker threads copy 32 MiB pieces into two pinned
    // buffers, a copy stream moves them on (hipMemcpyAsync from pageable memory stages through one thread of the runtime: 8 GB/s
    // measured on the GPU node, 40 % of the device-side ingest), and after every piece `after(bytes_there)` may launch work on the
    // compute stream that needs only the bytes [0, bytes_there): it is ordered behind that piece's copy by an event.
    hipStream_t copy_stream = nullptr;
    // (round 6: worker threads that live for the whole upload, each copying WHOLE 4 MiB pieces into slots of one pinned ring, the
    // calling thread moving the pieces on in order as they become ready.  Before: two 32 MiB buffers and a fresh set of threads per
    // piece -- a thousand thread creations per 2 GB file, and the link idle while a piece was being filled: 54 - 104 ms for 2 GB.)
    static constexpr int UP_SLOTS = 16;
    static constexpr size_t UP_PIECE = (size_t)4 << 20;
    void *up_ring = nullptr;
    hipEvent_t up_ev[UP_SLOTS] = {};
    template <class F>
    int upload(void *dst, const uint8_t *src, size_t n, F &&after) {
        if (bad(hipSetDevice(dev))) return 1;
        if (!copy_stream && bad(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking))) return 1;
        if (!up_ring && bad(hipHostMalloc(&up_ring, UP_SLOTS * UP_PIECE, hipHostMallocDefault))) return 1;
        for (int k = 0; k < UP_SLOTS; k++)
            if (!up_ev[k] && bad(hipEventCreateWithFlags(&up_ev[k], hipEventDisableTiming))) return 1;
        size_t piece = UP_PIECE;
        if (const char *e = getenv("KD_UPLOAD_CHUNK")) piece = std::min(UP_PIECE, (size_t)std::max(1, atoi(e)));      // (tests: many tiny pieces)
        const size_t n_pieces = (n + piece - 1) / piece;
        const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(UP_SLOTS / 2, kd_host_threads()), n_pieces));
        std::vector<std::atomic<unsigned char>> ready(n_pieces);
        for (auto &r : ready) r.store(0, std::memory_order_relaxed);
        std::atomic<size_t> issued{0};      // pieces whose copy to the device has been queued (their slot's event is recorded)
        std::atomic<bool> stop{false};
        std::vector<std::thread> th;
        for (unsigned w = 0; w < nt; w++)
            th.emplace_back([&, w] {
                for (size_t c = w; c < n_pieces && !stop.load(std::memory_order_relaxed); c += nt) {
                    const int slot = (int)(c % UP_SLOTS);
                    if (c >= (size_t)UP_SLOTS) {      // the slot's last occupant must have left for the device
                        while (issued.load(std::memory_order_acquire) <= c - UP_SLOTS && !stop.load(std::memory_order_relaxed)) std::this_thread::yield();
                        if (stop.load(std::memory_order_relaxed)) return;
                        (void)hipSetDevice(dev);
                        (void)hipEventSynchronize(up_ev[slot]);
                    }
                    const size_t o = c * piece;
                    memcpy((uint8_t *)up_ring + (size_t)slot * UP_PIECE, src + o, std::min(piece, n - o));
                    ready[c].store(1, std::memory_order_release);
                }
            });
        int rc = 0;
        typedef std::chrono::steady_clock clk;
        double t_wait = 0, t_api = 0, t_after = 0;      // (KD_INGEST_TRACE: where the calling thread's time goes)
        for (size_t c = 0; c < n_pieces && !rc; c++) {
            const clk::time_point t0 = clk::now();
            while (!ready[c].load(std::memory_order_acquire)) std::this_thread::yield();
            const clk::time_point t1 = clk::now();
            const int slot = (int)(c % UP_SLOTS);
            const size_t o = c * piece, len = std::min(piece, n - o);
            if (bad(hipMemcpyAsync((uint8_t *)dst + o, (uint8_t *)up_ring + (size_t)slot * UP_PIECE, len, hipMemcpyHostToDevice, copy_stream)) ||
                bad(hipEventRecord(up_ev[slot], copy_stream)) || bad(hipStreamWaitEvent(stream, up_ev[slot], 0)))
                rc = 1;
            last_up = slot;
            issued.store(c + 1, std::memory_order_release);
            const clk::time_point t2 = clk::now();
            if (!rc && after(o + len)) rc = 1;
            const clk::time_point t3 = clk::now();
            t_wait += std::chrono::duration<double, std::milli>(t1 - t0).count(); t_api += std::chrono::duration<double, std::milli>(t2 - t1).count();
            t_after += std::chrono::duration<double, std::milli>(t3 - t2).count();
        }
        stop.store(true);
        for (auto &x : th) x.join();
        if (getenv("KD_INGEST_TRACE"))
            fprintf(stderr, "kd upload: %zu pieces, %u copy threads; calling thread waited %.1f ms for filled pieces, %.1f ms in the copy / event calls, %.1f ms in the launches behind the pieces\n",
                    n_pieces, nt, t_wait, t_api, t_after);
        return rc;
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

    // host memory as the device addresses it: pinned memory the runtime knows (hipHostMalloc / hipHostRegister; torch's pin_memory) -> its
    // device pointer, anything else (pageable) -> NULL
    void *device_view(void *host) {
        if (!host || bad(hipSetDevice(dev))) return nullptr;
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (a.type != hipMemoryTypeHost) return nullptr;
        void *d = nullptr;
        if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return d;
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
        return launch_on(stream, name, k, grid, block, shmem, args...);
    }
    template <class K, class... A>
    int launch_on(hipStream_t st, const char *name, K k, unsigned grid, unsigned block, size_t shmem, A... args) {
        if (bad(hipSetDevice(dev))) return 1;
        if (shmem > 48 * 1024 &&
            bad(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)))
            return 1;
        Pending p;
        const bool timed = (prof == 1 || (prof == 2 && (!strcmp(name, "k_window") || !strcmp(name, "k_strip"))));
        if (timed) {
            if (!(p.a = take_event()) || !(p.b = take_event())) return 1;
            if (bad(hipEventRecord(p.a, st))) return 1;
        }
        if (trace) fprintf(stderr, "[kd] %s grid %u block %u lds %zu\n", name, grid, block, shmem);
        if (guard) guard_last_kernel() = name;
        k<<<dim3(grid), dim3(block), shmem, st>>>(args...);
        if (bad(hipGetLastError())) return 1;
        if ((trace == 1 || guard) && bad(hipStreamSynchronize(st))) return 1;
        if (timed) {
            if (bad(hipEventRecord(p.b, st))) return 1;
            p.name = name;
            pending.push_back(p);
        }
        return 0;
    }
    // ---- side streams (the device-side ingest's inflate groups: each group needs only the file's bytes up to its last block, and a group
    // alone leaves most of the chip idle -- a lane works ~20 ms on its block whatever the group's size): side_fork() orders the side
    // streams behind what the main stream has queued so far, side_after_upload(i) behind the upload piece recorded last, launch_side(i, ...)
    // launches there, side_join() puts the main stream behind all of them ----
    // (eight: a group holds its stream for one block's decode latency + its resolve pass, ~45 ms, and a 2 GB file delivers a group of
    // 8 192 blocks every ~8 ms: with four streams the groups queued behind each other for 100 ms after the last byte had arrived)
    static constexpr int N_SIDE = 8;
    hipStream_t side[N_SIDE] = {};
    hipEvent_t side_ev[N_SIDE] = {}, fork_ev = nullptr;
    int last_up = -1;
    int side_fork() {
        if (bad(hipSetDevice(dev))) return 1;
        if (!fork_ev && bad(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming))) return 1;
        if (bad(hipEventRecord(fork_ev, stream))) return 1;
        for (int i = 0; i < N_SIDE; i++) {
            if (!side[i] && bad(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking))) return 1;
            if (!side_ev[i] && bad(hipEventCreateWithFlags(&side_ev[i], hipEventDisableTiming))) return 1;
            if (bad(hipStreamWaitEvent(side[i], fork_ev, 0))) return 1;
        }
        last_up = -1;
        return 0;
    }
    int side_after_upload(int i) { return last_up >= 0 ? bad(hipStreamWaitEvent(side[i % N_SIDE], up_ev[last_up], 0)) : 0; }
    template <class K, class... A>
    int launch_side(int i, const char *name, K k, unsigned grid, unsigned block, size_t shmem, A... args) {
        return launch_on(side[i % N_SIDE], name, k, grid, block, shmem, args...);
    }
    int memset_side(int i, void *p, int v, size_t n) { return n ? bad(hipMemsetAsync(p, v, n, side[i % N_SIDE])) : 0; }
    int side_join() {
        for (int i = 0; i < N_SIDE; i++)
            if (side[i] && (bad(hipEventRecord(side_ev[i], side[i])) || bad(hipStreamWaitEvent(stream, side_ev[i], 0)))) return 1;
        return 0;
    }

    // timing events are kept and reused: hipEventCreate / hipEventDestroy around every timed launch cost the timed region of bench.py
    // -- which must measure its dominant kernel live -- tens of microseconds per step (round 6: the same steps ran 1.47 ms in a loop
    // without events and 1.55 ms in bench.py's)
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t take_event() {
        if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (bad(hipEventCreate(&e))) return nullptr;
        return e;
    }
    void profile_enable(int mode) {
        prof = mode;
        if (mode) for (int k = (int)ev_pool.size(); k < 64; k++) { hipEvent_t e = nullptr; if (hipEventCreate(&e) == hipSuccess) ev_pool.push_back(e); else break; }
    }
    int drain() {
        if (pending.empty()) return 0;
        if (bad(hipStreamSynchronize(stream))) return 1;
        for (auto &p : pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                auto &r = rows[p.name];
                r.first++; r.second += ms;
            }
            ev_pool.push_back(p.a); ev_pool.push_back(p.b);
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
