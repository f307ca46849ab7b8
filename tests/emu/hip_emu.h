// hip_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny functional stand-in for the handful of HIP device-side constructs the kindel_amd
// kernels use, so that the *same kernel source* (kindel_amd/csrc/kd_kernels.h) can be
// executed on the build container's CPU by tests/emu/emu_driver.cpp and diffed against the
// oracle.  The build container has no GPU and GPU time is scarce; this lets indexing and
// quirk-handling bugs be found before a kernel ever reaches an MI355X.
//
// It is NOT a backend: nothing in kindel_amd/ includes or links this file, the product
// library is built by hipcc for gfx950 only and fails loudly without a GPU.  Timing,
// memory-model and occupancy behaviour are not modelled -- only functional semantics:
//   * every work-item of a block is a FIBER (its own stack, a hand-written x86-64 context switch) and all fibers of a
//     block run on one OS thread, round robin, switching only at barriers: no OS synchronisation inside a block;
//     blocks are distributed over a few OS threads;
//   * __syncthreads() makes a fiber yield until all fibers of its block have arrived;
//   * __shared__ is a function-local `static thread_local` (one copy per OS thread = per block in flight);
//   * atomics map to GCC __atomic builtins (same arithmetic on LDS and global memory);
//   * wavefront operations (ballot, shuffles, readfirstlane, the wavefront-scope sync the kernels use between
//     lane-private LDS writes and cross-lane LDS reads) go through a barrier over the 64 fibers of a wavefront plus a
//     per-wavefront scratch row: every lane deposits its value, the wavefront meets, every lane reads what it
//     needs, the wavefront meets again.  All 64 lanes of a wavefront must reach such a call together (the kernels
//     only use them in wave-uniform control flow -- on the GPU a diverged ballot would be a different mask, not a hang).
// Kernels must not return before a later __syncthreads() (good HIP style anyway).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define KD_EMU 1

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_idx { unsigned x = 0, y = 0, z = 0; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

inline thread_local emu_idx threadIdx, blockIdx;
inline thread_local dim3 blockDim, gridDim;
inline thread_local unsigned char *emu_dyn_shared_ptr = nullptr;

// ---- fibers: one per work-item of the block this OS thread is executing ----
extern "C" void emu_switch(void **save_sp, void *new_sp);   // saves the callee-saved registers, swaps stacks
asm(R"(
    .text
    .globl emu_switch
    .type emu_switch, @function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch, .-emu_switch
)");

struct emu_block_ctx {
    unsigned nt = 0, cur = 0;
    void *sched_sp = nullptr;
    std::vector<void *> sp;
    std::vector<unsigned char> done;
    unsigned blk_arrived = 0, blk_gen = 0;
    unsigned wave_arrived[16] = {0}, wave_gen[16] = {0};
    unsigned long long wscratch[16][64];
    unsigned long long progress = 0;
    const std::function<void()> *body = nullptr;
};
inline thread_local emu_block_ctx *emu_ctx = nullptr;
static inline void emu_yield() {
    emu_block_ctx *c = emu_ctx;
    emu_switch(&c->sp[c->cur], c->sched_sp);
}
static void emu_trampoline() {
    emu_block_ctx *c = emu_ctx;
    (*c->body)();
    c = emu_ctx;
    c->done[c->cur] = 1; c->progress++;
    for (;;) emu_yield();
}
static inline void emu_wave_meet() {
    emu_block_ctx *c = emu_ctx;
    const unsigned w = c->cur / 64, lanes = std::min(64u, c->nt - 64 * w), my = c->wave_gen[w];
    if (++c->wave_arrived[w] == lanes) { c->wave_arrived[w] = 0; c->wave_gen[w]++; c->progress++; return; }
    while (emu_ctx->wave_gen[w] == my) emu_yield();
}

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ static __attribute__((noinline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define KD_MUL24(a, b) ((uint32_t)(a) * (uint32_t)(b))
#define KD_MUL24S(a, b) ((int32_t)(a) * (int32_t)(b))
#define KD_DYN_SHARED(type, name) type *name = reinterpret_cast<type *>(emu_dyn_shared_ptr)

static inline void __syncthreads() {
    emu_block_ctx *c = emu_ctx;
    const unsigned my = c->blk_gen;
    if (++c->blk_arrived == c->nt) { c->blk_arrived = 0; c->blk_gen++; c->progress++; return; }
    while (emu_ctx->blk_gen == my) emu_yield();
}

// ---- wavefront operations (kd_common.h: KD_WAVE_SYNC, kd_ballot, kd_shfl*, kd_readfirstlane) ----
#define KD_WAVE_SYNC() emu_wave_meet()
static inline unsigned long long emu_exchange(unsigned long long v, unsigned src_lane) {
    unsigned long long *scratch = emu_ctx->wscratch[threadIdx.x / 64];
    scratch[threadIdx.x & 63u] = v;
    emu_wave_meet();
    const unsigned long long r = scratch[src_lane & 63u];
    emu_wave_meet();
    return r;
}
static inline unsigned long long kd_ballot(bool pred) {
    unsigned long long *scratch = emu_ctx->wscratch[threadIdx.x / 64];
    scratch[threadIdx.x & 63u] = pred ? 1ULL : 0ULL;
    emu_wave_meet();
    unsigned long long m = 0;
    for (unsigned l = 0; l < 64; l++) m |= (scratch[l] & 1ULL) << l;
    emu_wave_meet();
    return m;
}
static inline uint32_t kd_lane_id() { return threadIdx.x & 63u; }
static inline uint32_t kd_mbcnt(unsigned long long mask) { return (uint32_t)__builtin_popcountll(mask & ((1ULL << (threadIdx.x & 63u)) - 1ULL)); }
static inline uint32_t kd_shfl(uint32_t v, unsigned src_lane) { return (uint32_t)emu_exchange(v, src_lane); }
static inline unsigned long long kd_shfl64(unsigned long long v, unsigned src_lane) { return emu_exchange(v, src_lane); }
static inline uint32_t kd_shfl_up(uint32_t v, unsigned d) {   // lanes below d keep their own value (HIP __shfl_up)
    const unsigned l = threadIdx.x & 63u;
    return (uint32_t)emu_exchange(v, l >= d ? l - d : l);
}
static inline unsigned long long kd_shfl_up64(unsigned long long v, unsigned d) {
    const unsigned l = threadIdx.x & 63u;
    return emu_exchange(v, l >= d ? l - d : l);
}
static inline uint32_t kd_shfl_xor(uint32_t v, unsigned m) { return (uint32_t)emu_exchange(v, (threadIdx.x & 63u) ^ m); }
static inline uint32_t kd_readfirstlane(uint32_t v) { return (uint32_t)emu_exchange(v, 0); }
static inline unsigned long long kd_readfirstlane64(unsigned long long v) { return emu_exchange(v, 0); }
static inline uint32_t kd_readlane(uint32_t v, unsigned src_lane) { return (uint32_t)emu_exchange(v, src_lane); }
static inline int kd_popcll(unsigned long long m) { return __builtin_popcountll(m); }
static inline uint32_t kd_wave_or(uint32_t v) {
    unsigned long long *scratch = emu_ctx->wscratch[threadIdx.x / 64];
    scratch[threadIdx.x & 63u] = v;
    emu_wave_meet();
    uint32_t r = 0;
    for (unsigned l = 0; l < 64; l++) r |= (uint32_t)scratch[l];
    emu_wave_meet();
    return r;
}
static inline uint32_t kd_wave_scan_add(uint32_t v) {   // inclusive prefix sum over the wavefront's lanes
    unsigned long long *scratch = emu_ctx->wscratch[threadIdx.x / 64];
    scratch[threadIdx.x & 63u] = v;
    emu_wave_meet();
    uint32_t r = 0;
    for (unsigned l = 0; l <= (threadIdx.x & 63u); l++) r += (uint32_t)scratch[l];
    emu_wave_meet();
    return r;
}
static inline uint32_t kd_wave_scan_max(uint32_t v) {   // running maximum over the wavefront's lanes (inclusive)
    unsigned long long *scratch = emu_ctx->wscratch[threadIdx.x / 64];
    scratch[threadIdx.x & 63u] = v;
    emu_wave_meet();
    uint32_t r = 0;
    for (unsigned l = 0; l <= (threadIdx.x & 63u); l++) r = std::max(r, (uint32_t)scratch[l]);
    emu_wave_meet();
    return r;
}
// v_perm_b32: byte i of the result = byte sel.byte[i] of {hi (bytes 4-7), lo (bytes 0-3)} for selectors 0-7; selector 0x0c = the
// constant 0x00, 0x0d and above = 0xff (the sign-replicating selectors 8-11 are not used by the kernels)
static inline uint32_t kd_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const unsigned long long src = ((unsigned long long)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t sb = (sel >> (8 * i)) & 0xffu;
        const uint32_t byte = sb < 8u ? (uint32_t)((src >> (8 * sb)) & 0xffu) : sb == 0x0cu ? 0u : sb > 0x0cu ? 0xffu : (abort(), 0u);
        r |= byte << (8 * i);
    }
    return r;
}
static inline uint32_t kd_opaque(uint32_t x) { return x; }
// v_alignbyte_b32: ({hi, lo} >> 8 * (sh & 3)) [31:0]
static inline uint32_t kd_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    return (uint32_t)(((((unsigned long long)hi) << 32) | lo) >> (8 * (sh & 3u)));
}

// ---- atomics (device-scope on the GPU; sequentially consistent here) ----
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) {
    return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
}
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline uint32_t atomicMax(uint32_t *p, uint32_t v) {
    uint32_t old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
static inline uint32_t atomicMin(uint32_t *p, uint32_t v) {
    uint32_t old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
static inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) {
    unsigned long long old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
static inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) {
    unsigned long long old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
static inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long cmp, unsigned long long v) {
    __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return cmp;
}
static inline uint32_t atomicCAS(uint32_t *p, uint32_t cmp, uint32_t v) {
    __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return cmp;
}

static inline unsigned long long kd_ld_acquire(const unsigned long long *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static inline void kd_spin_pause() { std::this_thread::yield(); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

namespace emu {

constexpr size_t kStack = 64 * 1024;   // per fiber

// Run kernel(args...) over grid x block (1-D): a few OS threads take blocks from a counter; the work-items of a block
// are fibers on that thread.
template <class K, class... A>
void launch(K kernel, dim3 grid, dim3 block, size_t dyn_shared, A... args) {
    const unsigned nt = block.x;
    if (nt > 1024) { fprintf(stderr, "hip_emu: block too large\n"); abort(); }
    const std::function<void()> body = [&]() { kernel(args...); };
    std::atomic<unsigned> next{0};
    auto worker = [&]() {
        std::unique_ptr<unsigned char[]> stacks(new unsigned char[(size_t)nt * kStack + 64]);
        std::unique_ptr<unsigned char[]> smem(new unsigned char[dyn_shared + 64]);
        emu_dyn_shared_ptr = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(smem.get()) + 63) & ~uintptr_t(63));
        emu_block_ctx ctx;
        ctx.nt = nt; ctx.body = &body; ctx.sp.resize(nt); ctx.done.resize(nt);
        emu_ctx = &ctx;
        blockDim = block; gridDim = grid;
        for (;;) {
            const unsigned b = next.fetch_add(1);
            if (b >= grid.x) break;
            blockIdx.x = b;
            ctx.blk_arrived = 0;
            for (unsigned w = 0; w < 16; w++) ctx.wave_arrived[w] = 0;
            for (unsigned t = 0; t < nt; t++) {
                // fresh stack: six zeroed callee-saved registers, the entry point, a dummy return slot (16-byte phase of a call)
                uintptr_t top = (reinterpret_cast<uintptr_t>(stacks.get()) + (size_t)(t + 1) * kStack) & ~uintptr_t(15);
                void **sp = reinterpret_cast<void **>(top) - 8;
                for (int k = 0; k < 6; k++) sp[k] = nullptr;
                sp[6] = reinterpret_cast<void *>(&emu_trampoline);
                sp[7] = nullptr;
                ctx.sp[t] = sp; ctx.done[t] = 0;
            }
            for (;;) {
                bool any = false;
                const unsigned long long before = ctx.progress;
                for (unsigned t = 0; t < nt; t++) {
                    if (ctx.done[t]) continue;
                    any = true;
                    ctx.cur = t; threadIdx.x = t;
                    emu_switch(&ctx.sched_sp, ctx.sp[t]);
                }
                if (!any) break;
                if (ctx.progress == before) { fprintf(stderr, "hip_emu: deadlock (a barrier some work-items never reach)\n"); abort(); }
            }
        }
        emu_ctx = nullptr; emu_dyn_shared_ptr = nullptr;
    };
    const unsigned nthreads = std::max(1u, std::min(std::min(grid.x, std::thread::hardware_concurrency()), 16u));
    if (nthreads == 1) { std::thread th(worker); th.join(); return; }
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nthreads; k++) th.emplace_back(worker);
    for (auto &x : th) x.join();
}

} // namespace emu
