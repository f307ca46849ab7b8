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
//   * one std::thread per work-item of a block, blocks executed one after another;
//   * __syncthreads() is a std::barrier over the block's threads;
//   * __shared__ is a function-local static (valid because blocks run sequentially);
//   * atomics map to GCC __atomic builtins (same arithmetic on LDS and global memory).
// Kernels must not return before a later __syncthreads() (good HIP style anyway).
#pragma once
#include <barrier>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
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
inline thread_local std::barrier<> *emu_block_barrier = nullptr;
inline unsigned char *emu_dyn_shared_ptr = nullptr;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define KD_MUL24(a, b) ((uint32_t)(a) * (uint32_t)(b))
#define KD_ALIGNBYTE(hi, lo, sh) ((uint32_t)((((uint64_t)(hi) << 32) | (uint64_t)(lo)) >> (8 * (sh))))
#define KD_DYN_SHARED(type, name) type *name = reinterpret_cast<type *>(emu_dyn_shared_ptr)

static inline void __syncthreads() { emu_block_barrier->arrive_and_wait(); }

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

namespace emu {

// Run kernel(args...) over grid x block (1-D), blocks sequentially, threads concurrently.
template <class K, class... A>
void launch(K kernel, dim3 grid, dim3 block, size_t dyn_shared, A... args) {
    const unsigned nt = block.x;
    std::barrier<> bar((std::ptrdiff_t)nt);
    std::unique_ptr<unsigned char[]> smem(new unsigned char[dyn_shared + 64]);
    emu_dyn_shared_ptr = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(smem.get()) + 63) & ~uintptr_t(63));
    auto body = [&](unsigned t) {
        emu_block_barrier = &bar;
        blockDim = block;
        gridDim = grid;
        threadIdx.x = t;
        for (unsigned b = 0; b < grid.x; b++) {
            blockIdx.x = b;
            kernel(args...);
            bar.arrive_and_wait(); // block boundary
        }
    };
    std::vector<std::thread> th;
    th.reserve(nt);
    for (unsigned t = 0; t < nt; t++) th.emplace_back(body, t);
    for (auto &x : th) x.join();
    emu_dyn_shared_ptr = nullptr;
}

} // namespace emu
