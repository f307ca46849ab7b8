// EXPERIMENT (round 6): what would moving an UNSORTED batch's packed bases into window order cost?
//
// The review asked that the payload scatter of the device sort (bases + CIGAR moved next to the 32-byte record k_sort_scatter already
// moves, so that k_window's base fetch of a shuffled batch is as local as a sorted one's) be measured, not argued.  Its best case is
// known from the bench: k_window on C3 shuffled takes 1.26 ms against 0.99 ms sorted -- 0.27 ms to win, if the moved bases made the
// walk exactly as fast as on sorted input.  Its cost is a gather: every record's 75 bytes of packed bases (C3: 150-base reads) read
// from where the file order left them, written where the window order wants them -- 16.66 M random 75-byte reads, 1.25 GB each way.
// This program times exactly that access pattern, alone, in the two decompositions a kernel could use (a lane per record, its five
// 16-byte chunks in flight together; a lane per chunk), on the real record count, with the destination rows 80 bytes apart (aligned
// chunks) and packed 75 apart.  If the gather alone costs more than 0.27 ms the scatter cannot pay, whatever the rest of it costs
// (the exclusive scan of the moved sizes in sorted order, the CIGAR words).
//
//   hipcc --offload-arch=gfx950 -O3 scripts/exp/payload_gather_calib.hip -o exp/payload_gather_calib && exp/payload_gather_calib
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct __attribute__((packed, aligned(1))) Chunk { uint32_t x, y, z, w; };

// lane = record: five chunks requested together, stored to the record's row (stride `pitch`)
__global__ void __launch_bounds__(256) k_gather_rec(const uint8_t *src, const uint32_t *perm, uint8_t *dst, uint32_t n, uint32_t pitch) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const Chunk *s = reinterpret_cast<const Chunk *>(src + (uint64_t)perm[k] * 75u);
    const Chunk c0 = s[0], c1 = s[1], c2 = s[2], c3 = s[3], c4 = s[4];
    Chunk *d = reinterpret_cast<Chunk *>(dst + (uint64_t)k * pitch);
    d[0] = c0; d[1] = c1; d[2] = c2; d[3] = c3;
    if (pitch >= 80u) d[4] = c4;
    else { uint8_t *t = dst + (uint64_t)k * pitch + 64u; const uint32_t w[3] = {c4.x, c4.y, c4.z}; for (int b = 0; b < 11; b++) t[b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3))); }
}
// lane = chunk: consecutive lanes write consecutive 16 bytes of the destination
__global__ void __launch_bounds__(256) k_gather_chunk(const uint8_t *src, const uint32_t *perm, uint8_t *dst, uint32_t n) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t k = (uint32_t)(idx / 5u), c = (uint32_t)(idx % 5u);
    if (k >= n) return;
    const Chunk v = reinterpret_cast<const Chunk *>(src + (uint64_t)perm[k] * 75u)[c];
    reinterpret_cast<Chunk *>(dst)[idx] = v;      // rows of 80 bytes
}

int main() {
    const uint32_t n = 16664231u;
    const size_t src_bytes = (size_t)n * 75 + 64;
    std::vector<uint32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937_64 rng(7);
    std::shuffle(perm.begin(), perm.end(), rng);
    uint8_t *src, *dst; uint32_t *dperm;
    CK(hipMalloc(&src, src_bytes)); CK(hipMalloc(&dst, (size_t)n * 80 + 64)); CK(hipMalloc(&dperm, (size_t)n * 4));
    CK(hipMemset(src, 0x21, src_bytes));
    CK(hipMemcpy(dperm, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) -> int {
        for (int w = 0; w < 3; w++) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 10; r++) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); sum += ms;
        }
        printf("%-44s avg %.4f ms  best %.4f ms  (%.2f TB/s of 75 B read + written per record)\n", name, sum / 10, best, 2.0 * n * 75 / (best * 1e-3) / 1e12);
        return 0;
    };
    if (timeit("lane per record, rows 80 B apart", [&] { hipLaunchKernelGGL(k_gather_rec, dim3((n + 255) / 256), dim3(256), 0, 0, src, dperm, dst, n, 80u); })) return 1;
    if (timeit("lane per record, rows packed (75 B)", [&] { hipLaunchKernelGGL(k_gather_rec, dim3((n + 255) / 256), dim3(256), 0, 0, src, dperm, dst, n, 75u); })) return 1;
    if (timeit("lane per 16-byte chunk, rows 80 B apart", [&] { hipLaunchKernelGGL(k_gather_chunk, dim3((unsigned)(((uint64_t)n * 5 + 255) / 256)), dim3(256), 0, 0, src, dperm, dst, n); })) return 1;
    // the sorted case for scale: the same kernels with the identity permutation (what a copy of the bases costs when nothing is scattered)
    CK(hipMemcpy(dperm, [&] { std::iota(perm.begin(), perm.end(), 0u); return perm.data(); }(), (size_t)n * 4, hipMemcpyHostToDevice));
    if (timeit("identity order: lane per record, rows 80 B", [&] { hipLaunchKernelGGL(k_gather_rec, dim3((n + 255) / 256), dim3(256), 0, 0, src, dperm, dst, n, 80u); })) return 1;
    if (timeit("identity order: lane per chunk", [&] { hipLaunchKernelGGL(k_gather_chunk, dim3((unsigned)(((uint64_t)n * 5 + 255) / 256)), dim3(256), 0, 0, src, dperm, dst, n); })) return 1;
    return 0;
}
