// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the pileup
// kernels (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ...
// other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O3 scripts/fetch_calib.hip -o exp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- exp/fetch_calib      (scripts/gpu_calib.sh parses the counters)
// Every kernel reads (or writes) each byte of a 2 GiB buffer exactly once (far beyond the 256 MB Infinity Cache), so the
// true HBM byte count is known: 2^31.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

struct __attribute__((packed, aligned(1))) Chunk { uint32_t x, y, z, w; };

// wide coalesced stream: 16 B per lane, consecutive lanes consecutive
__global__ void calib_stream16(const uint4 *p, size_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// k_window's plain walk: one lane per 75-byte record (150 packed bases), five UNALIGNED 16-byte loads per record,
// records back to back, lanes of a wavefront `rows` records apart
__global__ void calib_records75(const uint8_t *p, size_t n_rec, uint32_t rows, uint32_t *sink) {
    uint32_t acc = 0;
    const size_t per_block = (size_t)64 * rows * (blockDim.x / 64);
    for (size_t b0 = (size_t)blockIdx.x * per_block; b0 < n_rec; b0 += (size_t)gridDim.x * per_block) {
        const size_t wbase = b0 + (size_t)(threadIdx.x / 64) * 64 * rows;
        for (uint32_t r = 0; r < rows; r++) {
            const size_t rec = wbase + (size_t)(threadIdx.x & 63) * rows + r;
            if (rec >= n_rec) continue;
            const Chunk *c = reinterpret_cast<const Chunk *>(p + rec * 75);
#pragma unroll
            for (int k = 0; k < 5; k++) {   // the 5th load overlaps the next record by 5 bytes, like the kernel's
                const Chunk v = c[k];
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}
// 16-byte records gathered by a permutation-free stride (k_cold_lane / rinfo style: 16 B per lane, coalesced)
__global__ void calib_stream8(const uint2 *p, size_t n8, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const uint2 v = p[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// 32-bit atomic adds, consecutive lanes consecutive dwords (the table flush)
__global__ void calib_atomic_flush(uint32_t *p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) atomicAdd(&p[i], 1u);
}
// plain 16-byte stores (memset-like)
__global__ void calib_store16(uint4 *p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1u, 2u, 3u, 4u);
}

int main() {
    const size_t N = (size_t)1 << 31;
    uint8_t *buf; uint32_t *sink;
    if (hipMalloc(&buf, N + 64) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, N + 64);
    hipDeviceSynchronize();
    const int grid = 256 * 8, block = 256;
    for (int rep = 0; rep < 3; rep++) {
        calib_stream16<<<grid, block>>>((const uint4 *)buf, N / 16, sink);
        calib_records75<<<grid, block>>>(buf, N / 75, 16, sink);
        calib_stream8<<<grid, block>>>((const uint2 *)buf, N / 8, sink);
        calib_atomic_flush<<<grid, block>>>((uint32_t *)buf, N / 4);
        calib_store16<<<grid, block>>>((uint4 *)buf, N / 16);
    }
    hipDeviceSynchronize();
    printf("fetch_calib: each kernel touched %zu bytes once\n", N);
    return 0;
}
