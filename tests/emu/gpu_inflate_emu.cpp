// gpu_inflate_emu.cpp -- TEST INFRASTRUCTURE ONLY: scripts/gpu_inflate_proto.h (the GPU-side DEFLATE prototype) executed on the CPU
// through tests/emu/hip_emu.h, for tests/test_gpu_inflate_proto.py (checked against zlib).
//   g++ -O1 -std=c++20 -shared -fPIC -pthread tests/emu/gpu_inflate_emu.cpp -o tests/emu/libgpu_inflate_emu.so
#include "hip_emu.h"
#include "../../scripts/gpu_inflate_proto.h"

extern "C" int gi_inflate_blocks(const uint8_t *comp, const GiBlock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, int, float *ms) {
    emu::launch(k_gpu_inflate, dim3(n_blocks), dim3(64), 0, comp, blocks, n_blocks, out, status);
    if (ms) *ms = 0;
    return 0;
}

extern "C" int gi_crc_blocks(const uint8_t *comp, const GiBlock *blocks, uint32_t n_blocks, const uint8_t *out, uint32_t *n_bad, uint32_t grid) {
    emu::launch(k_bgzf_crc, dim3(grid), dim3(64), 0, comp, blocks, n_blocks, out, n_bad);
    return 0;
}
