// gpu_inflate_emu.cpp -- TEST INFRASTRUCTURE ONLY: scripts/gpu_inflate_proto.h (the GPU-side DEFLATE prototype) executed on the CPU
// through tests/emu/hip_emu.h, for tests/test_gpu_inflate_proto.py (checked against zlib).
//   g++ -O1 -std=c++20 -shared -fPIC -pthread tests/emu/gpu_inflate_emu.cpp -o tests/emu/libgpu_inflate_emu.so
#include "hip_emu.h"
#include "../../scripts/gpu_inflate_proto.h"
#include "../../kindel_amd/csrc/kd_gpu_inflate2.h"

extern "C" int gi_inflate_blocks(const uint8_t *comp, const GiBlock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, int, float *ms) {
    emu::launch(k_gpu_inflate, dim3(n_blocks), dim3(64), 0, comp, blocks, n_blocks, out, status);
    if (ms) *ms = 0;
    return 0;
}

extern "C" int gi_crc_blocks(const uint8_t *comp, const GiBlock *blocks, uint32_t n_blocks, const uint8_t *out, uint32_t *n_bad, uint32_t grid) {
    emu::launch(k_bgzf_crc, dim3(grid), dim3(64), 0, comp, blocks, n_blocks, out, n_bad);
    return 0;
}

// round 6: the two-pass inflate (kd_gpu_inflate2.h): lane per block -> tokens, wavefront per block -> matches
extern "C" int gi_inflate_blocks2(const uint8_t *comp, const GiBlock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, int, float *ms) {
    unsigned long long total_out = 0;
    for (uint32_t b = 0; b < n_blocks; b++) total_out = std::max<unsigned long long>(total_out, blocks[b].out_off + blocks[b].out_len);
    const size_t n_tokens = (size_t)gi2_tok_off(total_out, n_blocks) + 16;
    uint32_t *tokens = (uint32_t *)malloc(n_tokens * 4), *n_tok = (uint32_t *)malloc(((size_t)n_blocks + 1) * 4);
    uint32_t *canary = tokens + n_tokens - 16;
    for (int k = 0; k < 16; k++) canary[k] = 0xA5A5A5A5u;
    // (a launch smaller than the work: the lanes that finish first take the rest from the counter, as on the GPU)
    uint32_t work = 0;
    const unsigned wgs = std::max(1u, ((n_blocks + GI2_WG - 1) / GI2_WG + 1) / 2);
    emu::launch(k_inflate_tokens, dim3(wgs), dim3(GI2_WG), GI2_LDS_BYTES, comp, blocks, n_blocks, out, tokens, n_tok, status, 0u, &work);
    emu::launch(k_inflate_resolve, dim3(n_blocks), dim3(64), 0, blocks, n_blocks, out, (const uint32_t *)tokens, (const uint32_t *)n_tok, (const uint32_t *)status, 0u);
    int rc = 0;
    for (int k = 0; k < 16; k++) if (canary[k] != 0xA5A5A5A5u) rc = 7;      // wrote past the token regions
    free(tokens); free(n_tok);
    if (ms) *ms = 0;
    return rc;
}
