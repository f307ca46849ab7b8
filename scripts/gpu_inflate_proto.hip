// gpu_inflate_proto.hip -- PROTOTYPE driver of scripts/gpu_inflate_proto.h (not part of libkindel_hip.so):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC scripts/gpu_inflate_proto.hip -o exp/libgpu_inflate_proto.so
// C-ABI for scripts/gpu_inflate_proto.py: all pointers are DEVICE pointers but `ms`.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "gpu_inflate_proto.h"
#include "../kindel_amd/csrc/kd_gpu_inflate2.h"

extern "C" int gi_inflate_blocks(const uint8_t *comp, const GiBlock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, int repeat,
                                 float *ms) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    float best = 1e30f;
    for (int r = 0; r < (repeat > 0 ? repeat : 1); r++) {
        hipEventRecord(e0, 0);
        k_gpu_inflate<<<n_blocks, 64, 0, 0>>>(comp, blocks, n_blocks, out, status);
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) return 2;
        float t = 0;
        hipEventElapsedTime(&t, e0, e1);
        if (t < best) best = t;
    }
    if (ms) *ms = best;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gi_crc_blocks(const uint8_t *comp, const GiBlock *blocks, uint32_t n_blocks, const uint8_t *out, uint32_t *n_bad, uint32_t grid) {
    k_bgzf_crc<<<grid, 64, 0, 0>>>(comp, blocks, n_blocks, out, n_bad);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}

// round 6: the two-pass inflate (kd_gpu_inflate2.h).  ms[0] = both passes, ms[1] = pass 1 (k_inflate_tokens) alone, best of `repeat`.
extern "C" int gi_inflate_blocks2(const uint8_t *comp, const GiBlock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, int repeat,
                                  float *ms, unsigned long long total_out) {
    hipEvent_t e0, e1, e2;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess) return 1;
    uint32_t *tokens = nullptr, *n_tok = nullptr;
    const size_t n_tokens = (size_t)gi2_tok_off(total_out, n_blocks) + 16;
    if (hipMalloc(&tokens, n_tokens * 4) != hipSuccess || hipMalloc(&n_tok, ((size_t)n_blocks + 2) * 4) != hipSuccess) return 4;
    uint32_t *work = n_tok + n_blocks + 1;
    int cus = 256;
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, 0) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
    const int wpc = getenv("GI2_WPC") ? atoi(getenv("GI2_WPC")) : 4;      // (experiment: resident wavefronts per CU the launch asks for)
    const unsigned wgs = (unsigned)std::min<size_t>(((size_t)n_blocks + GI2_WG - 1) / GI2_WG, (size_t)cus);      // one workgroup of four wavefronts per CU
    (void)wpc;
    if (hipFuncSetAttribute((const void *)k_inflate_tokens, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GI2_LDS_BYTES) != hipSuccess) return 5;
    float best = 1e30f, best1 = 1e30f;
    for (int r = 0; r < (repeat > 0 ? repeat : 1); r++) {
        hipMemsetAsync(work, 0, 4, 0);
        hipEventRecord(e0, 0);
        k_inflate_tokens<<<wgs, GI2_WG, GI2_LDS_BYTES, 0>>>(comp, blocks, n_blocks, out, tokens, n_tok, status, 0u, work);
        hipEventRecord(e1, 0);
        k_inflate_resolve<<<n_blocks, 64, 0, 0>>>(blocks, n_blocks, out, tokens, n_tok, status, 0u);
        hipEventRecord(e2, 0);
        if (hipEventSynchronize(e2) != hipSuccess) return 2;
        float t = 0, t1 = 0;
        hipEventElapsedTime(&t, e0, e2); hipEventElapsedTime(&t1, e0, e1);
        if (t < best) { best = t; best1 = t1; }
    }
    if (ms) { ms[0] = best; ms[1] = best1; }
    hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2);
    hipFree(tokens); hipFree(n_tok);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
