// issue_calib.hip -- what a wave64 instruction costs on gfx950, for the instructions k_window is made of.
//   hipcc --offload-arch=gfx950 -O3 scripts/issue_calib.hip -o exp/issue_calib && exp/issue_calib > profiles/valu_issue_calibration.json
// Every CU runs WPS wavefronts per SIMD (256-thread workgroups); each wavefront issues ITER x UNROLL independent copies of
// the instruction under test between two s_memtime reads (the shader clock).  Reported per test:
//   cyc_per_inst_simd = elapsed cycles of the slowest wavefront / (instructions issued on its SIMD)   [VALU view]
//   cyc_per_inst_cu   = the same / (instructions issued on its CU)                                   [LDS / vector-memory view]
// (one LDS and one vector-memory pipeline per CU, four VALUs).  `wall_ms` cross-checks the clock.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define ITER 2048
#define UNROLL 16

enum {
    T_V_ADD, T_V_ADD_SDWA, T_V_PERM, T_V_LSHL_ADD, T_V_MAD_U24, T_V_BFE, T_V_AND, T_V_BFI, T_V_AND_OR, T_V_LSHRREV_B64, T_V_CNDMASK,
    T_DS_ADD_CONSEC, T_DS_ADD_STRIDE2, T_DS_ADD_STRIDE4, T_DS_ADD_STRIDE8, T_DS_ADD_STRIDE32, T_DS_ADD_RANDOM, T_DS_ADD_SAME,
    T_DS_ADD_GROUPS16, T_DS_ADD_PAIRMAJOR19, T_DS_ADD_SAME_PER16, T_DS_ADD_U64_CONSEC, T_DS_ADD_RTN_CONSEC, T_DS_READ_U8_CONSEC,
    T_DS_WRITE_B32_CONSEC, T_DS_BPERMUTE,
    T_LD_UBYTE, T_LD_USHORT, T_LD_DWORD, T_LD_DWORDX4, T_LD_UBYTE_GROUPS16,
    T_COUNT
};
static const char *NAMES[T_COUNT] = {
    "v_add_u32", "v_add_u32_sdwa", "v_perm_b32", "v_lshl_add_u32", "v_mad_u32_u24", "v_bfe_u32", "v_and_b32", "v_bfi_b32", "v_and_or_b32",
    "v_lshrrev_b64", "v_cndmask_b32",
    "ds_add_u32 consecutive dwords (conflict free)", "ds_add_u32 stride 2 dwords (2-way)", "ds_add_u32 stride 4 (4-way)",
    "ds_add_u32 stride 8 (8-way)", "ds_add_u32 stride 32 (32-way)", "ds_add_u32 random dword of 4096 per lane",
    "ds_add_u32 one address for all lanes", "ds_add_u32 four groups of 16 consecutive dwords at random bases, rows 2 KB apart chosen per lane",
    "ds_add_u32 k_window today: random pair * 19 + random channel 0..4", "ds_add_u32 one address per 16 lanes",
    "ds_add_u64 consecutive qwords", "ds_add_rtn_u32 consecutive", "ds_read_u8 consecutive bytes", "ds_write_b32 consecutive", "ds_bpermute_b32",
    "global_load_ubyte consecutive lanes (64 B per wave, L2/L1 resident)", "global_load_ushort consecutive (128 B per wave)",
    "global_load_dword consecutive (256 B per wave)", "global_load_dwordx4 consecutive (1 KiB per wave)",
    "global_load_ubyte four groups of 16 consecutive bytes, groups 75 B apart"};

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int T>
__global__ void __launch_bounds__(256) k_test(const uint8_t *gbuf, uint32_t *sink, unsigned long long *cycles) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[5120];   // 20 KiB: seven workgroups per CU fit
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    for (uint32_t i = t; i < 5120; i += 256) lds[i] = 0;
    __syncthreads();
    uint32_t a[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) a[u] = hash32(t * 131u + u);
    uint32_t b = hash32(t) | 1u, c = 0x03020100u;
    // LDS byte addresses (relative to this wavefront's 8 KiB quarter unless the pattern needs all of it)
    const uint32_t lbase = (uint32_t)(uintptr_t)lds;
    uint32_t addr = lbase;
    if (T == T_DS_ADD_CONSEC || T == T_DS_ADD_RTN_CONSEC || T == T_DS_WRITE_B32_CONSEC) addr = lbase + wave * 4096 + lane * 4;
    if (T == T_DS_ADD_STRIDE2) addr = lbase + wave * 4096 + lane * 8;
    if (T == T_DS_ADD_STRIDE4) addr = lbase + wave * 4096 + lane * 16;
    if (T == T_DS_ADD_STRIDE8) addr = lbase + wave * 4096 + lane * 32;
    if (T == T_DS_ADD_STRIDE32) addr = lbase + (lane * 128 + wave * 4) % 16384;
    if (T == T_DS_ADD_RANDOM) addr = lbase + (hash32(t * 7919u + blockIdx.x) % 4096u) * 4;
    if (T == T_DS_ADD_SAME) addr = lbase + wave * 256;
    if (T == T_DS_ADD_SAME_PER16) addr = lbase + wave * 256 + (lane >> 4) * 4;
    if (T == T_DS_ADD_GROUPS16)   // group g of 16 lanes: consecutive site pairs from a random start; the row (nibble) differs per lane
        addr = lbase + (hash32((t >> 4) * 977u + blockIdx.x) % 384u + (lane & 15u)) * 4 + (hash32(t) % 5u) * 2048u;
    if (T == T_DS_ADD_PAIRMAJOR19) addr = lbase + ((hash32(t * 31u + blockIdx.x) % 160u) * 19u + hash32(t) % 5u) * 4;
    if (T == T_DS_ADD_U64_CONSEC) addr = lbase + wave * 4096 + lane * 8;
    if (T == T_DS_READ_U8_CONSEC) addr = lbase + wave * 4096 + lane;
    if (T == T_DS_BPERMUTE) addr = ((lane * 4u + lane / 4u) & 63u) * 4u;
    const uint8_t *gp = gbuf + (size_t)blockIdx.x * 4096 + wave * 1024;
    if (T == T_LD_UBYTE) gp += lane;
    if (T == T_LD_USHORT) gp += lane * 2;
    if (T == T_LD_DWORD) gp += lane * 4;
    if (T == T_LD_DWORDX4) gp = gbuf + (size_t)blockIdx.x * 16384 + wave * 4096 + lane * 16;
    if (T == T_LD_UBYTE_GROUPS16) gp += (lane >> 4) * 75 + (lane & 15u);
    uint32_t acc = 0;
    uint4 acc4 = make_uint4(0, 0, 0, 0);
    unsigned long long d64 = 0x0000000100000001ULL, dd[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) dd[u] = hash32(u + t) * 0x100000001ULL;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (T == T_V_ADD) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[u]) : "v"(b));
            if (T == T_V_ADD_SDWA) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[u]) : "v"(b));
            if (T == T_V_PERM) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[u]) : "v"(b), "v"(c));
            if (T == T_V_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %1, 3, %0" : "+v"(a[u]) : "v"(b));
            if (T == T_V_MAD_U24) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b), "v"(c));
            if (T == T_V_BFE) asm volatile("v_bfe_u32 %0, %0, 4, 28" : "+v"(a[u]));
            if (T == T_V_AND) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[u]) : "v"(b));
            if (T == T_V_BFI) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b), "v"(c));
            if (T == T_V_AND_OR) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b), "v"(c));
            if (T == T_V_LSHRREV_B64) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(dd[u]));
            if (T == T_V_CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[u]) : "v"(b));
            if (T >= T_DS_ADD_CONSEC && T <= T_DS_ADD_SAME_PER16) {
                if (u % 4 == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(b) : "memory");
                if (u % 4 == 1) asm volatile("ds_add_u32 %0, %1 offset:1024" ::"v"(addr), "v"(b) : "memory");
                if (u % 4 == 2) asm volatile("ds_add_u32 %0, %1 offset:2048" ::"v"(addr), "v"(b) : "memory");
                if (u % 4 == 3) asm volatile("ds_add_u32 %0, %1 offset:3072" ::"v"(addr), "v"(b) : "memory");
            }
            if (T == T_DS_ADD_U64_CONSEC) asm volatile("ds_add_u64 %0, %1" ::"v"(addr), "v"(d64) : "memory");
            if (T == T_DS_ADD_RTN_CONSEC) asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(a[u]) : "v"(addr), "v"(b) : "memory");
            if (T == T_DS_READ_U8_CONSEC) asm volatile("ds_read_u8 %0, %1" : "=v"(a[u]) : "v"(addr) : "memory");
            if (T == T_DS_WRITE_B32_CONSEC) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(b) : "memory");
            if (T == T_DS_BPERMUTE) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(a[u]) : "v"(addr), "v"(b) : "memory");
            if (T == T_LD_UBYTE || T == T_LD_UBYTE_GROUPS16) asm volatile("global_load_ubyte %0, %1, off" : "=v"(a[u]) : "v"(gp) : "memory");
            if (T == T_LD_USHORT) asm volatile("global_load_ushort %0, %1, off" : "=v"(a[u]) : "v"(gp) : "memory");
            if (T == T_LD_DWORD) asm volatile("global_load_dword %0, %1, off" : "=v"(a[u]) : "v"(gp) : "memory");
            if (T == T_LD_DWORDX4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(acc4) : "v"(gp) : "memory");
        }
        if (T >= T_DS_ADD_CONSEC) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc ^= a[u];
    acc ^= acc4.x ^ acc4.y ^ acc4.z ^ acc4.w ^ (uint32_t)d64;
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc ^= (uint32_t)dd[u];
    if (acc == 0x12345678u) *sink = acc + lds[t];
    if (lane == 0) atomicMax(cycles, t1 - t0);
}

template <int T>
static void run(int wps, const uint8_t *gbuf, uint32_t *sink, unsigned long long *d_cyc, int n_cus, bool last) {
    // wps wavefronts per SIMD = wps 256-thread workgroups per CU (a workgroup puts one wavefront on each SIMD)
    const int grid = n_cus * wps;
    hipMemset(d_cyc, 0, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_test<T><<<grid, 256>>>(gbuf, sink, d_cyc);   // warm-up
    hipDeviceSynchronize();
    hipMemset(d_cyc, 0, 8);
    hipEventRecord(e0);
    k_test<T><<<grid, 256>>>(gbuf, sink, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc = 0;
    hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    const double per_wave = (double)ITER * UNROLL;
    // s_memtime ticks = shader cycles (MI355X_MICROARCH.md); the wall clock of the same launch is printed beside them
    printf("  {\"inst\": \"%s\", \"waves_per_simd\": %d, \"cycles_slowest_wave\": %llu, \"wall_ms\": %.4f, "
           "\"cyc_per_inst_simd\": %.3f, \"cyc_per_inst_cu\": %.3f, \"ns_per_inst_simd\": %.4f, \"ns_per_inst_cu\": %.4f}%s\n",
           NAMES[T], wps, cyc, ms, cyc / (per_wave * wps), cyc / (per_wave * wps * 4), ms * 1e6 / (per_wave * wps),
           ms * 1e6 / (per_wave * wps * 4), last ? "" : ",");
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int T>
static void sweep(const uint8_t *gbuf, uint32_t *sink, unsigned long long *d_cyc, int n_cus, bool last = false) {
    run<T>(1, gbuf, sink, d_cyc, n_cus, false);
    run<T>(2, gbuf, sink, d_cyc, n_cus, false);
    run<T>(4, gbuf, sink, d_cyc, n_cus, false);
    run<T>(7, gbuf, sink, d_cyc, n_cus, last);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cus = p.multiProcessorCount;
    uint8_t *gbuf; uint32_t *sink; unsigned long long *d_cyc;
    const size_t gb = (size_t)n_cus * 8 * 16384 + 65536;
    hipMalloc(&gbuf, gb); hipMalloc(&sink, 4); hipMalloc(&d_cyc, 8);
    hipMemset(gbuf, 1, gb);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz_max\": %d, \"iter\": %d, \"unroll\": %d,\n"
           " \"note\": \"ns per wave64 instruction at saturation = wall time / instructions per SIMD (VALU) or per CU (LDS, vector memory); "
           "cyc_* from s_memtime of the slowest wavefront\",\n \"tests\": [\n",
           p.name, n_cus, p.clockRate / 1000, ITER, UNROLL);
    sweep<T_V_ADD>(gbuf, sink, d_cyc, n_cus); sweep<T_V_ADD_SDWA>(gbuf, sink, d_cyc, n_cus); sweep<T_V_PERM>(gbuf, sink, d_cyc, n_cus);
    sweep<T_V_LSHL_ADD>(gbuf, sink, d_cyc, n_cus); sweep<T_V_MAD_U24>(gbuf, sink, d_cyc, n_cus); sweep<T_V_BFE>(gbuf, sink, d_cyc, n_cus);
    sweep<T_V_AND>(gbuf, sink, d_cyc, n_cus); sweep<T_V_BFI>(gbuf, sink, d_cyc, n_cus); sweep<T_V_AND_OR>(gbuf, sink, d_cyc, n_cus);
    sweep<T_V_LSHRREV_B64>(gbuf, sink, d_cyc, n_cus); sweep<T_V_CNDMASK>(gbuf, sink, d_cyc, n_cus);
    sweep<T_DS_ADD_CONSEC>(gbuf, sink, d_cyc, n_cus); sweep<T_DS_ADD_STRIDE2>(gbuf, sink, d_cyc, n_cus); sweep<T_DS_ADD_STRIDE4>(gbuf, sink, d_cyc, n_cus);
    sweep<T_DS_ADD_STRIDE8>(gbuf, sink, d_cyc, n_cus); sweep<T_DS_ADD_STRIDE32>(gbuf, sink, d_cyc, n_cus); sweep<T_DS_ADD_RANDOM>(gbuf, sink, d_cyc, n_cus);
    sweep<T_DS_ADD_SAME>(gbuf, sink, d_cyc, n_cus); sweep<T_DS_ADD_GROUPS16>(gbuf, sink, d_cyc, n_cus); sweep<T_DS_ADD_PAIRMAJOR19>(gbuf, sink, d_cyc, n_cus);
    sweep<T_DS_ADD_SAME_PER16>(gbuf, sink, d_cyc, n_cus); sweep<T_DS_ADD_U64_CONSEC>(gbuf, sink, d_cyc, n_cus); sweep<T_DS_ADD_RTN_CONSEC>(gbuf, sink, d_cyc, n_cus);
    sweep<T_DS_READ_U8_CONSEC>(gbuf, sink, d_cyc, n_cus); sweep<T_DS_WRITE_B32_CONSEC>(gbuf, sink, d_cyc, n_cus); sweep<T_DS_BPERMUTE>(gbuf, sink, d_cyc, n_cus);
    sweep<T_LD_UBYTE>(gbuf, sink, d_cyc, n_cus); sweep<T_LD_USHORT>(gbuf, sink, d_cyc, n_cus); sweep<T_LD_DWORD>(gbuf, sink, d_cyc, n_cus);
    sweep<T_LD_DWORDX4>(gbuf, sink, d_cyc, n_cus); sweep<T_LD_UBYTE_GROUPS16>(gbuf, sink, d_cyc, n_cus, true);
    printf(" ]}\n");
    return 0;
}
