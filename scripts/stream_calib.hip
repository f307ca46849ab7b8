// stream_calib.hip -- what k_prep's access pattern can reach on gfx950: N records whose fields lie in SEVEN arrays (six of
// 4 bytes, one of 8: the kd_batch structure-of-arrays) are read once and a 16-byte result per record is written.
//   hipcc --offload-arch=gfx950 -O3 scripts/stream_calib.hip -o exp/stream_calib && exp/stream_calib > profiles/r03_stream_calibration.json
// Variants (all touch the same bytes: 32 B read + 16 B written per record):
//   dword_strided   lane = record, R records per lane 256 apart, the loads of U records issued together (k_prep before round 3's
//                   rewrite: R = 32, U = 4): global_load_dword / dwordx2, one global_store_dwordx4 per record
//   x4_lane4        lane = FOUR consecutive records: one global_load_dwordx4 per 4-byte array (two for the 8-byte one),
//                   results stored as four dwordx4 with a lane stride of 64 B
//   x4_lane4_t      the same loads, results transposed through wavefront-private LDS so that every store instruction writes
//                   16 B per lane at a lane stride of 16 B (1 KiB contiguous per instruction)
//   read_only_*     the loads alone (a 4-byte sum per workgroup is the only store)
// Reported: GB/s over (read + written) bytes, best of 5 launches, n = 16.7 M records (C3's batch).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef unsigned long long u64;
struct Arrs { const uint32_t *a0, *a1, *a2, *a3, *a4, *a5; const u64 *b; uint4 *out; uint32_t *sink; u64 n; };

template <int R, int U, bool STORE>
__global__ void __launch_bounds__(256) k_dword_strided(Arrs A) {
    const uint32_t t = threadIdx.x;
    const u64 chunk0 = (u64)blockIdx.x * 256 * R;
    uint32_t acc = 0;
    for (int it0 = 0; it0 < R; it0 += U) {
        uint32_t v[U][6]; u64 w[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            u64 i = chunk0 + (u64)(it0 + u) * 256 + t; if (i >= A.n) i = 0;
            v[u][0] = A.a0[i]; v[u][1] = A.a1[i]; v[u][2] = A.a2[i]; v[u][3] = A.a3[i]; v[u][4] = A.a4[i]; v[u][5] = A.a5[i]; w[u] = A.b[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 i = chunk0 + (u64)(it0 + u) * 256 + t;
            uint4 r; r.x = v[u][0] + v[u][1]; r.y = v[u][2] ^ v[u][3]; r.z = v[u][4] + (uint32_t)w[u]; r.w = v[u][5] + (uint32_t)(w[u] >> 32);
            if (STORE) { if (i < A.n) A.out[i] = r; } else acc += r.x + r.y + r.z + r.w;
        }
    }
    if (!STORE && acc == 0x12345678u) A.sink[blockIdx.x] = acc;
}


// G groups of 1024 records per workgroup pass, lane = 4 consecutive records; U groups' loads in flight together
template <int G, int U, int MODE>   // MODE 0: read only, 1: stores strided by 64 B, 2: stores transposed
__global__ void __launch_bounds__(256) k_x4_lane4(Arrs A) {
    const uint32_t t = threadIdx.x;
    const u64 chunk0 = (u64)blockIdx.x * 1024 * G;
    uint32_t acc = 0;
    for (int g0 = 0; g0 < G; g0 += U) {
        uint4 v[U][6]; uint4 w[U][2];
#pragma unroll
        for (int u = 0; u < U; u++) {
            u64 i = chunk0 + (u64)(g0 + u) * 1024 + 4 * t; if (i + 4 > A.n) i = 0;
            v[u][0] = *(const uint4 *)(A.a0 + i); v[u][1] = *(const uint4 *)(A.a1 + i); v[u][2] = *(const uint4 *)(A.a2 + i);
            v[u][3] = *(const uint4 *)(A.a3 + i); v[u][4] = *(const uint4 *)(A.a4 + i); v[u][5] = *(const uint4 *)(A.a5 + i);
            w[u][0] = *(const uint4 *)(A.b + i); w[u][1] = *(const uint4 *)(A.b + i + 2);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 i = chunk0 + (u64)(g0 + u) * 1024 + 4 * t;
            uint4 r[4];
            const uint32_t *p0 = &v[u][0].x, *p1 = &v[u][1].x, *p2 = &v[u][2].x, *p3 = &v[u][3].x, *p4 = &v[u][4].x, *p5 = &v[u][5].x;
            const uint32_t wl[4] = {w[u][0].x, w[u][0].z, w[u][1].x, w[u][1].z}, wh[4] = {w[u][0].y, w[u][0].w, w[u][1].y, w[u][1].w};
#pragma unroll
            for (int k = 0; k < 4; k++) { r[k].x = p0[k] + p1[k]; r[k].y = p2[k] ^ p3[k]; r[k].z = p4[k] + wl[k]; r[k].w = p5[k] + wh[k]; }
            if (MODE == 0) { for (int k = 0; k < 4; k++) acc += r[k].x + r[k].y + r[k].z + r[k].w; }
            else if (MODE == 1) { if (i + 4 <= A.n) for (int k = 0; k < 4; k++) A.out[i + k] = r[k]; }
            else {
                // through LDS, wavefront-private (no barrier): lane l writes its four results, then reads records 64 k + l
                __shared__ uint4 s_t[4][256];
                uint4 *mine = s_t[t >> 6];
                const uint32_t lane = t & 63u;
#pragma unroll
                for (int k = 0; k < 4; k++) mine[4 * lane + k] = r[k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const u64 wb0 = chunk0 + (u64)(g0 + u) * 1024 + (u64)(t & ~63u) * 4;
                if (wb0 + 256 <= A.n) {
#pragma unroll
                    for (int k = 0; k < 4; k++) A.out[wb0 + 64 * k + lane] = mine[64 * k + lane];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    if (MODE == 0 && acc == 0x12345678u) A.sink[blockIdx.x] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const u64 n = 16660000ULL / 4096 * 4096;
    std::vector<uint32_t> h(n);
    for (u64 i = 0; i < n; i++) h[i] = (uint32_t)(i * 2654435761u);
    uint32_t *a[6]; u64 *b; uint4 *out, *out2; uint32_t *sink;
    for (int k = 0; k < 6; k++) { CK(hipMalloc(&a[k], n * 4 + 64)); CK(hipMemcpy(a[k], h.data(), n * 4, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&b, n * 8 + 64)); CK(hipMemcpy(b, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy((char *)b + n * 4, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, n * 16 + 64)); CK(hipMalloc(&out2, n * 16 + 64)); CK(hipMalloc(&sink, 1 << 20));
    Arrs A{a[0], a[1], a[2], a[3], a[4], a[5], b, out, sink, n};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Run { const char *name; double bytes; float ms; };
    std::vector<Run> runs;
    auto time = [&](const char *name, double bytes, auto launch) -> int {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
        }
        runs.push_back({name, bytes, best});
        return 0;
    };
    const double rb = 32.0 * n, wb = 16.0 * n;
#define STRIDED(R, U, S, nm) if (time(nm, rb + (S ? wb : 0), [&] { k_dword_strided<R, U, S><<<(unsigned)((n + 256 * R - 1) / (256 * R)), 256>>>(A); })) return 1;
#define LANE4(G, U, M, nm) if (time(nm, rb + (M ? wb : 0), [&] { k_x4_lane4<G, U, M><<<(unsigned)((n + 1024 * G - 1) / (1024 * G)), 256>>>(A); })) return 1;
    STRIDED(32, 4, true, "dword_strided R32 U4 + store16")
    STRIDED(32, 8, true, "dword_strided R32 U8 + store16")
    STRIDED(8, 4, true, "dword_strided R8 U4 + store16")
    STRIDED(32, 4, false, "read_only dword_strided R32 U4")
    STRIDED(8, 8, false, "read_only dword_strided R8 U8")
    LANE4(8, 1, 0, "read_only x4_lane4 G8 U1")
    LANE4(8, 2, 0, "read_only x4_lane4 G8 U2")
    LANE4(2, 2, 0, "read_only x4_lane4 G2 U2")
    LANE4(8, 1, 1, "x4_lane4 G8 U1 + store16 stride 64 B")
    LANE4(8, 2, 1, "x4_lane4 G8 U2 + store16 stride 64 B")
    LANE4(2, 1, 1, "x4_lane4 G2 U1 + store16 stride 64 B")
    LANE4(8, 1, 2, "x4_lane4 G8 U1 + store16 transposed through LDS")
    LANE4(8, 2, 2, "x4_lane4 G8 U2 + store16 transposed through LDS")
    LANE4(2, 1, 2, "x4_lane4 G2 U1 + store16 transposed through LDS")
    // the two store layouts must agree
    A.out = out; k_x4_lane4<8, 1, 1><<<(unsigned)((n + 8191) / 8192), 256>>>(A);
    A.out = out2; k_x4_lane4<8, 1, 2><<<(unsigned)((n + 8191) / 8192), 256>>>(A);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h1(n * 4), h2(n * 4);
    CK(hipMemcpy(h1.data(), out, n * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), out2, n * 16, hipMemcpyDeviceToHost));
    u64 diff = 0; for (u64 i = 0; i < n * 4; i++) diff += h1[i] != h2[i];
    printf("{\n \"n_records\": %llu, \"bytes_read_per_record\": 32, \"bytes_written_per_record\": 16, \"transposed_equals_strided\": %s,\n \"runs\": [\n", n, diff ? "false" : "true");
    for (size_t i = 0; i < runs.size(); i++)
        printf("  {\"name\": \"%s\", \"ms\": %.4f, \"GBps\": %.0f}%s\n", runs[i].name, runs[i].ms, runs[i].bytes / runs[i].ms * 1e-6, i + 1 < runs.size() ? "," : "");
    printf(" ]\n}\n");
    return diff ? 2 : 0;
}
