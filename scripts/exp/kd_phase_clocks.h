// scripts/exp/kd_phase_clocks.h -- PROFILING OVERLAY, never part of the product build.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -include scripts/exp/kd_phase_clocks.h \
//         kindel_amd/csrc/kindel_hip.hip kindel_amd/csrc/kd_decode.cpp -lz -o exp/libkd_phase.so
//   KD_BENCH_LIB=exp/libkd_phase.so python bench.py --no-graph
// Fills the hooks kd_common.h leaves empty: every wavefront of k_window adds the clocks it spent per phase (dequeue, zeroing,
// classification, plain walk, complex walk, barrier wait, flush) to eight extra status words, kd_finalize prints their shares.
// Results are unchanged (the tables are the product's); only timing is perturbed by the clock reads.
#pragma once
#define KD_PHASE_CLOCKS 1
#ifndef KD_PHASE_CLOCKS_ROWS_ONLY
#define KD_PHASE_CLOCKS_ROWS_ONLY 0   // 1: only the row pass (k_window<true>) adds its clocks
#endif
#define KD_PHASE_DECL \
    long long c_zero = 0, c_cls = 0, c_plain = 0, c_cplx = 0, c_wait = 0, c_flush = 0, c_deq = 0, c_mark = clock64();
#define KD_MARK(acc) { const long long n_ = clock64(); acc += n_ - c_mark; c_mark = n_; }
#define KD_PHASE_COMMIT(status, rows)                                                                              \
    if ((threadIdx.x & 63u) == 0 && ((rows) || !KD_PHASE_CLOCKS_ROWS_ONLY)) {   /* lane 0 of every wavefront */    \
        atomicAdd(&status[KDS_DBG0], (kd_u64)c_deq); atomicAdd(&status[KDS_DBG0 + 1 * KDS_STRIDE], (kd_u64)c_zero);                 \
        atomicAdd(&status[KDS_DBG0 + 2 * KDS_STRIDE], (kd_u64)c_cls); atomicAdd(&status[KDS_DBG0 + 3 * KDS_STRIDE], (kd_u64)c_plain);                \
        atomicAdd(&status[KDS_DBG0 + 4 * KDS_STRIDE], (kd_u64)c_cplx); atomicAdd(&status[KDS_DBG0 + 5 * KDS_STRIDE], (kd_u64)c_wait);                \
        atomicAdd(&status[KDS_DBG0 + 6 * KDS_STRIDE], (kd_u64)c_flush); atomicAdd(&status[KDS_DBG0 + 7 * KDS_STRIDE], 1ULL);                         \
    }
#define KD_PHASE_REPORT(h)                                                                                         \
    {                                                                                                              \
        const char *nm_[8] = {"dequeue", "zero", "classify", "plain", "complex", "barrier-wait", "flush", "waves"}; \
        double tot_ = 0;                                                                                           \
        for (int k_ = 0; k_ < 7; k_++) tot_ += (double)(h)[KDS_DBG0 + k_ * KDS_STRIDE];                            \
        fprintf(stderr, "k_window phase clocks (sum over wavefronts):");                                           \
        for (int k_ = 0; k_ < 8; k_++)                                                                             \
            fprintf(stderr, " %s=%.3g(%.1f%%)", nm_[k_], (double)(h)[KDS_DBG0 + k_ * KDS_STRIDE], 100.0 * (h)[KDS_DBG0 + k_ * KDS_STRIDE] / tot_); \
        fprintf(stderr, "\n");                                                                                     \
    }
// k_prep: every wavefront's start, loop end and end on the constant-rate clock (wall_clock64: 100 MHz) -- sums of the loop and
// the tail, the longest wavefront, the first start and the last end (the kernel's span as its wavefronts see it).
#define KD_PREP_CLK_DECL const unsigned long long pc_t0 = wall_clock64(); unsigned long long pc_t1 = pc_t0;
#define KD_PREP_CLK_LOOP_END pc_t1 = wall_clock64();
#define KD_PREP_CLK_COMMIT(status)                                                                                 \
    {                                                                                                              \
        const unsigned long long pc_t2 = wall_clock64();                                                           \
        if (threadIdx.x == 0) {                                                                                    \
            atomicAdd(&status[KDS_DBG0 + 8 * KDS_STRIDE], pc_t1 - pc_t0); atomicAdd(&status[KDS_DBG0 + 9 * KDS_STRIDE], pc_t2 - pc_t1);   \
            atomicMax(&status[KDS_DBG0 + 10 * KDS_STRIDE], ((pc_t2 - pc_t0) << 20) | (blockIdx.x < 0xfffffu ? blockIdx.x : 0xfffffu)); atomicMax(&status[KDS_DBG0 + 11 * KDS_STRIDE], ~pc_t0);        \
            atomicMax(&status[KDS_DBG0 + 12 * KDS_STRIDE], pc_t2); atomicAdd(&status[KDS_DBG0 + 13 * KDS_STRIDE], 1ULL);                  \
        }                                                                                                          \
    }
#undef KD_PHASE_REPORT
#define KD_PHASE_REPORT(h)                                                                                         \
    {                                                                                                              \
        const char *nm_[8] = {"dequeue", "zero", "classify", "plain", "complex", "barrier-wait", "flush", "waves"}; \
        double tot_ = 0;                                                                                           \
        for (int k_ = 0; k_ < 7; k_++) tot_ += (double)(h)[KDS_DBG0 + k_ * KDS_STRIDE];                            \
        fprintf(stderr, "k_window phase clocks (sum over wavefronts):");                                           \
        for (int k_ = 0; k_ < 8; k_++)                                                                             \
            fprintf(stderr, " %s=%.3g(%.1f%%)", nm_[k_], (double)(h)[KDS_DBG0 + k_ * KDS_STRIDE], 100.0 * (h)[KDS_DBG0 + k_ * KDS_STRIDE] / tot_); \
        fprintf(stderr, "\n");                                                                                     \
        const double nw_ = (double)(h)[KDS_DBG0 + 13 * KDS_STRIDE];                                                \
        if (nw_ > 0)                                                                                               \
            fprintf(stderr, "k_prep wavefronts (last batch; 100 MHz ticks as us): %.0f waves, loop avg %.2f us, tail avg %.2f us, longest %.2f us (wavefront %llu), first start -> last end %.2f us\n", \
                    nw_, (h)[KDS_DBG0 + 8 * KDS_STRIDE] / nw_ / 100.0, (h)[KDS_DBG0 + 9 * KDS_STRIDE] / nw_ / 100.0, ((h)[KDS_DBG0 + 10 * KDS_STRIDE] >> 20) / 100.0, \
                    (unsigned long long)((h)[KDS_DBG0 + 10 * KDS_STRIDE] & 0xfffffu), \
                    ((double)(h)[KDS_DBG0 + 12 * KDS_STRIDE] - (double)(~(h)[KDS_DBG0 + 11 * KDS_STRIDE])) / 100.0);    \
    }
