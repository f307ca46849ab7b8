// kd_kernels.h -- device code of the MI355X (gfx950 / CDNA4) pileup + consensus engine.
//
// Replaces the two Python loops of the reference:
//   parse_records record/CIGAR loop      /root/reference/kindel/kindel.py:40-81
//   consensus_sequence per-site loop     /root/reference/kindel/kindel.py:384-430
// (plus consensus() :369-381 and the depth min/max of build_report :450,477-479).
//
// This is integer histogramming: HBM/LDS-atomic bound, no MFMA.  Layout and kernels are
// described in DESIGN.md.  Short version:
//   tables   u32 tab[KD_NCH][S]  channel-major over "G-space" (all contigs back to back,
//            len+1 slots each, padded to 64) so that 64 lanes walking 64 consecutive sites
//            hit 64 consecutive dwords (coalesced atomics / stores, conflict-free LDS banks).
//   kd_prep.h     k_prep: lane per read, one wavefront per workgroup: classify (skip / regular / irregular / long CIGAR, plain), footprint,
//                 stats, deterministic insertion-event slots
//   kd_long.h     reads with > 16 CIGAR words (wavefront per read, tiles of 64 ops, lane = op): k_prep_long validates,
//                 k_long_reduce hands out slots, k_long_expand writes the read's ROW (one symbol per site: base / deleted /
//                 nothing, + "insertion in front"), its insertion events and its clips
//   kd_plan.h     k_plan_*: window -> candidate range (binary search on sorted starts) -> work items;
//                 k_sort_*: bucket sort by window (unsorted batches, the rows of long reads)
//   kd_window.h   k_window: persistent workgroups pull (window, slice) items; ONE LANE PER READ (second pass: per long
//                 read's ROW), 8 bases per dword, every base one ds_add_u32 into an LDS histogram (weights, deletions
//                 and both soft-clip weight tables; u16 counters, two sites per dword); one coalesced flush of
//                 the non-zero counters per item into HBM
//   kd_readwise.h k_cold_lane: short reads with S or I: clip start/end counters, insertion events;
//                 k_pileup_wave: one wavefront per read, every reference quirk incl. Python negative-index wrap,
//                 32-bit atomics straight to HBM: irregular reads, KD_MODE_GLOBAL
//   kd_errors.h   k_errors: which read / which reference exception (rare path; rides as k_cold_lane's last workgroup)
//   kd_ins.h      k_ins_*: insertion events -> open-addressing hash multiset -> per-site unique max
//   kd_cns.h      k_cns_*: per-site argmax / tie / indel rules, exclusive scan, byte emission
//   kd_gpu_inflate.h, kd_gpu_inflate2.h, kd_ingest.h   the device-side ingest (opt-in): k_gpu_inflate (raw DEFLATE of BGZF blocks, one
//                 wavefront each) or k_inflate_tokens + k_inflate_resolve (round 6: a lane per block records the matches, a wavefront per
//                 block resolves them), k_bam_*: the BAM record chain walked from speculative, verified starts -> the kd_batch arrays in HBM
//
// The file has no host API calls and only uses __syncthreads + atomics across lanes, so
// tests/emu/ can execute the same source on the CPU for logic checks (test infrastructure).
#pragma once
#include "kd_common.h"
#include "kd_prep.h"
#include "kd_long.h"
#include "kd_errors.h"
#include "kd_readwise.h"
#include "kd_plan.h"
#include "kd_window.h"
#include "kd_coop.h"
#include "kd_strip.h"
#include "kd_ins.h"
#include "kd_cns.h"
#include "kd_ingest.h"
#include "kd_gpu_inflate2.h"
