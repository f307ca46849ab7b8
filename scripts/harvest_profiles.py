#!/usr/bin/env python
"""Copy the judged summaries of the last `scripts/gpu_round.sh` run from gpurun_out/ (scratch) into profiles/ (tracked).

  profiles/<tag>_c3_kernel_stats.csv          rocprofv3 --kernel-trace --stats, per-kernel summary of `python bench.py`
  profiles/<tag>_c3_pmc_per_dispatch.json     rocprofv3 --pmc passes (scripts/gpu_pmc.sh), counters averaged per launch
  profiles/pmc_traffic.json                   HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (read by bench.py)
  profiles/<tag>_c3_bench.json                the bench line (default invocation, with cpu_baseline)
  profiles/<tag>_c3_bench_under_rocprof.json  the bench line of the run rocprofv3 traced
  profiles/<tag>_{C2,C4,C5}_bench.json        the other SURVEY 8d configs
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
# files older than this many hours are another round's leftovers (gpurun_out/ is merged into, never cleaned)
newer_than = __import__("time").time() - 3600.0 * float(sys.argv[2] if len(sys.argv) > 2 else 6)


def first_json_line(path):
    for line in open(path):
        if line.startswith('{"metric'):
            return json.loads(line)
    raise SystemExit("no bench line in " + path)


st = sorted(glob.glob(os.path.join(O, "prof_c3", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)   # newest run first
if st:
    shutil.copy(st[0], os.path.join(P, "%s_c3_kernel_stats.csv" % tag))
    rows = list(csv.reader(open(st[0])))   # the same table without the synthetic-input generator's torch kernels
    with open(os.path.join(P, "%s_c3_kernel_stats_kd_only.csv" % tag), "w", newline="") as fh:
        w = csv.writer(fh)
        def kname(n):     # "void k_window<false>(KdReads, ...)" -> "k_window<false>"
            n = n.split("(")[0]
            return n[5:] if n.startswith("void ") else n
        for r in rows[:1] + [r for r in rows[1:] if r and kname(r[0]).startswith("k_")]:
            w.writerow([kname(r[0])] + r[1:])
for src, dst in (("bench_c3.json", "%s_c3_bench.json"), ("prof_c3_bench.json", "%s_c3_bench_under_rocprof.json"),
                 ("bench_C2.json", "%s_C2_bench.json"), ("bench_C4.json", "%s_C4_bench.json"), ("bench_C5.json", "%s_C5_bench.json"),
                 ("exp_strip.json", "%s_c3_bench_mode_strip.json"), ("exp_shuf.json", "%s_c3_bench_shuffled.json")):
    p = os.path.join(O, src)
    if os.path.exists(p) and os.path.getmtime(p) > newer_than:
        json.dump(first_json_line(p), open(os.path.join(P, dst % tag), "w"), indent=1, sort_keys=True)

def pmc_per_dispatch(tagname):
    """counters of one workload averaged per launch, as scripts/gpu_pmc.sh left them on the GPU box (pmc_<tag>_summary.json)"""
    p = os.path.join(O, "pmc_%s_summary.json" % tagname)
    if not os.path.exists(p) or os.path.getmtime(p) < newer_than:
        return {}
    return json.load(open(p))


# profiles/pmc_traffic.json: {"<config>|<mode>|<order>": {kernel: {"bytes", "raw_bytes"}}, "_note": ...} -- bench.py looks its
# workload up by that key and reports null when no record of it exists
note = ("(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch.  The factors are calibrated on this GPU in the kernels' own access patterns "
        "(profiles/fetch_calibration.json, scripts/fetch_calib.hip: a 2 GiB buffer touched exactly once): FETCH_SIZE counts half "
        "the bytes (x2.00 for coalesced 8- and 16-byte streams; k_window's per-lane unaligned 16-byte loads over 75-byte records "
        "fetch 1.30x their unique bytes, which this figure includes as real traffic), WRITE_SIZE x1.00 for stores and for atomics "
        "(whose read half is not counted).  raw_bytes = (FETCH_SIZE + WRITE_SIZE)*1024, uncorrected")
tp = os.path.join(P, "pmc_traffic.json")
traffic = {}
if os.path.exists(tp):
    old = json.load(open(tp))
    traffic = {k: v for k, v in old.items() if "|" in k}       # (a pre-round-4 file was keyed by kernel: dropped)
def bench_library(tagname):
    """library_sha256 of this round's bench file of the same workload (None: no such file, or written by a bench.py without it)"""
    src = {"C3": "bench_c3.json", "C2": "bench_C2.json", "C4": "bench_C4.json", "C5": "bench_C5.json", "C3_shuffle": "exp_shuf.json"}[tagname]
    p = os.path.join(O, src)
    if not os.path.exists(p) or os.path.getmtime(p) < newer_than:
        return None
    return first_json_line(p).get("library_sha256")


for tagname, key in (("C3", "C3|auto|sorted"), ("C2", "C2|auto|sorted"), ("C4", "C4|auto|sorted"), ("C5", "C5|auto|sorted"),
                     ("C3_shuffle", "C3|auto|shuffle-records")):
    per = pmc_per_dispatch(tagname)
    if not per:
        continue
    # records that agree with each other: a counter pass taken with ANOTHER build of the library than the bench file of the same
    # workload is refused (round 4 committed a C4 bench file quoting 8.91 GB next to a PMC table with 6.08 GB)
    lib_pmc, lib_bench = per.pop("_library_sha256", None), bench_library(tagname)
    if lib_bench is not None and lib_pmc != lib_bench:
        print("REFUSED: the PMC pass of %s ran library %s, its bench file %s -- rerun both on one build" % (tagname, lib_pmc, lib_bench))
        continue
    per = {k: c for k, c in per.items() if not k.startswith("_")}
    json.dump(per, open(os.path.join(P, "%s_%s_pmc_per_dispatch.json" % (tag, tagname if tagname != "C3" else "c3")), "w"), indent=1, sort_keys=True)
    rec = {k: dict(bytes=int((2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024),
                   raw_bytes=int((c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024))
           for k, c in per.items() if "FETCH_SIZE" in c and "WRITE_SIZE" in c}
    if rec:
        rec["_measured"] = tag
        rec["_library_sha256"] = lib_pmc
        traffic[key] = rec
if traffic:
    traffic["_note"] = note
    json.dump(traffic, open(tp, "w"), indent=1, sort_keys=True)
    # the round's bench files looked their `roofline.traffic` up in the table as it was WHEN THEY RAN (the previous round's passes):
    # where this round's pass of the same workload ran the same library, the file quotes this round's figure
    for dst, key in (("%s_c3_bench.json", "C3|auto|sorted"), ("%s_c3_bench_under_rocprof.json", "C3|auto|sorted"), ("%s_C2_bench.json", "C2|auto|sorted"),
                     ("%s_C4_bench.json", "C4|auto|sorted"), ("%s_C5_bench.json", "C5|auto|sorted"), ("%s_c3_bench_shuffled.json", "C3|auto|shuffle-records")):
        bp = os.path.join(P, dst % tag)
        rec = traffic.get(key)
        if not os.path.exists(bp) or not rec or rec.get("_measured") != tag:
            continue
        d = json.load(open(bp))
        r = d.get("roofline") or {}
        if d.get("library_sha256") == rec.get("_library_sha256") and r.get("kernel") in rec:
            r["traffic"] = rec[r["kernel"]]["bytes"]
            r["traffic_source"] = "profiles/pmc_traffic.json: this round's rocprofv3 --pmc passes of the same workload and library build (%s), filled in by scripts/harvest_profiles.py" % rec.get("_library_sha256")
            json.dump(d, open(bp, "w"), indent=1, sort_keys=True)
for src, dst in (("e2e_c3_full.json", "e2e_c3_full.json"), ("fetch_calibration.json", "fetch_calibration.json"),
                 ("e2e_sweep.txt", "%s_e2e_thread_chunk_sweep.txt" % tag)):
    p = os.path.join(O, src)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(P, dst))
for src, dst in (("strong_projection_C3.json", "%s_strong_scaling_projection_C3.json"), ("strong_projection_C4.json", "%s_strong_scaling_projection_C4.json"),
                 ("e2e_c3_full_phred.json", "%s_e2e_c3_full_phred.json"), ("step_timeline.txt", "%s_c3_step_timeline.txt"),
                 ("exchange_ab.json", "%s_exchange_host_side_ab.json")):
    p = os.path.join(O, src)
    if os.path.exists(p) and os.path.getmtime(p) > newer_than:
        shutil.copy(p, os.path.join(P, dst % tag))
p = os.path.join(O, "exp_shuf_global.json")
if os.path.exists(p) and os.path.getmtime(p) > newer_than:
    json.dump(first_json_line(p), open(os.path.join(P, "%s_c3_bench_shuffled_global_counters.json" % tag), "w"), indent=1, sort_keys=True)
mr = {}
for f in sorted(glob.glob(os.path.join(O, "multirank_*.clean.json"))):
    if os.path.getmtime(f) < newer_than:      # (another round's leftovers)
        continue
    d = json.load(open(f))
    mr[os.path.basename(f)[len("multirank_"):-len(".clean.json")]] = {k: d.get(k) for k in (
        "n_gpus", "scaling", "value", "ms_per_step", "fasta_sha256", "config", "other_scaling")}
if mr:
    mr["_note"] = ("bench.py --gpus N with N ranks SHARING the one GPU of the box over gloo: exercises sharded synthesis / routed reads, kd_set_shard "
                   "with shard-local tables and the one fixed-size all-gather; NOT a scaling measurement.  The strong-scaling FASTA must equal the "
                   "single-GPU FASTA of the same config (checked by scripts/gpu_multirank.sh).")
    json.dump(mr, open(os.path.join(P, "%s_multirank_gloo_one_gpu.json" % tag), "w"), indent=1, sort_keys=True)
print("profiles/ refreshed with tag", tag, "from", O)
