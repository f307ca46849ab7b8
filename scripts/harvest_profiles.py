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
        for r in rows[:1] + [r for r in rows[1:] if r and r[0].startswith("k_")]:
            w.writerow([r[0].split("(")[0]] + r[1:])
for src, dst in (("bench_c3.json", "%s_c3_bench.json"), ("prof_c3_bench.json", "%s_c3_bench_under_rocprof.json"),
                 ("bench_C2.json", "%s_C2_bench.json"), ("bench_C4.json", "%s_C4_bench.json"), ("bench_C5.json", "%s_C5_bench.json")):
    p = os.path.join(O, src)
    if os.path.exists(p):
        json.dump(first_json_line(p), open(os.path.join(P, dst % tag), "w"), indent=1, sort_keys=True)

acc = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for i in (1, 2, 3, 4):
    for f in sorted(glob.glob(os.path.join(O, "pmc_%d" % i, "**", "*counter_collection.csv"), recursive=True),
                    key=os.path.getmtime, reverse=True)[:1]:   # the newest run only (gpurun merges, it does not delete)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            k = k.split("<")[0] if k.startswith("void ") is False else k
            if not k.startswith("k_"):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
per = {k: {c: v / max(len(ndisp[(k, c)]), 1) for c, v in cs.items()} for k, cs in acc.items()}
if per:
    json.dump(per, open(os.path.join(P, "%s_c3_pmc_per_dispatch.json" % tag), "w"), indent=1, sort_keys=True)
    note = ("(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch: gfx950 FETCH_SIZE halving correction of MI355X_MICROARCH.md applied; "
            "raw_bytes = uncorrected")
    traffic = {k: dict(bytes=int((2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024),
                       raw_bytes=int((c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024), note=note)
               for k, c in per.items() if "FETCH_SIZE" in c}
    json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print("profiles/ refreshed with tag", tag, "from", O)
