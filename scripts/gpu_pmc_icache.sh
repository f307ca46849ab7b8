#!/bin/bash
# instruction-cache counters of k_window (counters only with --kernel-trace, as gpurun requires)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out; R=$PWD
cd /tmp
rm -rf $O/pmc_ic
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_ic -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $O/pmc_ic.out 2> $O/pmc_ic.err
echo "rc=$?"
python - <<PY
import csv, glob, collections
fs = glob.glob("$O/pmc_ic/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen=collections.defaultdict(set)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0][:40]
    if not k.startswith("k_"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen[k].add(r["Dispatch_Id"])
for k in ("k_window","k_prep","k_cold_lane"):
    if k in acc: print(k, {c: "%.4g"%(v/len(seen[k])) for c,v in acc[k].items()})
PY
