#!/bin/bash
# Round 3: bench + LDS conflict counters of k_window for library variants in exp/ (VARIANTS="occ4 nosort_occ4").
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
for v in ${VARIANTS:-occ4 nosort_occ4}; do
  if [ $v = base ]; then unset KD_BENCH_LIB; else export KD_BENCH_LIB=$R/exp/libkd_$v.so; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > $O/r3e_$v.json 2> $O/r3e_$v.err
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/r3e_pmc_$v -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > /dev/null 2> $O/r3e_pmc_$v.err)
  python - <<PY
import json, csv, glob, collections
try:
    d=json.load(open("$O/r3e_$v.json")); print("$v: %.3f ms/step, k_window %.4f ms"%(d["ms_per_step"], d["kernels"]["k_window"]["avg_ms"]))
except Exception as e: print("$v bench failed", e)
fs = glob.glob("$O/r3e_pmc_$v/**/*counter_collection.csv", recursive=True)
if fs:
    acc = collections.defaultdict(float); seen=set()
    for r in csv.DictReader(open(fs[0])):
        if r["Kernel_Name"].split("(")[0].strip().endswith("k_window") or r["Kernel_Name"].startswith("k_window("):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
    n=max(len(seen),1)
    print("   pmc per launch:", {c: "%.4g"%(x/n) for c,x in acc.items()})
    json.dump({c: x/n for c,x in acc.items()}, open("$O/r3e_pmc_$v.json","w"))
PY
done
