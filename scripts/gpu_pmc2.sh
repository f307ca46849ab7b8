#!/bin/bash
# focused PMC passes (counters only with --kernel-trace, as gpurun requires): BENCH_ARGS / KD_BENCH_LIB select the variant
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
R=$PWD
TAG=${TAG:-x}
cd /tmp
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" ; do
  i=$((i+1))
  rm -rf $O/pmc2_${TAG}_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc2_${TAG}_$i -- $BENCH > $O/pmc2_${TAG}_$i.out 2> $O/pmc2_${TAG}_$i.err
  echo "set $i rc=$?"
done
python - <<PY
import csv, glob, collections
for i in (1,2):
    fs = glob.glob("$O/pmc2_${TAG}_%d/**/*counter_collection.csv"%i, recursive=True)
    if not fs: print("set",i,"no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen=set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:40]
        if not k.startswith("k_window"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
    for k in acc:
        print("$TAG", i, k, "dispatches", len(seen), {c: "%.4g"%(v/max(len(seen),1)) for c,v in acc[k].items()})
PY
