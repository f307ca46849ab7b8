#!/bin/bash
# PMC passes on bench C3 (counters only with --kernel-trace, as gpurun requires)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
R=$PWD
cd /tmp
true
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph ${BENCH_ARGS:-}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_${PMC_TAG:-C3}_$i -- $BENCH > $O/pmc_${PMC_TAG:-C3}_$i.out 2> $O/pmc_${PMC_TAG:-C3}_$i.err
  echo "set $i rc=$?"
done
python - <<PY
import csv, glob, collections
for i in (1,2,3,4):
    fs = glob.glob("$O/pmc_${PMC_TAG:-C3}_%d/**/*counter_collection.csv"%i, recursive=True)
    if not fs: print("set",i,"no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:40]
        if not (k.startswith("k_") or "k_pileup" in k): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    seen=set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:40]
        if (k.startswith("k_") or "k_pileup" in k): seen.add((k, r["Dispatch_Id"]))
    for k,_ in seen: cnt[k]+=1
    for k in sorted(acc, key=lambda k:-sum(acc[k].values()))[:4]:
        print(i, k, "dispatches", cnt[k], {c: "%.4g"%(v/max(cnt[k],1)) for c,v in acc[k].items()})
PY
