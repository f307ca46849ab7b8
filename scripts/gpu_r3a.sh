#!/bin/bash
# Round 3, first call: instruction issue calibration + the VALU / LDS activity counters of k_window (bench C3).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
echo "== issue calibration"; timeout 300 exp/issue_calib > $O/valu_issue_calibration.json 2> $O/issue_calib.err; echo "rc=$?"; tail -3 $O/valu_issue_calibration.json
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcv_$i -- $BENCH > $O/pmcv_$i.out 2> $O/pmcv_$i.err
  echo "set $i rc=$?"
done
python - <<PY
import csv, glob, collections, json
res = {}
for i in (1,2):
    fs = glob.glob("$O/pmcv_%d/**/*counter_collection.csv"%i, recursive=True)
    if not fs: print("set",i,"no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen=set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:40]
        if not k.startswith("k_"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen.add((k, r["Dispatch_Id"]))
    cnt = collections.Counter(k for k,_ in seen)
    for k in acc:
        res.setdefault(k, {}).update({c: v/max(cnt[k],1) for c,v in acc[k].items()})
for k in ("k_window","k_prep","k_cold_lane"):
    print(k, {c: "%.4g"%v for c,v in res.get(k,{}).items()})
json.dump(res, open("$O/pmc_valu_activity.json","w"), indent=1)
PY
