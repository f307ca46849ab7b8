#!/bin/bash
# Attribute k_window's HBM traffic to its sources: FETCH_SIZE / WRITE_SIZE per launch of the product library and of the
# measurement-only builds in exp/ (kd_window.h: -DKD_EXP_NOSEQ, -DKD_EXP_NOFLUSH).  Counters only with --kernel-trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib="$R/exp/libkd_$v.so"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/attr_${v}_$ctr
    (cd /tmp && KD_BENCH_LIB=$lib timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/attr_${v}_$ctr -- \
        python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $O/attr_${v}_$ctr.out 2> $O/attr_${v}_$ctr.err) || echo "$v $ctr rc=$?"
  done
  python - <<PY
import csv, glob, collections, json
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("$O/attr_${v}_%s/**/*counter_collection.csv" % ctr, recursive=True)
    if not fs: continue
    acc = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if r["Counter_Name"] != ctr: continue
        acc[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k in acc: res.setdefault(k, {})[ctr] = acc[k] / max(len(disp[k]), 1) * 1024
ms = json.load(open("$O/attr_${v}_FETCH_SIZE.out"))["ms_per_step"] if glob.glob("$O/attr_${v}_FETCH_SIZE.out") else None
for k in ("k_window", "k_prep", "k_cold_lane"):
    if k in res: print("$v", k, "FETCH raw %.3f GB  WRITE %.3f GB" % (res[k].get("FETCH_SIZE", 0) / 1e9, res[k].get("WRITE_SIZE", 0) / 1e9))
json.dump(res, open("$O/attrib_$v.json", "w"))
PY
done
