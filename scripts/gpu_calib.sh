#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (scripts/fetch_calib.hip) -> gpurun_out/fetch_calibration.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/calib_*
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/calib_$c -- $R/exp/fetch_calib > $O/calib_$c.out 2> $O/calib_$c.err; echo "calib $c rc=$?"
done
python - <<PY
import csv, glob, json, collections
N = 2**31
out = {"bytes_touched_once": N, "note": "factor = true bytes / (counter x 1024): multiply a kernel's counter by 1024 x factor to get bytes; "
       "2 GiB buffer read or written exactly once per launch (beyond the 256 MB Infinity Cache), average of 3 launches", "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("$O/calib_%s/**/*counter_collection.csv" % c, recursive=True)
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != c: continue
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        m = sum(v) / len(v)
        out["kernels"].setdefault(k, {})[c] = m
        out["kernels"][k][c + "_bytes_x1024"] = m * 1024
        out["kernels"][k][c + "_factor"] = (N / (m * 1024)) if m else None
json.dump(out, open("$O/fetch_calibration.json", "w"), indent=1)
for k, v in out["kernels"].items(): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
