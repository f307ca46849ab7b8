#!/bin/bash
# one bench step as a dispatch timeline (rocprofv3 --kernel-trace): kernel, start offset, duration, idle gap before it
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/tl
# TL_CMD: another command whose last step is wanted (e.g. a 1/8 shard: "scripts/strong_projection.py --ranks 8 --only-rank 3 --steps 3 --warmup 1 --no-profile"), TL_OUT its file
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl -- python $R/${TL_CMD:-bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} --no-graph --e2e-scale 0} > $O/tl_bench.json 2> $O/tl.err)
python - <<PY | tee $O/${TL_OUT:-step_timeline.txt}
import csv, glob
rows=[]
for f in glob.glob("$O/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]))
for f in glob.glob("$O/tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction","")[:30]))
rows.sort()
# last step = from the last k_prep dispatch on
idx=[i for i,r in enumerate(rows) if r[2].startswith("k_prep")]
start=idx[-1]
# include what precedes k_prep in that step (memsets) back to the previous step's last copy
j=start
while j>0 and not rows[j-1][2].startswith("COPY DEVICE_TO_HOST") and start-j<12: j-=1
t0=rows[j][0]; prev=None; tot_busy=0
for s,e,n in rows[j:]:
    gap=(s-prev)/1e3 if prev else 0.0
    print("%-46s +%8.1f us  dur %8.1f us  gap %7.1f us"%(n,(s-t0)/1e3,(e-s)/1e3,gap))
    prev=max(prev or e,e); tot_busy+=(e-s)/1e3
print("span %.1f us busy %.1f us"%((prev-t0)/1e3,tot_busy))
PY
rm -rf $O/tl
