#!/bin/bash
# Round 3: C5 (long reads) -- per-kernel PMC counters (two passes) next to the bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
CFG=${CFG:-C5}
timeout 600 python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --e2e-scale 0 > $O/r3l_$CFG.json 2> $O/r3l_$CFG.err
P1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES FETCH_SIZE WRITE_SIZE"
(cd /tmp && timeout 600 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $O/r3l_pmc1_$CFG -- python $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-graph --e2e-scale 0 > /dev/null 2> $O/r3l_pmc1.err)
(cd /tmp && timeout 600 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $O/r3l_pmc2_$CFG -- python $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-graph --e2e-scale 0 > /dev/null 2> $O/r3l_pmc2.err)
python - <<PY
import json, csv, glob, collections
d=json.load(open("$O/r3l_$CFG.json")); print("$CFG: %.3f ms/step"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items() if v["avg_ms"]>0.03})
out=collections.defaultdict(dict)
for p in (1,2):
    fs = glob.glob("$O/r3l_pmc%d_$CFG/**/*counter_collection.csv"%p, recursive=True)
    if not fs: print("no pmc", p); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen=collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0].split()[-1]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen[k].add(r["Dispatch_Id"])
    for k in acc:
        for c,x in acc[k].items(): out[k][c]=x/max(len(seen[k]),1)
        out[k]["dispatches_seen"]=len(seen[k])
json.dump(out, open("$O/r3l_pmc_$CFG.json","w"), indent=1)
for k in ("k_window","k_cold_long","k_prep_long","k_sort_count","k_sort_scatter"):
    if k in out: print(k, {c:"%.4g"%v for c,v in out[k].items()})
PY
