#!/bin/bash
# Round 3: instruction counters of k_gpu_inflate alone (scripts/gpu_inflate_proto.py, one launch per quality mode)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; rm -rf $O/pmc_gi; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc_gi -- python $R/scripts/gpu_inflate_proto.py --lib $R/exp/libgi_clocks.so --clocks --no-verify > $O/pmc_gi.out 2> $O/pmc_gi.err
grep -E "clocks" $O/pmc_gi.out
python - <<PY
import csv, glob, collections
fs = glob.glob("$O/pmc_gi/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(fs[0])):
    if "k_gpu_inflate" not in r["Kernel_Name"]: continue
    acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
for d, c in sorted(acc.items(), key=lambda kv: int(kv[0])): print("dispatch", d, {k: "%.3e" % v for k, v in sorted(c.items())})
PY
