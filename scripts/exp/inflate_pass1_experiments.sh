#!/bin/bash
# EXPERIMENT (round 6): what bounds k_inflate_tokens at full occupancy?  (68 568 blocks: 61 ms; a quarter of them, one wavefront per
# CU: 30 ms.)  Resident wavefronts per CU 1 / 2 / 4, with and without the literal stores; then counters of the standard build.
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R; O=gpurun_out/inflate_exp; mkdir -p $O
for lib in "" _nostore; do for wpc in 1 2 4; do
  echo "lib=proto$lib GI2_WPC=$wpc: $(GI2_WPC=$wpc timeout 600 python scripts/gpu_inflate_proto.py --config C3 --scale 1.0 --no-verify --two-pass --lib exp/libgpu_inflate_proto$lib.so 2>&1 | grep "^phred" | cut -c1-120)"
done; done
export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40); rm -rf $O/pmc_$tag
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$tag -- python $R/scripts/gpu_inflate_proto.py --config C3 --scale 1.0 --no-verify --two-pass > /dev/null 2> $R/$O/pmc_$tag.err)
  python - "$O/pmc_$tag" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:24]
        if "inflate" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(n[k]), {c: "%.4g" % (v / max(1, len(n[k]))) for c, v in acc[k].items()})
PY
done
