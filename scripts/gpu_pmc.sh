#!/bin/bash
# PMC passes on one bench workload (BENCH_ARGS, tag PMC_TAG; counters only with --kernel-trace, as gpurun requires)
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
  case " ${PMC_SETS:-1 2 3 4} " in *" $i "*) ;; *) continue ;; esac      # PMC_SETS="3 4": the traffic passes only
  rm -rf $O/pmc_${PMC_TAG:-C3}_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_${PMC_TAG:-C3}_$i -- $BENCH > $O/pmc_${PMC_TAG:-C3}_$i.out 2> $O/pmc_${PMC_TAG:-C3}_$i.err
  echo "set $i rc=$?"
done
python - <<PY
# counters of every kd kernel averaged per launch -> gpurun_out/pmc_<tag>_summary.json; the raw per-dispatch CSVs (tens of MB: the
# merge back from the GPU box is capped at 64 MiB) are dropped
import csv, glob, collections, json, os, shutil
tag = "${PMC_TAG:-C3}"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); ids = collections.defaultdict(set)
for i in [int(x) for x in "${PMC_SETS:-1 2 3 4}".split()]:
    d = "$O/pmc_%s_%d" % (tag, i)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            k = k[5:] if k.startswith("void ") else k
            k = {"k_window<false>": "k_window", "k_window<true>": "k_window_rows"}.get(k, k.split("<")[0])
            if not k.startswith("k_"): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); ids[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    shutil.rmtree(d, ignore_errors=True)
per = {k: {c: v / max(len(ids[(k, c)]), 1) for c, v in cs.items()} for k, cs in acc.items()}
out = "$O/pmc_%s_summary.json" % tag
old = json.load(open(out)) if os.path.exists(out) else {}
for k, cs in per.items(): old.setdefault(k, {}).update(cs)        # (a traffic-only rerun keeps the other counters)
# the library these passes ran (bench.py prints it on its JSON line): harvest_profiles.py compares it with the bench files'
try:
    lines = [l for i in [int(x) for x in "${PMC_SETS:-1 2 3 4}".split()] for l in open("$O/pmc_%s_%d.out" % (tag, i)) if l.startswith('{"metric')]
    libs = sorted(set(json.loads(l).get("library_sha256") for l in lines))
    old["_library_sha256"] = libs[0] if len(libs) == 1 else libs
except Exception as e:
    old["_library_sha256"] = None
json.dump(old, open(out, "w"), indent=1, sort_keys=True)
for k in sorted((k for k in per if not k.startswith("_")), key=lambda k: -per[k].get("FETCH_SIZE", per[k].get("SQ_BUSY_CYCLES", 0)))[:4]:
    print(tag, k, {c: "%.4g" % v for c, v in sorted(per[k].items())})
PY
