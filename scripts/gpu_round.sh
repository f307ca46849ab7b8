#!/bin/bash
# Round driver on one MI355X (through gpurun): `bash scripts/gpu_round.sh STAGE...`, results under gpurun_out/ (scratch; the
# judged summaries are copied into profiles/ by scripts/harvest_profiles.py).  Stages:
#   smoke  tests  test1 (TEST1_ARGS: one selection, output shown)  variants  envruns  psweep  exch  pending  fetch  bench  cfg  shuf  prof  pmc:<CONFIG>[:shuffle]  proj  sweep  timeline  e2e  multirank
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
for st in "$@"; do
  echo "== $st"
  case $st in
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
    tests) timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log ;;
    test1) timeout ${TEST1_TIMEOUT:-600} python -m pytest tests -m gpu -q -x -s -p no:cacheprovider ${TEST1_ARGS:-} > $O/pytest_test1.log 2>&1; echo "test1 rc=$?"; tail -${TEST1_TAIL:-30} $O/pytest_test1.log | cut -c1-400 ;;
    variants) # VARIANTS="name ..." (exp/libkd_<name>.so; "-" = the product), VARIANT_ARGS = bench arguments: k_window / step time of each
          for v in ${VARIANTS:--}; do lib=""; [ "$v" != "-" ] && lib=$R/exp/libkd_$v.so
            KD_BENCH_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --e2e-scale 0 ${VARIANT_ARGS:-} > $O/var_$v.json 2> $O/var_$v.err
            python -c "import json,sys; d=[json.loads(l) for l in open('$O/var_$v.json') if l.startswith('{\"metric')][-1]; print('$v', '%.4f ms'%d['ms_per_step'], {k: x['avg_ms'] for k, x in d['kernels'].items() if k in ('k_window','k_prep','k_cold_lane','k_window_rows','k_long_expand')}, d.get('fasta_sha256','')[:12])" || tail -3 $O/var_$v.err
          done ;;
    envruns) # ENVRUNS="KNOB=value ..." (measurement knobs the engine reads from the environment): the product library under each
          for e in ${ENVRUNS:-}; do
            env $e timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --e2e-scale 0 ${VARIANT_ARGS:-} > $O/env_$e.json 2> $O/env_$e.err
            python -c "import json,sys; d=[json.loads(l) for l in open('$O/env_$e.json') if l.startswith('{\"metric')][-1]; print('$e', '%.4f ms'%d['ms_per_step'], {k: x['avg_ms'] for k, x in d['kernels'].items() if k in ('k_window','k_prep','k_cold_lane')}, d.get('fasta_sha256','')[:12])" || tail -3 $O/env_$e.err
          done ;;
    psweep) # one rank's shard of the strong decomposition under each library variant (VARIANTS) and tuning (SWEEP_TUNINGS)
          for v in ${VARIANTS:--}; do lib=""; [ "$v" != "-" ] && lib=$R/exp/libkd_$v.so
            KD_BENCH_LIB=$lib timeout 300 python scripts/strong_projection.py --config ${PSWEEP_CONFIG:-C3} --ranks ${PSWEEP_RANKS:-8} --only-rank 3 --tunings "${SWEEP_TUNINGS:-0:0}" --out $O/psweep_$v.json > /dev/null 2> $O/psweep_$v.err; echo "$v rc=$?"
            python -c "import json; d=json.load(open('$O/psweep_$v.json')); [print('$v', r['n_ranks'], {t: round(x, 4) for t, x in pr.get('tried', {}).items()}, {k: round(x, 4) for k, x in pr.get('kernels', {}).items() if k in ('k_window', 'k_prep', 'k_cold_lane')}) for r in d['rows'] for pr in r['per_rank']]" || tail -3 $O/psweep_$v.err
          done ;;
    pending) # what round 4 left to be measured first: status words 4 KB apart; k_prep's wavefront times on a 1/8 shard
          [ -f $R/exp/libkd_stride512.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-function -DKDS_STRIDE=512 kindel_amd/csrc/kindel_hip.hip kindel_amd/csrc/kd_decode.cpp -lz -o $R/exp/libkd_stride512.so
          [ -f $R/exp/libkd_phase.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-function -include scripts/exp/kd_phase_clocks.h kindel_amd/csrc/kindel_hip.hip kindel_amd/csrc/kd_decode.cpp -lz -o $R/exp/libkd_phase.so
          VARIANTS="- stride512" bash $R/scripts/gpu_round.sh variants psweep 2>&1 | grep "^- \|^stride512"
          KD_BENCH_LIB=$R/exp/libkd_phase.so timeout 300 python scripts/strong_projection.py --config C3 --ranks 8 --only-rank 3 --steps 5 --warmup 2 --no-profile 2>&1 >/dev/null | grep "k_prep wavefronts\|k_window phase" | tail -2 ;;
    exch) timeout 300 python scripts/exp/exchange_ab.py --out $O/exchange_ab.json 2> $O/exchange_ab.err | cut -c1-600; tail -2 $O/exchange_ab.err ;;
    fetch) # FETCH_SIZE / WRITE_SIZE per launch of the same variants (one rocprofv3 --pmc pass each)
          for v in ${VARIANTS:--}; do lib=""; [ "$v" != "-" ] && lib=$R/exp/libkd_$v.so
            for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/fetch_${v}_$c
              (cd /tmp && KD_BENCH_LIB=$lib timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/fetch_${v}_$c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --e2e-scale 0 ${VARIANT_ARGS:-} > /dev/null 2> $O/fetch_${v}_$c.err)
              python - <<PY
import csv, glob, collections
fs = glob.glob("$O/fetch_${v}_$c/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); ids = collections.defaultdict(set)
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:24]
        if k.startswith("k_"): acc[k] += float(r["Counter_Value"]); ids[k].add(r["Dispatch_Id"])
print("$v $c", {k: "%.4g" % (acc[k] / len(ids[k])) for k in sorted(acc, key=lambda k: -acc[k])[:5]})
PY
              rm -rf $O/fetch_${v}_$c
            done; done ;;
    bench) timeout 900 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"; tail -2 $O/bench_c3.err ;;
    cfg) for c in ${CONFIGS:-C2 C4 C5}; do timeout 600 python bench.py --config $c --steps 10 --warmup 3 --e2e-scale 0 > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"; done ;;
    shuf) timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --shuffle > $O/exp_shuf.json 2> $O/exp_shuf.err; echo "rc=$?"
          KD_SORT_GLOBAL=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --shuffle > $O/exp_shuf_global.json 2> $O/exp_shuf_global.err ;;
    prof) rm -rf $O/prof_c3; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python $R/bench.py --no-cpu-baseline > $O/prof_c3_bench.json 2> $O/prof_c3.err); echo "rc=$?"
          find $O/prof_c3 -type f ! -name "*kernel_stats.csv" -delete ;;      # (the per-dispatch trace is tens of MB: the merge back is capped at 64 MiB)
    pmc:*) spec=${st#pmc:}; cfg=${spec%%:*}; extra=""; tag=$cfg; case $spec in *:shuffle) extra="--shuffle"; tag=${cfg}_shuffle;; esac
           BENCH_ARGS="--config $cfg $extra" PMC_TAG=$tag bash scripts/gpu_pmc.sh 2>&1 | tail -8 ;;
    proj) for c in ${PROJ_CONFIGS:-C3 C4}; do timeout 900 python scripts/strong_projection.py --config $c --tunings "${PROJ_TUNINGS:-0:0}" --out $O/strong_projection_$c.json > /dev/null 2> $O/proj_$c.err; echo "$c rc=$?"; done ;;
    sweep) T="${SWEEP_TUNINGS:-0:0,448:512,448:1024,448:2048,448:4096,256:512,256:1024,256:2048,192:512,192:1024,192:2048,128:512,128:1024,128:4096,64:1024}"
           timeout 600 python scripts/strong_projection.py --config C3 --ranks 8 --only-rank 3 --tunings "$T" --out $O/sweep_C3_rank3of8.json > /dev/null 2> $O/sweep_c3.err; echo "c3/8 rc=$?"
           timeout 600 python scripts/strong_projection.py --config C4 --ranks 8 --only-rank 3 --tunings "$T" --out $O/sweep_C4_rank3of8.json > /dev/null 2> $O/sweep_c4.err; echo "c4/8 rc=$?"
           timeout 600 python scripts/strong_projection.py --config C2 --ranks 1 --tunings "$T" --out $O/sweep_C2.json > /dev/null 2> $O/sweep_c2.err; echo "c2 rc=$?"
           timeout 600 python scripts/strong_projection.py --config C5 --ranks 1 --tunings "$T" --out $O/sweep_C5.json > /dev/null 2> $O/sweep_c5.err; echo "c5 rc=$?" ;;
    sweep_prep) for pp in 4 8 16 32 64; do KD_PREP_PER=$pp timeout 600 python scripts/strong_projection.py --config C3 --ranks 8 --only-rank 3 --out $O/sweep_C3_rank3of8_prep$pp.json > /dev/null 2>> $O/sweep_c3.err; done ;;
    timeline) bash scripts/gpu_timeline.sh 2>&1 | tail -3 ;;
    e2e) timeout 900 python scripts/e2e_bench.py --scale 1.0 --repeat 3 --qual phred --out $O/e2e_c3_full_phred.json > /dev/null 2> $O/e2e.err; tail -c 400 $O/e2e_c3_full_phred.json; echo
         timeout 900 python scripts/e2e_bench.py --scale 1.0 --repeat 3 --out $O/e2e_c3_full.json > /dev/null 2>> $O/e2e.err; tail -c 400 $O/e2e_c3_full.json ;;
    multirank) bash scripts/gpu_multirank.sh 2>&1 | tail -12 ;;
    *) echo "unknown stage $st" ;;
  esac
done
python - <<PY
import json, glob, os
O="$O"
def line(p):
    for l in open(p):
        if l.startswith('{"metric'): return json.loads(l)
for c in ["c3","C2","C4","C5"]:
    p=os.path.join(O,"bench_%s.json"%c)
    if not os.path.exists(p): continue
    try:
        d=line(p); print(c, "%.3e ev/s"%d["value"], "%.4f ms"%d["ms_per_step"], "replay", d.get("replay_ms_per_step"), "kern %.3f"%d["kernel_ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["step_frac"],
                         {k:v["avg_ms"] for k,v in d["kernels"].items()}, (d.get("cpu_baseline") or {}).get("bit_exact_vs_gpu"))
    except Exception as e: print(c, "failed", e)
for p in sorted(glob.glob(os.path.join(O,"strong_projection_*.json"))+glob.glob(os.path.join(O,"sweep_*.json"))):
    try:
        d=json.load(open(p))
        for r in d["rows"]:
            print(os.path.basename(p), r["n_ranks"], r["projected_step_ms"], r.get("projected_speedup"), [(q["rank"], q["step_ms"], q["kernel_ms"], q["k_window_ms"], q["k_prep_ms"], q.get("tuning")) for q in r["per_rank"]][:8])
            if "sweep" in p: print("   tried", r["per_rank"][0].get("tried"), r["per_rank"][0].get("kernels"))
    except Exception as e: print(p, "failed", e)
PY
