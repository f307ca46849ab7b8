#!/bin/bash
# Round-2 measurement driver (one gpurun call): parity suite, bench lines for library variants, rocprofv3 stats, PMC.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
run() { tag=$1; shift; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $O/exp_$tag.json 2> $O/exp_$tag.err || tail -3 $O/exp_$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/exp_$tag.json")); k=d["kernels"]
    print("$tag", "%.3f ms"%d["ms_per_step"], "ksum %.3f"%d["kernel_ms_per_step"], "%.3g ev/s"%d["value"], d["fasta_sha256"][:8], " ".join("%s=%.3f"%(n[2:],k[n]["avg_ms"]) for n in sorted(k, key=lambda n:-k[n]["avg_ms"]*k[n]["launches_per_step"])[:8]))
except Exception as e: print("$tag failed", e)
PY
}
for a in "$@"; do
  case $a in
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1;;
    tests) timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -20;;
    base) run base;;
    full) timeout 900 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench_c3.json')); print('C3 %.3f ms %.3g ev/s frac %s cpu %s'%(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('cpu_baseline')))";;
    win) run win --mode window;;
    plain) run plain --synth clip_p=0.0 --synth indel_p=0.0 --synth planted=0;;
    winplain) run winplain --mode window --synth clip_p=0.0 --synth indel_p=0.0 --synth planted=0;;
    c2|c4|c5) C=$(echo $a | tr a-z A-Z); timeout 900 python bench.py --config $C --steps 3 --warmup 1 > $O/bench_$C.json 2> $O/bench_$C.err; python -c "
import json; d=json.load(open('$O/bench_$C.json')); k=d['kernels']; print('$C %.3f ms %.3g ev/s exact=%s'%(d['ms_per_step'], d['value'], d.get('cpu_baseline',{}).get('bit_exact_vs_gpu')), ' '.join('%s=%.3f'%(n[2:],k[n]['avg_ms']) for n in sorted(k, key=lambda n:-k[n]['avg_ms']*k[n]['launches_per_step'])[:6]))";;
    shuf) run shuf --shuffle;;
    lib:*) L=${a#lib:}; KD_BENCH_LIB=exp/libkd_$L.so run lib_$L;;
    prof) rm -rf $O/prof_c3; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python $R/bench.py --no-cpu-baseline > $O/prof_c3_bench.json 2> $O/prof_c3.err); echo "prof rc=$?"; f=$(find $O/prof_c3 -name "*kernel_stats.csv" | head -1); head -12 $f;;
    pmc) rm -rf $O/pmc_*; bash scripts/gpu_pmc.sh > $O/pmc_summary.txt 2>&1; grep -E "^[1-4] k_(strip|window|prep|cold|cns)" $O/pmc_summary.txt;;
  esac
done
