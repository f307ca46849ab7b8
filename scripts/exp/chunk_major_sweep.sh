cd /root/repo
for lib in "" exp/libkd_occ5_V.so exp/libkd_occ4_V.so exp/libkd_occ4.so; do
  echo "== lib ${lib:-product} =="
  env KD_BENCH_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --e2e-scale 0 --sweep window:448:0,window:512:0,window:576:0,window:448:0,window:512:0,window:576:0 2>&1 >/dev/null | grep '"sweep"' | python -c "
import json,sys
print(' | '.join('%s %.4f' % (json.loads(l)['sweep'].split(':')[1], json.loads(l)['ms_per_step']) for l in sys.stdin))"
done
bash scripts/exp/lib_ab.sh 2 C3 - exp/libkd_occ5_V.so
