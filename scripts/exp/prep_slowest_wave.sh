cd /root/repo; O=gpurun_out/prep_clocks; mkdir -p $O
for r in 0 3 7; do
  KD_BENCH_LIB=exp/libkd_phase.so timeout 600 python scripts/strong_projection.py --config C3 --ranks 8 --only-rank $r --steps 3 --warmup 1 --no-profile --out $O/p.json > /dev/null 2> $O/w$r.err
  echo "rank $r of 8: $(grep 'k_prep wavefronts' $O/w$r.err | tail -3 | tr '\n' ' ')"
done
KD_BENCH_LIB=exp/libkd_phase.so timeout 600 python scripts/strong_projection.py --config C4 --ranks 8 --only-rank 3 --steps 3 --warmup 1 --no-profile --out $O/p.json > /dev/null 2> $O/c4.err
echo "C4 rank 3 of 8: $(grep 'k_prep wavefronts' $O/c4.err | tail -2 | tr '\n' ' ')"
