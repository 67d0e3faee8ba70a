mkdir -p gpurun_out/r3j; O=gpurun_out/r3j
P=halo2-snark-aggregator_amd
cp $P/libh2agg.so /tmp/base.so
for rep in 1 2 3 4; do
  cp /tmp/base.so $P/libh2agg.so; echo "base"; python tools/steps_time.py 20 40 2>/dev/null | tail -2
  cp tools/libh2agg_sortprio.so $P/libh2agg.so; echo "sortprio"; python tools/steps_time.py 20 40 2>/dev/null | tail -2
done > $O/ab_sortprio.txt 2>&1
cp /tmp/base.so $P/libh2agg.so
cat $O/ab_sortprio.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
