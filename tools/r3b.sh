mkdir -p gpurun_out/r3b; O=gpurun_out/r3b
./tools/ubench_waves > $O/ubench_waves.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
P=halo2-snark-aggregator_amd
cp $P/libh2agg.so /tmp/base.so
for lv in 2 3; do echo "base LEVEL=$lv"; LEVEL=$lv python tools/steps_time.py 20 40; done > $O/steps_base.txt 2>&1
cp tools/libh2agg_sortprio.so $P/libh2agg.so
for lv in 2 3; do echo "sortprio LEVEL=$lv"; LEVEL=$lv python tools/steps_time.py 20 40; done > $O/steps_sortprio.txt 2>&1
cp /tmp/base.so $P/libh2agg.so
cat $O/ubench_waves.txt $O/steps_base.txt $O/steps_sortprio.txt; tail -c 300 $O/bench_default.json
