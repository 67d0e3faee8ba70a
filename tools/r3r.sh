mkdir -p gpurun_out/r3r; O=$(pwd)/gpurun_out/r3r; root=$(pwd)
export TMPDIR=/tmp
for W in 16 17; do
cd /tmp; rm -rf /tmp/prof_tl && WINDOW=$W rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o tl -- python $root/tools/steps_time.py 20 12 > $O/steps_$W.txt 2>&1
cd $root; python tools/timeline.py /tmp/prof_tl > $O/timeline_$W.txt 2>&1
echo "== window $W"; sed -n 14,30p $O/timeline_$W.txt
done
for rep in 1 2 3; do for W in 16 17; do echo "window $W"; WINDOW=$W python tools/steps_time.py 20 40 2>/dev/null | tail -2; done; done
