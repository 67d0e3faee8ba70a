mkdir -p gpurun_out/r3n; O=$(pwd)/gpurun_out/r3n; root=$(pwd)
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/prof_tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o tl -- python $root/tools/steps_time.py 20 12 > $O/steps_under_rocprof.txt 2>&1
cd $root
python tools/timeline.py /tmp/prof_tl > $O/timeline.txt 2>&1
head -60 $O/timeline.txt
