#!/bin/bash
# Soak of the stand-in based multi-rank tests on a GPU box: N fresh processes of
# tests/rccl_stub_ranks.py per configuration, tallied.   usage: tools/soak_standins.sh <N threads-mode> <N procs-mode> [out]
set -u
NT=${1:-100}; NP=${2:-30}; OUT=${3:-gpurun_out/soak_standins.txt}
mkdir -p "$(dirname "$OUT")" /tmp/soak
hipcc -O1 -shared -fPIC -Wl,-soname,librccl.so.1 tests/cpp/rccl_stub.cpp -o /tmp/soak/librccl.so.1 -lrt || exit 1
: > "$OUT"
run() {  # world mode how n loops
  local ok=0 bad=0 t0=$(date +%s)
  for i in $(seq 1 "$4"); do
    if timeout 600 python tests/rccl_stub_ranks.py /tmp/soak/librccl.so.1 "$1" "$2" "$3" "$5" > /tmp/soak/last.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); echo "--- FAILED world $1 $2 $3 run $i" >> "$OUT"; tail -5 /tmp/soak/last.log >> "$OUT"; fi
  done
  echo "world $1 mode $2 ranks-as-$3: $ok passed, $bad failed of $4 fresh processes x $5 sharded calls each ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT"
}
run 2 ok threads "$NT" 3
run 3 ok threads $((NT / 4)) 3
run 2 ok procs "$NP" 3
run 3 ok procs $((NP / 3)) 3
run 2 fail3 threads 5 1
run 3 fail4 procs 5 1
