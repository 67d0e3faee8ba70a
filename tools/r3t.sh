for rep in 1 2; do for W in 16 17; do for TP in 0 1 -1; do echo "window $W tail_prio $TP"; H2AGG_TAIL_PRIO=$TP WINDOW=$W python tools/steps_time.py 20 40 2>/dev/null | tail -2; done; done; done
