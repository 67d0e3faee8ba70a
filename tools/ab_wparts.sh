#!/bin/bash
# A/B on one box: window-sum partial workgroups per window (H2AGG_WPARTS, measure build) for the batched fixed-base instance MSMs
P=halo2-snark-aggregator_amd
cp $P/libh2agg.so /tmp/keep.so; cp tools/libh2agg_measure.so $P/libh2agg.so
for round in 1 2 3; do
  for w in 1 2 4 8; do
    for pr in 4 16; do
      echo "wparts $w proofs $pr: $(H2AGG_WPARTS=$w python tools/agg_leg_phases.py --proofs $pr --reps 30 2>/dev/null | grep -E 'instance_wait|total' | awk '{print $1, $2}' | tr '\n' ' ')"
    done
  done
done
cp /tmp/keep.so $P/libh2agg.so
