mkdir -p gpurun_out/r3k; O=gpurun_out/r3k
for rep in 1 2 3; do
  echo "defer on"; python tools/steps_time.py 20 40 2>/dev/null | tail -2
  echo "defer off"; H2AGG_DEFER_TAILS=0 python tools/steps_time.py 20 40 2>/dev/null | tail -2
done > $O/ab_defer.txt 2>&1
cat $O/ab_defer.txt
for sl in 2 3 4 6 8; do echo "pcie slices $sl"; H2AGG_PCIE_SLICES=$sl python tools/pcie_rate.py 2>/dev/null | tail -3; done > $O/pcie_slices.txt 2>&1; cat $O/pcie_slices.txt
