#!/bin/bash
# Which leg of bench.py fails?  (debug helper: each leg as a run of its own, return codes only)
B="python bench.py --steps 10 --warmup 3 --windows 1 --no-cpu-baseline --no-traffic --no-secondary --no-dropin --no-dry-leg --no-native-leg"
$B --n 40 --space-type sphere --p 1 > /dev/null 2> gpurun_out/leg_c3.err; echo "c3 rc=$?"
$B --config c4 > /dev/null 2> gpurun_out/leg_c4.err; echo "c4 rc=$?"
$B --config c5 > /dev/null 2> gpurun_out/leg_c5.err; echo "c5 rc=$?"
for f in gpurun_out/leg_c?.err; do echo "== $f"; tail -n 3 $f; done
