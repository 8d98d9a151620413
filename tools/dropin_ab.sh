#!/bin/bash
# A/B of bench.py's `dropin` legs (the reference's own train_step on the drop-in modules: eager / captured) with one clica_set_tuning key off / on:
#   bash tools/dropin_ab.sh <key> [outdir]      (REPS=n)
key=$1; out=${2:-gpurun_out/dropin_ab_$1}
mkdir -p $out
FLAGS="--steps 100 --warmup 10 --windows 1 --no-cpu-baseline --no-native-leg --no-secondary --no-traffic --no-dry-leg --no-roofline"
for rep in $(seq 1 ${REPS:-2}); do
for mode in 0 1; do
  python - $FLAGS > $out/bench_${mode}_$rep.json 2> $out/err_$mode.txt <<PY
import sys
from cl_ica_amd import _lib
_lib.check(_lib.load().clica_set_tuning(b"$key", $mode), "clica_set_tuning")
sys.argv = ["bench.py"] + sys.argv[1:]
import bench
bench.main()
PY
  python - "$key=$mode" $out/bench_${mode}_$rep.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'engine %.1f' % d['value'], {k: round(v["value"], 1) for k, v in d["dropin"].items() if isinstance(v, dict) and "value" in v})
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
done
