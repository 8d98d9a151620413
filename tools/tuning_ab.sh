#!/bin/bash
# A/B of the headline step with one clica_set_tuning key off / on (two runs each, interleaved): bash tools/tuning_ab.sh <key> [outdir]
key=$1; out=${2:-gpurun_out/ab_$1}
mkdir -p $out
FLAGS="--steps 200 --warmup 20 --no-cpu-baseline --no-dropin --no-native-leg --no-secondary --no-traffic --no-dry-leg"
for rep in 1 2; do
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
    print(sys.argv[1], 'steps/s %.1f' % d['value'], 'ms %.4f' % d['ms_per_step'],
          ' | '.join('%s %.1f' % (k['op'], k.get('in_step_us') or k['avg_us']) for k in d.get('kernels', [])),
          'loss %.1f+%.1f' % (d['loss_kernel']['fwd_us'], d['loss_kernel']['bwd_us']) if 'loss_kernel' in d else '', 'final_loss %.6f' % d['final_loss'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
done
