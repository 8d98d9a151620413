#!/bin/bash
# Round-6 A/B on one box: product library vs variants (CLICA_LIB), headline step rate + in-step encoder kernel times, then the phase trace.
#   bash tools/r6_ab.sh <outdir> <tag> [<tag> ...]     tag "default" = the product library, otherwise cl_ica_amd/lib/libclica_hip_<tag>.so
out=$1; shift
mkdir -p $out
for rep in $(seq 1 ${REPS:-2}); do
for tag in "$@"; do
  if [ "$tag" = "default" ]; then unset CLICA_LIB; else export CLICA_LIB=$PWD/cl_ica_amd/lib/libclica_hip_$tag.so; fi
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dropin --no-native-leg --no-secondary --no-traffic --no-dry-leg 2>$out/err_$tag.txt | tail -1 > $out/bench_${tag}_$rep.json
  python - "$tag" $out/bench_${tag}_$rep.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], 'steps/s %.1f' % d['value'], 'ms %.4f' % d['ms_per_step'],
          ' | '.join('%s %.1f' % (k['op'], k.get('in_step_us') or k['avg_us']) for k in d.get('kernels', [])),
          'loss %.1f+%.1f' % (d['loss_kernel']['fwd_us'], d['loss_kernel']['bwd_us']) if 'loss_kernel' in d else '', 'final_loss %.6f' % d['final_loss'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
done
unset CLICA_LIB
