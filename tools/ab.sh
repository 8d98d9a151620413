#!/bin/bash
# A/B helper: run bench.py (headline config, no side legs) once per library variant and print step rate + in-step kernel times.
#   tools/ab.sh default spread0 ...      (variant TAG -> cl_ica_amd/lib/libclica_hip_TAG.so; "default" = the product library)
for tag in "$@"; do
  if [ "$tag" = "default" ]; then unset CLICA_LIB; else export CLICA_LIB=$PWD/cl_ica_amd/lib/libclica_hip_$tag.so; fi
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-dropin --no-native-leg $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', 'steps/s %.1f' % d['value'], 'ms %.4f' % d['ms_per_step'], ' | '.join('%s %.1f/%.1f' % (k['op'], k.get('in_step_us') or 0, k['avg_us']) for k in d.get('kernels', [])), 'loss %.1f+%.1f' % (d['loss_kernel']['fwd_us'], d['loss_kernel']['bwd_us']) if 'loss_kernel' in d else '')
"
done
