#!/bin/bash
# A/B of the config-3 leg (bench.py `secondary`) between library variants: bash tools/c3_lib_ab.sh <outdir> default <tag> ...   (REPS=n)
out=$1; shift
mkdir -p $out
FLAGS="--steps 100 --warmup 10 --windows 3 --no-conv-configs --no-native-leg --no-dropin --no-dry-leg --no-traffic --no-cpu-baseline"
for rep in $(seq 1 ${REPS:-2}); do
for tag in "$@"; do
  if [ "$tag" = "default" ]; then unset CLICA_LIB; else export CLICA_LIB=$PWD/cl_ica_amd/lib/libclica_hip_$tag.so; fi
  python bench.py $FLAGS 2>$out/err_$tag.txt | tail -1 > $out/c3_${tag}_$rep.json
  python - "$tag" $out/c3_${tag}_$rep.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); s = d["secondary"]
    print(sys.argv[1], 'headline %.1f' % d['value'], {k: round(s[k]["value"], 1) for k in ("pool_6144", "pool_49152_emulated_8_ranks")},
          ' | '.join('%s x%s %.1f' % (r.get("op"), r.get("launches_per_step"), r.get("avg_us")) for r in s["pool_6144"].get("kernels", [])[:3]))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
done
unset CLICA_LIB
