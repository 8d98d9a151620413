#!/bin/bash
# A/B of the config-3 leg (bench.py `secondary`): generic vs specialised f16x2 epilogue of gemm_split_k (clica_set_tuning "gemm16_epilogue").
# usage (GPU box): bash tools/c3_ab.sh  ->  gpurun_out/c3_ab_{generic,specialised}.json
mkdir -p gpurun_out
FLAGS="--steps 100 --warmup 10 --windows 3 --no-conv-configs --no-native-leg --no-dropin --no-dry-leg --no-traffic --no-cpu-baseline"
for mode in 0 1; do
  name=$([ $mode = 0 ] && echo generic || echo specialised)
  python - $FLAGS > gpurun_out/c3_ab_$name.json 2> gpurun_out/c3_ab_$name.err <<PY
import sys
from cl_ica_amd import _lib
_lib.check(_lib.load().clica_set_tuning(b"gemm16_epilogue", $mode), "clica_set_tuning")
sys.argv = ["bench.py"] + sys.argv[1:]
import bench
bench.main()
PY
  python - <<PY
import json
d = json.loads(open("gpurun_out/c3_ab_$name.json").read().strip().splitlines()[-1])
s = d["secondary"]
print("$name", {k: round(s[k]["value"], 1) for k in ("pool_6144", "pool_49152_emulated_8_ranks")}, s["pool_6144"].get("roofline", {}).get("avg_launch_us"), s["pool_6144"].get("roofline", {}).get("frac_issued"))
for r in s["pool_6144"].get("kernels", []):
    print("   ", r.get("op"), r.get("kernel", "")[:60], r.get("launches_per_step"), r.get("avg_us"))
PY
done
