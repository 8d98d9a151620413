#!/bin/bash
# Quick per-kernel durations of the headline step (kernel trace only): bash tools/quick_trace.sh <tag> [extra bench args]
TAG=${1:-q}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $ROOT/bench.py --steps 100 --warmup 10 --windows 1 --no-cpu-baseline --no-native-leg --no-dropin --no-secondary --no-traffic --no-dry-leg "$@" > $OUT/bench.json 2> $OUT/err.txt
python - <<PY
import csv, statistics
# forward and backward-chain launches of the whole-encoder kernel separately (they alternate in a step)
rows = [r for r in csv.DictReader(open("$OUT/t_kernel_trace.csv")) if "mlp_split_k" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows][-400:]
print("mlp_split_k forward %.2f us, chain %.2f us (median of the last %d launches each)" % (statistics.median(d[0::2]), statistics.median(d[1::2]), len(d) // 2))
PY
rm -f $OUT/*kernel_trace.csv $OUT/*agent_info.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f} us {r['Percentage']:>6s}%")
PY
tail -c 600 $OUT/bench.json
