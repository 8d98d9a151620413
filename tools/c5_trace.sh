#!/bin/bash
# Per-kernel durations of BASELINE configs[4] (kitti_masks solver step): bash tools/c5_trace.sh <tag> [f16x2|f32]
TAG=${1:-c5q}; ARITH=${2:-f16x2}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CLICA_CONV_ARITH=$ARITH
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $ROOT/bench.py --config c5 --steps 10 > $OUT/bench.json 2> $OUT/err.txt
python - <<PY
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/t_kernel_trace.csv")):
    d[(r["Kernel_Name"][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("LDS_Block_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:40]:
    v = v[len(v)//3:]
    print("%-62s grid %-8s lds %-7s n %4d  median %8.1f us" % (k[0], k[1], k[2], len(v), sorted(v)[len(v)//2]))
PY
rm -f $OUT/*kernel_trace.csv $OUT/*agent_info.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("arith $ARITH: GPU kernel time %.1f ms" % (tot / 1e6))
for r in rows[:24]:
    print(f"{r['Name'][:80]:80s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f} us {r['Percentage']:>6s}%")
PY
tail -1 $OUT/bench.json | cut -c1-400
