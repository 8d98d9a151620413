#!/bin/bash
# One iteration of BASELINE configs[4] as the ordered list of its launches with durations (which call of a kernel is the slow one):
#   bash tools/c5_step_sequence.sh [tag]      (GPU box, via gpurun) -> gpurun_out/<tag>/sequence.txt
TAG=${1:-c5seq}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $ROOT/bench.py --config c5 --steps 10 > $OUT/bench.json 2> $OUT/err.txt
python - <<PY > $OUT/sequence.txt
import csv
rows = sorted(csv.DictReader(open("$OUT/t_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last full iteration: from the last-but-one launch of the first-stage forward to the last one
idx = [i for i, n in enumerate(names) if "fwd_first16_k" in n or "conv_fwd_image_valu_k" in n]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
tot = 0.0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print("%9.1f us  +%7.1f us  grid %-8s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, r.get("Grid_Size_X", r.get("Grid_Size")), r["Kernel_Name"].split("(")[0].replace("void ", "")[:90]))
print("launches %d, kernel time %.1f us, span %.1f us" % (b - a, tot, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
PY
rm -f $OUT/*kernel_trace.csv $OUT/*agent_info.csv
cat $OUT/sequence.txt
