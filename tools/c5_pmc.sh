#!/bin/bash
# PMC counters of the kernels of BASELINE configs[4]'s step (separate passes, kernel trace only):  bash tools/c5_pmc.sh <tag>
set -u
TAG=${1:-c5pmc}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --config c5 --steps 3"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 250 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p --output-format csv -- $CMD > /dev/null 2> $OUT/p$i.err
  f=$(find $OUT/p$i -name "p_counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" > $OUT/set$i.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0][-60:] + " g" + r.get("Grid_Size", "")
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (n, r["Dispatch_Id"])
    if key not in seen: seen.add(key); cnt[n] += 1
for n in acc:
    if "conv" not in n: continue
    print(n, "launches", cnt[n], " ".join("%s=%.4g" % (k, v / cnt[n]) for k, v in sorted(acc[n].items())))
PY
  fi
  rm -rf $OUT/p$i
done
cat $OUT/set*.txt
