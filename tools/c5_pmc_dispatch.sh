#!/bin/bash
# PMC counters PER DISPATCH for the kernels of BASELINE configs[4]'s step whose name contains $1 (default stream16_k<4>): the launches of one
# kernel at one grid size differ by stage, c5_pmc.sh averages them.  bash tools/c5_pmc_dispatch.sh 'stream16_k<4>' [tag]   (GPU box, via gpurun)
set -u
FILTER=${1:-stream16_k<4>}
TAG=${2:-c5pmcd}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --config c5 --steps 3"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_STALL_sum"; do
  i=$((i+1))
  timeout 250 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p --output-format csv -- $CMD > /dev/null 2> $OUT/p$i.err
  f=$(find $OUT/p$i -name "p_counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$FILTER" > $OUT/set$i.txt <<'PY'
import csv, sys, collections
flt = sys.argv[2]
d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if flt not in r["Kernel_Name"]: continue
    d.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(d)
per_step = 3 if len(ids) % 3 == 0 else 1
for j, i in enumerate(ids[-per_step:]):
    print("dispatch %d of the last step:" % j, " ".join("%s=%.4g" % kv for kv in sorted(d[i].items())))
PY
  fi
  rm -rf $OUT/p$i
done
cat $OUT/set*.txt
