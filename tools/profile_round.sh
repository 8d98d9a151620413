#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the bench command + separate PMC passes.
# usage (via gpurun): bash tools/profile_round.sh r1
set -u
TAG=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PARTS=${PARTS:-"head c3"}
if [[ "$PARTS" == *head* ]]; then
CMD="python $ROOT/bench.py --steps 100 --warmup 10 --windows 1 --no-cpu-baseline --no-native-leg --no-dropin --no-secondary --no-traffic --no-dry-leg"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
CMD2="python $ROOT/bench.py --steps 10 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline --no-native-leg --no-dropin --no-secondary --no-traffic --no-dry-leg"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- $CMD2 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p --output-format csv -- $CMD2 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT/pmc_sq -o p --output-format csv -- $CMD2 > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS -d $OUT/pmc_lds -o p --output-format csv -- $CMD2 > /dev/null 2> $OUT/pmc_lds.err
# vector-ALU occupancy of the pair sweeps: busy / wave cycles, instruction mix
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD -d $OUT/pmc_valu -o p --output-format csv -- $CMD2 > /dev/null 2> $OUT/pmc_valu.err
# L2 hit rate (per-XCD L2: do the tiles that share an operand find it there?)
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $OUT/pmc_l2 -o p --output-format csv -- $CMD2 > /dev/null 2> $OUT/pmc_l2.err
# the native fp32-MFMA mode (bench.py --native-fp32), kernel trace only
rocprofv3 --kernel-trace --stats -d $OUT/trace_native -o t --output-format csv -- python $ROOT/bench.py --steps 100 --warmup 10 --windows 1 --no-cpu-baseline --no-roofline --no-dropin --no-secondary --no-traffic --native-fp32 > $OUT/bench_native_under_rocprof.json 2> $OUT/trace_native.err
rm -f $OUT/trace_native/*agent_info.csv $OUT/trace_native/*kernel_trace.csv
fi
if [[ "$PARTS" == *c3* ]]; then
# BASELINE config 3 on one rank (n = 40, sphere, p = 1: the per-layer fp32-MFMA kernels): kernel trace + HBM-side traffic
C3="python $ROOT/bench.py --n 40 --space-type sphere --p 1 --steps 20 --warmup 5 --windows 1 --no-cpu-baseline --no-native-leg --no-dropin --no-secondary --no-traffic --no-dry-leg"
rocprofv3 --kernel-trace --stats -d $OUT/trace_c3 -o t --output-format csv -- $C3 > $OUT/bench_c3_under_rocprof.json 2> $OUT/trace_c3.err
C3B="python $ROOT/bench.py --n 40 --space-type sphere --p 1 --steps 4 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline --no-native-leg --no-dropin --no-secondary --no-traffic --no-dry-leg"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_c3_fetch -o p --output-format csv -- $C3B > /dev/null 2> $OUT/pmc_c3_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_c3_write -o p --output-format csv -- $C3B > /dev/null 2> $OUT/pmc_c3_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_c3_sq -o p --output-format csv -- $C3B > /dev/null 2> $OUT/pmc_c3_sq.err
rm -f $OUT/trace_c3/*agent_info.csv $OUT/trace_c3/*kernel_trace.csv $OUT/pmc_c3_*/*agent_info.csv $OUT/pmc_c3_*/*kernel_trace.csv
fi
ls -la $OUT $OUT/trace | head -30
# keep only the small summaries (trace CSV of every dispatch can be large)
rm -f $OUT/trace/*agent_info.csv
du -sh $OUT
