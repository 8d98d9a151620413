#!/bin/bash
# Kernel-trace stats of BASELINE configs[3] / [4] on one GPU:  bash tools/profile_c45.sh  (GPU box, via gpurun)
# -> gpurun_out/profile_r4_c4|c5/{t_kernel_stats.csv, bench.json}; condensed into profiles/ by tools/conv_share_summary.py
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in ${@:-c4 c5}; do
  OUT=$ROOT/gpurun_out/profile_${PTAG:-r4}_$c
  mkdir -p $OUT
  # un-profiled first run: MIOpen benchmarks its candidate kernels the first time it sees a convolution on a box (hundreds of thousands
  # of launches that would drown the trace); its user find-db remembers the choice for the profiled run
  python $ROOT/bench.py --config $c --steps 3 > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python $ROOT/bench.py --config $c --steps 10 > $OUT/bench.json 2> $OUT/trace.err
  find $OUT/trace -name "t_kernel_stats.csv" -exec cp {} $OUT/t_kernel_stats.csv \;
  rm -rf $OUT/trace
  tail -1 $OUT/bench.json | cut -c1-300
done
