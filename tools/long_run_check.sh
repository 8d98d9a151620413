#!/bin/bash
# End-to-end check of the science in each encoder arithmetic: the reference's default unsupervised phase (shortened: 30 001 steps) through
# the CLI, final disentanglement scores, wall time, the engine's own record (scale flags, loss-guard counters).
for ar in f16 bf16; do
  echo "== CLICA_SPLIT_ARITH=$ar"
  ( time CLICA_SPLIT_ARITH=$ar python -m cl_ica_amd.train_mlp --n 10 --p 2 --batch-size 6144 --n-steps ${STEPS:-30001} --only-unsupervised --n-log-steps 5000 --seed 0 2>&1 | grep -E "Step: |engine:|linear mean|perm mean|note" ) 2>&1 | tail -14
done
