#!/bin/bash
# A/B of the training-path loss kernels between library variants (tools/loss_train_probe.py per variant, alternating):
#   tools/ab_loss.sh default head      (TAG -> cl_ica_amd/lib/libclica_hip_TAG.so; "default" = the product library)
for tag in "$@" "$@"; do
  if [ "$tag" = "default" ]; then unset CLICA_LIB; else export CLICA_LIB=$PWD/cl_ica_amd/lib/libclica_hip_$tag.so; fi
  echo "== $tag"; python tools/loss_train_probe.py one 2>&1 | grep "B="
done
