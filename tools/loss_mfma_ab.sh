#!/bin/bash
# A/B of the p = 2 training loss sweeps on the matrix cores (csrc/lp_mfma.hip): library variants x plan settings, timed by tools/loss_train_probe.py
# (graph replay of clica_lp_loss_fwd_train / clica_lp_loss_bwd_sym_train at pool 6 144 and 49 152).
#   T=1 W=3 TAGS="default abl3" bash tools/loss_mfma_ab.sh       (GPU box; variants: make -C cl_ica_amd/csrc variant SRC=lp_mfma.hip TAG=abl3 EXTRA=-DLP2_ABLATE=3;
#   LP2_ABLATE bits: 1 = no MFMAs, 2 = no vector work in the backward block;  CLICA_LP_MFMA=0 in the environment: the VALU sweeps)
cd $GRAFT_REPO_ROOT
export CLICA_LP_MFMA_T=${T:-1} CLICA_LP_MFMA_WG_PER_CU=${W:-3}
for tag in ${TAGS:-default abl3}; do
  if [ "$tag" = "default" ]; then unset CLICA_LIB; else export CLICA_LIB=$PWD/cl_ica_amd/lib/libclica_hip_$tag.so; fi
  echo "== $tag T=$CLICA_LP_MFMA_T W=$CLICA_LP_MFMA_WG_PER_CU"; timeout 300 python tools/loss_train_probe.py one 2>/dev/null | head -2
done
