for tag in default head default head; do
  if [ "$tag" = "default" ]; then unset CLICA_LIB; else export CLICA_LIB=$PWD/cl_ica_amd/lib/libclica_hip_$tag.so; fi
  echo "== $tag"; python tools/loss_train_probe.py one 2>&1 | grep "n=40\|B3=49152"
done
