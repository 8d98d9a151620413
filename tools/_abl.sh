for a in 0 1 2 3; do echo "ablate $a"; CLICA_FWD_IMAGE_ABLATE=$a CLICA_C5_GRAPH=0 bash tools/c5_trace.sh s22/a$a f16x2 2>&1 | grep -E "fwd_image" | head -2; done
