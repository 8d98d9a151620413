import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_mlp as T
for dims in ([4, 40, 200, 200, 200, 200, 40, 4], [4, 40, 200, 40, 4], [4, 40, 200, 200, 40, 4]):
    for M in (96, 100, 112, 128, 144, 160, 192, 256, 1024):
        for rep in range(3):
            try:
                T.test_split_bf16_wgrad_matches_fp64(dims, M)
                res = "ok"
            except AssertionError as e:
                res = "FAIL " + str(e)[:150].replace("\n", " ")
            print(dims, M, rep, res, flush=True)
