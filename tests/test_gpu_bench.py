"""bench.py end to end on the GPU (short run): the contract JSON line with its roofline / kernels / loss legs and the split probe."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_emits_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "kernels", "loss_kernel", "split_bf16_probe"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["value"] > 100
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and rf["peak"] == 157.3 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert line["split_bf16_probe"]["value"] > 100
    assert abs(line["final_loss"]) < 20 and abs(line["split_bf16_probe"]["final_loss"]) < 20
