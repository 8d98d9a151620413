"""bench.py end to end on the GPU (short run): the contract JSON line with its roofline / kernels / loss legs and the native-fp32 leg."""
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
              "data", "config", "roofline", "kernels", "loss_kernel", "native_fp32"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["value"] > 100
    assert line["metric"] == "training steps/sec (B=6144, n=10 MLP)" and "f16x2 split" in line["dtype"]
    rf = line["roofline"]      # headline mode: issued fp16 flops (3 x algorithmic) against the dense fp16 / bf16 peak, fp32-equivalent figure next to it
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and rf["peak"] == 2500.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    fe = rf["fp32_equivalent"]
    # (the fp32-EQUIVALENT rate may exceed the fp32 matrix peak: the three fp16 products of one fp32 product cost 0.19 of its matrix time)
    assert fe["peak"] == 157.3 and 0 < fe["frac"] < 5.0 and abs(3 * fe["achieved"] - rf["achieved"]) < 0.05 * rf["achieved"]
    b3 = line["split_bf16x3"]      # the rounds 3-4 arithmetic on the same box in the same call
    assert b3["value"] > 100 and abs(b3["final_loss"]) < 20
    dr = line["dry_ranks_8"]       # rank 0 of an 8-rank job planned, captured with its RCCL collectives and run on this GPU
    assert "error" not in dr, dr
    assert dr["value"] > 100 and dr["negatives_pool"] == 8 * 6144 and dr["collective_bytes_per_step"] > 0 and dr["plan"]["planned_ranks"] == 8
    nat = line["native_fp32"]
    assert nat["value"] > 100 and nat["dtype"] == "f32" and nat["roofline"]["peak"] == 157.3 and 0 < nat["roofline"]["frac"] < 1
    assert abs(line["final_loss"]) < 20 and abs(nat["final_loss"]) < 20
