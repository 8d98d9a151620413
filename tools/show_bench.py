#!/usr/bin/env python3
"""Pretty-print the last JSON line of a bench.py output file."""
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{b['value']:.1f} {b['unit']}  {b['ms_per_step']:.4f} ms/step  loss {b.get('final_loss')}")
if "roofline" in b:
    print("roofline:", b["roofline"])
for r in b.get("kernels", []):
    print(f"  {r['op']:18s} {r['launches_per_step']}x {r['avg_us']:8.1f} us  {r['tflops']:6.1f} TF  {r['kernel']}")
if "loss_kernel" in b:
    print("  loss:", b["loss_kernel"])
if "cpu_baseline" in b:
    print("  cpu:", b["cpu_baseline"])
