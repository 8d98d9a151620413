#!/usr/bin/env python3
"""profiles/r4_c4_*, r4_c5_*: who owns the GPU time of BASELINE configs[3] / [4]?  Reads gpurun_out/profile_r4_<c>/t_kernel_stats.csv
(rocprofv3 --kernel-trace --stats of `bench.py --config <c>`, tools/profile_c45.sh), buckets the kernels and writes
profiles/r4_<c>_kernel_stats.csv (verbatim copy) + profiles/r4_<c>_summary.md."""
import csv, json, os, shutil, sys

BUCKETS = [
    ("HIP library (cl_ica_amd: conv stack of config 5 since the clica_conv_* kernels, head Linear / Softclip / LeakyReLU, Lp loss sweeps, flat Adam)", lambda n: "clica::" in n),
    ("MIOpen / rocBLAS convolution kernels (forward, data and weight gradients)",
     lambda n: any(t in n.lower() for t in ("conv", "igemm", "sp3", "cijk_", "gemm", "winograd", "im2col", "col2im", "miopen", "xdlops", "implicit"))
     and "batchnorm" not in n.lower() and "batch_norm" not in n.lower()),
    ("BatchNorm (MIOpen / ATen)", lambda n: "batchnorm" in n.lower() or "batch_norm" in n.lower() or "bn_" in n.lower()),
    ("ATen element-wise / reductions / pooling / copies (ReLU, residual adds, max-pool, avg-pool, layout, zero_)", lambda n: True),
]
TAG = os.environ.get("PTAG", "r4")
for c in sys.argv[1:] or ["c4", "c5"]:
    src = f"gpurun_out/profile_{TAG}_{c}"
    rows = list(csv.DictReader(open(f"{src}/t_kernel_stats.csv")))
    shutil.copy(f"{src}/t_kernel_stats.csv", f"profiles/{TAG}_{c}_kernel_stats.csv")
    bench = json.loads(open(f"{src}/bench.json").read().strip().splitlines()[-1])
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    shares = [[name, 0.0, 0, []] for name, _ in BUCKETS]
    for r in rows:
        for i, (_, pred) in enumerate(BUCKETS):
            if pred(r["Name"]):
                shares[i][1] += float(r["TotalDurationNs"]); shares[i][2] += int(r["Calls"]); shares[i][3].append(r)
                break
    lines = [f"# BASELINE configs[{3 if c == 'c4' else 4}] on one MI355X: `rocprofv3 --kernel-trace --stats -- python bench.py --config {c} --steps 10`", "",
             bench["workload"], "",
             f"bench line under the profiler: **{bench['value']:.1f} steps/s** ({bench['ms_per_step']:.2f} ms/step, {bench['parameters']} parameters, "
             f"final loss {bench['final_loss']:.4f}); GPU kernel time in the trace: {tot / 1e6:.1f} ms over all warm-up + timed steps.", "",
             "| share of GPU kernel time | launches | bucket |", "|---|---|---|"]
    for name, ns, calls, _ in shares:
        lines.append(f"| {100 * ns / tot:.1f} % | {calls} | {name} |")
    lines += ["", "Top kernels:", "", "| kernel | calls | avg us | % time |", "|---|---|---|---|"]
    for r in rows[:14]:
        lines.append(f"| `{r['Name'].split('(')[0][:110]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |")
    open(f"profiles/{TAG}_{c}_summary.md", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))
