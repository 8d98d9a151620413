#!/usr/bin/env python3
"""Condense gpurun_out/profile_<tag>/ (rocprofv3 kernel-trace stats + PMC passes, see
tools/profile_round.sh) into profiles/<tag>_*.{csv,md} -- the committed, judged artefacts."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
src = f"gpurun_out/profile_{tag}"
os.makedirs("profiles", exist_ok=True)
shutil.copy(f"{src}/trace/t_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
shutil.copy(f"{src}/bench_under_rocprof.json", f"profiles/{tag}_bench_under_rocprof.json")
if os.path.exists(f"{src}/trace_native/t_kernel_stats.csv"):       # native fp32-MFMA mode (kernel trace only)
    shutil.copy(f"{src}/trace_native/t_kernel_stats.csv", f"profiles/{tag}_native_fp32_kernel_stats.csv")
    shutil.copy(f"{src}/bench_native_under_rocprof.json", f"profiles/{tag}_native_fp32_bench_under_rocprof.json")


def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"_n": len(next(iter(cs.values())))} for k, cs in d.items()}


pm = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds", "pmc_valu", "pmc_l2"):
    p = f"{src}/{name}/p_counter_collection.csv"
    if os.path.exists(p):
        for k, v in agg(p).items():
            pm.setdefault(k, {}).update(v)
stats = list(csv.DictReader(open(f"{src}/trace/t_kernel_stats.csv")))
bench = json.load(open(f"{src}/bench_under_rocprof.json"))
lines = [f"# Profile {tag}: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-native-leg --no-dropin`", "",
         f"bench line under the profiler: {bench['value']:.1f} steps/s, {bench['ms_per_step']:.3f} ms/step ({bench['dtype']}); roofline entry: "
         f"`{bench['roofline']['kernel']}` {bench['roofline']['achieved']} TFLOP/s ({bench['roofline']['frac']:.3f} of {bench['roofline']['peak']}), "
         f"avg {bench['roofline']['avg_launch_us']} us/launch (device time stamps around the launches inside the replayed step graph, bench.py roofline leg).", "",
         "PMC columns come from separate `--pmc` passes (FETCH_SIZE / WRITE_SIZE in KB per launch, as reported; per "
         "MI355X_MICROARCH.md FETCH_SIZE under-counts wide coalesced reads by 2x on gfx950 -> `fetch_x2_MB`). "
         "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs).", "",
         "VALU busy = SQ_ACTIVE_INST_VALU x 4 (quad-cycles -> cycles) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the share of SIMD cycles "
         "in which the vector ALU is executing; VALU/wave = SQ_INSTS_VALU / SQ_WAVES; wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / "
         "barrier), issue-stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.", "",
         "L2 hit = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) over the eight per-XCD L2s.", "",
         "| kernel | calls | avg us | % time | fetch_x2 MB | write MB | L2 hit | MfmaUtil | VALU busy | VALU inst/launch | LDS inst/launch | wait | issue-stall | LDS bank conflicts |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for r in stats:
    name = r["Name"]
    c = pm.get(name, {})
    fetch = f"{2 * c['FETCH_SIZE'] / 1024:.1f}" if "FETCH_SIZE" in c else ""
    write = f"{c['WRITE_SIZE'] / 1024:.1f}" if "WRITE_SIZE" in c else ""
    util = ""
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and c.get("GRBM_GUI_ACTIVE", 0) > 0:
        util = f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.2f}"
    conf = f"{c['SQ_LDS_BANK_CONFLICT']:.0f}" if "SQ_LDS_BANK_CONFLICT" in c else ""
    short = name.split("(")[0].replace("void ", "")
    vbusy = f"{4 * c['SQ_ACTIVE_INST_VALU'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.2f}" if c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE") else ""
    vinst = f"{c['SQ_INSTS_VALU']:.0f}" if "SQ_INSTS_VALU" in c else ""
    linst = f"{c['SQ_INSTS_LDS']:.0f}" if "SQ_INSTS_LDS" in c else ""
    wait = f"{c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f}" if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c else ""
    stall = f"{c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f}" if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in c else ""
    l2 = f"{c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.2f}" if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0 else ""
    lines.append(f"| `{short}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} | {fetch} | {write} | {l2} | {util} | {vbusy} | "
                 f"{vinst} | {linst} | {wait} | {stall} | {conf} |")
open(f"profiles/{tag}_summary.md", "w").write("\n".join(lines) + "\n")
# per-launch HBM-side traffic (FETCH_SIZE x 2 + WRITE_SIZE, bytes) of every kernel symbol: bench.py's roofline.traffic
traffic = {}
for name, c in pm.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic[name.split("(")[0].replace("void ", "")] = {"fetch_x2_bytes": 2 * c["FETCH_SIZE"] * 1024, "write_bytes": c["WRITE_SIZE"] * 1024,
                                                            "launches_sampled": c["_n"]}
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of tools/profile_round.sh {tag} (KB as reported; FETCH x2 per "
                     "MI355X_MICROARCH.md); per-launch averages", "kernels": traffic}, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
print("\n".join(lines[:14]))

# ---- BASELINE config 3 (n = 40) profile: kernel stats + traffic + MFMA busy
if os.path.exists(f"{src}/trace_c3/t_kernel_stats.csv"):
    shutil.copy(f"{src}/trace_c3/t_kernel_stats.csv", f"profiles/{tag}_c3_kernel_stats.csv")
    shutil.copy(f"{src}/bench_c3_under_rocprof.json", f"profiles/{tag}_c3_bench_under_rocprof.json")
    pm3 = {}
    for name in ("pmc_c3_fetch", "pmc_c3_write", "pmc_c3_sq"):
        p_ = f"{src}/{name}/p_counter_collection.csv"
        if os.path.exists(p_):
            for k, v in agg(p_).items():
                pm3.setdefault(k, {}).update(v)
    b3 = json.load(open(f"{src}/bench_c3_under_rocprof.json"))
    out3 = [f"# Profile {tag}, BASELINE config 3 on one rank: `bench.py --n 40 --space-type sphere --p 1` (B = 6144, 13.6 M parameters, 1.005 TFLOP of encoder GEMMs per step)", "",
            f"bench line under the profiler: {b3['value']:.1f} steps/s, {b3['ms_per_step']:.3f} ms/step ({b3['dtype']}); roofline entry `{b3['roofline']['kernel']}` "
            f"{b3['roofline']['achieved']} TFLOP/s = {b3['roofline']['frac']:.3f} of {b3['roofline']['peak']}.", "",
            "| kernel | calls | avg us | % time | fetch_x2 MB | write MB | MfmaUtil |", "|---|---|---|---|---|---|---|"]
    tr3 = {}
    for r in csv.DictReader(open(f"{src}/trace_c3/t_kernel_stats.csv")):
        c = pm3.get(r["Name"], {})
        fetch = f"{2 * c['FETCH_SIZE'] / 1024:.1f}" if "FETCH_SIZE" in c else ""
        write = f"{c['WRITE_SIZE'] / 1024:.1f}" if "WRITE_SIZE" in c else ""
        util = f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.2f}" if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and c.get("GRBM_GUI_ACTIVE", 0) > 0 else ""
        short = r["Name"].split("(")[0].replace("void ", "")
        out3.append(f"| `{short}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} | {fetch} | {write} | {util} |")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            tr3[short] = {"fetch_x2_bytes": 2 * c["FETCH_SIZE"] * 1024, "write_bytes": c["WRITE_SIZE"] * 1024, "launches_sampled": c["_n"]}
    open(f"profiles/{tag}_c3_summary.md", "w").write("\n".join(out3) + "\n")
    json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh {tag} on bench.py --n 40 --space-type sphere --p 1", "kernels": tr3},
              open(f"profiles/{tag}_c3_traffic.json", "w"), indent=1)
    print("\n".join(out3[:12]))
