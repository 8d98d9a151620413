#!/usr/bin/env python3
"""GPU timeline of one drop-in train_step from a rocprofv3 kernel trace: where the device waits for the host.
    rocprofv3 --kernel-trace -d gpurun_out/dropin_tl -o t --output-format csv -- python tools/dropin_profile.py
    python tools/dropin_timeline.py gpurun_out/dropin_tl
Prints the median step's kernels: start offset, duration and the idle gap in front of each (us)."""
import csv, glob, os, sys

root = sys.argv[1]
path = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a step ends with the Adam kernel
ends = [i for i, r in enumerate(rows) if "adam" in r[2].lower()]
steps = [rows[a + 1:b + 1] for a, b in zip(ends[:-1], ends[1:])]
steps = steps[len(steps) // 2:]                      # steady state
steps.sort(key=lambda s: s[-1][1] - s[0][0])
med = steps[len(steps) // 2]
t0 = med[0][0]
busy = 0
prev_end = None
print(f"{len(steps)} steps; median step: first kernel start -> last kernel end {(med[-1][1] - t0) / 1e3:.1f} us, {len(med)} kernels")
for s, e, name in med:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    busy += e - s
    short = name.split("(")[0][-60:]
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  gap {gap:7.1f}  {short}")
    prev_end = e
print(f"busy {busy / 1e3:.1f} us of {(med[-1][1] - t0) / 1e3:.1f}")
# step-to-step period (adam end to adam end)
per = sorted((b[-1][1] - a[-1][1]) / 1e3 for a, b in zip(sorted(steps, key=lambda s: s[0][0])[:-1], sorted(steps, key=lambda s: s[0][0])[1:]))
print(f"period median {per[len(per) // 2]:.1f} us")
