#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv compactly; optional substring filter."""
import csv, sys
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")[:70]
    if flt in n:
        print(f"{n:72s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:9.2f} us {float(r['Percentage']):6.2f}%")
