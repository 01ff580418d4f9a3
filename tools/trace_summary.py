#!/usr/bin/env python3
"""Per-kernel mean duration (us) from a rocprofv3 kernel-trace CSV directory; optional name filter."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
acc = defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kss::", "")
    acc[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if k.startswith(("grid_", "nn_", "corr_", "finalize", "scan_", "pack_")):
        v2 = sorted(v)
        print("%-40s n=%4d mean=%8.2f med=%8.2f min=%8.2f" % (k[:40], len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0]))
