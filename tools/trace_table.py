#!/usr/bin/env python3
"""Markdown table of per-kernel durations from a rocprofv3 --kernel-trace CSV directory.
usage: trace_table.py <dir> <title> <command line that was profiled>"""
import csv, glob, sys
from collections import defaultdict
d, title, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
files = glob.glob(d + "/*/*_kernel_trace.csv")
acc = defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kss::", "")
        acc[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values()) or 1.0
print("## %s\n\nCommand: `rocprofv3 --kernel-trace --output-format csv -- %s`\n" % (title, cmd))
print("| kernel | calls | total ms | avg us | median us | min us | max us | % |\n|---|---|---|---|---|---|---|---|")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    s = sorted(v)
    print("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.2f | %.1f |" % (k[:70], len(v), sum(v) / 1e3, sum(v) / len(v), s[len(s) // 2], s[0], s[-1], 100 * sum(v) / tot))
print()
