#!/bin/bash
# kernel trace of the C3 batch (cell lists only): per-launch durations of the fused pass in launch order
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pc3q
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pc3q -- python3 tools/bench_configs.py c3 --grid-only > gpurun_out/pc3q.txt 2>&1 || { tail -3 gpurun_out/pc3q.txt; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/pc3q/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'grid_pass' in r['Kernel_Name'] or 'gridb_pass' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000 for r in rows]
print("grid_pass launches:", len(d))
print("us:", " ".join("%.0f" % x for x in d))
PY
