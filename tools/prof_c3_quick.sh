#!/bin/bash
# kernel trace of the C3 batch (cell lists only): per-kernel average durations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pc3q
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc3q -- python3 tools/bench_configs.py c3 --grid-only > gpurun_out/pc3q.txt 2>&1 || { tail -3 gpurun_out/pc3q.txt; exit 1; }
tail -1 gpurun_out/pc3q.txt | cut -c1-300
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/pc3q/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['TotalDurationNs']) > 20000: print("  %-70s calls %4s avg %9.1f us" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
