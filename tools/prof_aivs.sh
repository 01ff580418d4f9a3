#!/bin/bash
# kernel trace of the AIVS down-sampler on the reference's Bunny pair (tools/register_time.py): per-kernel totals of the aivs_* kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/paivs
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/paivs -- python3 tools/register_time.py > gpurun_out/paivs.txt 2>&1 || { tail -3 gpurun_out/paivs.txt; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/paivs/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'aivs' in n or 'cand_' in n or 'rot_search' in n or 'preshape' in n:
        print("%-60s calls %5s  total %9.1f us  avg %8.2f us" % (n[:60], r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3))
PY
