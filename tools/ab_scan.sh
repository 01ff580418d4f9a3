#!/bin/bash
# A/B of the cell-count scan on the 1M-point pair: this repository's three-phase scan against the configured library scan
# (KSS_SCAN_LIB=1), the scan kernels' time by rocprofv3.  (Round 3 chose the library configuration with this script and a
# temporary KSS_SCAN_CFG hook: 21 / 32 / 48 / 64 items per lane -> 110 / 89 / 81 / 76 us; the default configuration: 127.)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in own lib; do
  rm -rf gpurun_out/pscan
  if [ $cfg = lib ]; then export KSS_SCAN_LIB=1; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pscan -- python3 tools/bench_configs.py c4 --grid-only > gpurun_out/pscan.txt 2>&1 || { tail -3 gpurun_out/pscan.txt; exit 1; }
  echo "cfg $cfg: $(grep -o '"nn_sweep_ms_grid": [0-9.]*' gpurun_out/pscan.txt | tail -1)"
  python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/pscan/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'scan' in n.lower() or 'lookback' in n.lower():
        print("    %-60s calls %s avg %.1f us" % (n[:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
