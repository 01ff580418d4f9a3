cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pc3 && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pc3 -- python3 tools/bench_configs.py c3 --pairs ${PAIRS:-1024} > gpurun_out/c3.txt 2>&1; tail -1 gpurun_out/c3.txt; python - <<'PY'
import csv,glob
from collections import defaultdict
f=glob.glob('gpurun_out/pc3/*/*_kernel_trace.csv')[0]
acc=defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("kss::","")
    acc[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1])):
    print("%-60s n=%5d total %9.1f us  avg %8.1f  min %8.1f" % (k[:60], len(v), sum(v), sum(v)/len(v), min(v)))
PY
