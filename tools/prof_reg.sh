cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/preg && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/preg -- python3 tools/register_time.py > gpurun_out/reg.txt 2>&1; python - <<'PY'
import csv,glob
from collections import defaultdict
f=glob.glob('gpurun_out/preg/*/*_kernel_trace.csv')[0]
acc=defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("kss::","")
    acc[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1]))[:8]:
    print("%-50s n=%6d total %9.1f us avg %7.1f" % (k[:50], len(v), sum(v), sum(v)/len(v)))
PY
