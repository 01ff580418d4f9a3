cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pf3 && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pf3 -- python3 tools/aivs_time.py > gpurun_out/f3.txt 2>&1; grep -v amdgpu gpurun_out/f3.txt | tail -3; python - <<'PY'
import csv,glob
from collections import defaultdict
f=glob.glob('gpurun_out/pf3/*/*_kernel_trace.csv')[0]
acc=defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("kss::","")
    acc[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1]))[:14]:
    print("%-70s n=%5d total %10.1f us  avg %9.1f  max %9.1f" % (k[:70], len(v), sum(v), sum(v)/len(v), max(v)))
PY
