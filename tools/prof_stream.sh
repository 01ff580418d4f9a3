cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pt && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt -- python3 tools/bench_configs.py stream > gpurun_out/stream.txt 2>&1; tail -1 gpurun_out/stream.txt; python - <<'PY'
import csv,glob
from collections import defaultdict
f=glob.glob('gpurun_out/pt/*/*_kernel_trace.csv')[0]
acc=defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("kss::","")
    acc[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in acc.items():
    if k.startswith(("preshape","sum_col","pose")): print(k, len(v), round(sum(v)/len(v),1), round(min(v),1))
PY
