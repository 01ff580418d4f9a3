#!/bin/bash
# SQ counters of the fused pass on one large well-posed pair (tools/big_pair_time.py): issue rates per launch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
N=${1:-1000000}
rm -rf gpurun_out/pbig
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d gpurun_out/pbig/sq -- python3 tools/big_pair_time.py $N 10 > gpurun_out/pbig_sq.txt 2>&1 || { tail -3 gpurun_out/pbig_sq.txt; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d gpurun_out/pbig/sq2 -- python3 tools/big_pair_time.py $N 10 > gpurun_out/pbig_sq2.txt 2>&1 || { tail -3 gpurun_out/pbig_sq2.txt; }
python3 - <<'PY'
import csv, glob, collections
for d in ('sq', 'sq2'):
    fs = glob.glob('gpurun_out/pbig/%s/**/*counter_collection.csv' % d, recursive=True)
    if not fs: print(d, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0][:70]
        if 'grid_pass' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (k, r['Dispatch_Id'])
        if key not in seen: seen.add(key); cnt[k] += 1
    for k in acc:
        print(k, "launches", cnt[k])
        for c, v in sorted(acc[k].items()): print("   %-28s %14.0f per launch" % (c, v / cnt[k]))
PY
