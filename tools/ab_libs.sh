#!/bin/bash
# A/B of two builds of libkssicp.so on the same GPU box: tools/ab_libs.sh <alt.so> [rounds]
# (alt.so: a variant library placed under kss-icp_amd/lib/).  Prints C2 iterations/s and C3 seconds for each, interleaved.
ALT=$1; N=${2:-2}
L=kss-icp_amd/lib
cp $L/libkssicp.so $L/_main.so
for r in $(seq $N); do
  for v in main alt; do
    if [ $v = main ]; then cp $L/_main.so $L/libkssicp.so; else cp $L/$ALT $L/libkssicp.so; fi
    c2=$(timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print('%.0f' % json.loads(sys.stdin.read())['value'])") || exit 1
    c3=$(timeout -k 10 200 python tools/bench_configs.py c3 --grid-only 2>/dev/null | tail -1 | python -c "import sys,json; print('%.3f ms' % (1e3*json.loads(sys.stdin.read())['grid']['seconds']))") || exit 1
    echo "$v: C2 $c2 it/s   C3 $c3"
  done
done
cp $L/_main.so $L/libkssicp.so
