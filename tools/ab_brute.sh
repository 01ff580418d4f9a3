#!/bin/bash
# A/B of variant builds of libkssicp.so on the brute-force sweep: tools/ab_brute.sh <alt.so> ...
# prints bench.py's brute_force leg (C2, exact brute force) for each
L=kss-icp_amd/lib
cp $L/libkssicp.so $L/_main.so
for v in _main.so "$@" _main.so; do
  cp $L/$v $L/libkssicp.so
  echo "$v: $(timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['brute_force']; print('%.1f it/s, sweep %.1f us, frac %.3f' % (b['value'], 1e3*b['roofline']['avg_launch_ms'], b['roofline']['frac']))")" || exit 1
done
cp $L/_main.so $L/libkssicp.so
