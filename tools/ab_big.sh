#!/bin/bash
# A/B of variant builds of libkssicp.so on large single pairs: tools/ab_big.sh <alt1.so> [alt2.so ...]
# (variants placed under kss-icp_amd/lib/).  Prints tools/big_pair_time.py's per-iteration time for 1M and 300k points.
L=kss-icp_amd/lib
cp $L/libkssicp.so $L/_main.so
for v in _main.so "$@" _main.so; do
  cp $L/$v $L/libkssicp.so
  for n in 1000000 300000; do
    echo "$v n=$n: $(timeout -k 10 200 python tools/big_pair_time.py $n 30 2>/dev/null | grep 'per iteration')" || exit 1
  done
done
cp $L/_main.so $L/libkssicp.so
