#!/bin/bash
# A/B timing of library builds under ab/*.so within ONE gpurun call (box-to-box variance is 5-10%)
# usage: scripts/ab_hmc.sh name1 name2 ...   (each ab/lib_<name>.so), two rounds each, interleaved
for round in 1 2; do
  for n in "$@"; do
    echo "== $n (round $round)"
    GLMMR_MCML_LIB=$PWD/ab/lib_$n.so timeout -k 10 200 python scripts/time_hmc.py 5000 1024 2>&1 | grep "hmc warm=10"
  done
done
