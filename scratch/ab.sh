#!/bin/bash
# usage: ab.sh "<bench args>" variant...
args="$1"; shift
for rep in 1 2; do
  for v in "$@"; do
    HG_LIB_PATH=$PWD/scratch/libhg_$v.so python bench.py $args --no-cpu-baseline 2>/dev/null | python scratch/ab_print.py $v
  done
done
