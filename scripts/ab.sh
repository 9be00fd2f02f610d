#!/bin/bash
# A/B of library builds inside ONE gpurun call (box-to-box variance is +-7 %): build variants as
# scripts/libhg_<variant>.so, then: ab.sh "<bench args>" variant...
args="$1"; shift
for rep in 1 2; do
  for v in "$@"; do
    HG_LIB_PATH=$PWD/scripts/libhg_$v.so python bench.py $args --no-cpu-baseline 2>/dev/null | python scripts/ab_print.py $v
  done
done
