#!/bin/bash
# A/B of environment switches inside ONE gpurun call: ab_env.sh "<bench args>" "VAR=a VAR2=b" "VAR=c" ...
args="$1"; shift
for rep in 1 2; do
  for v in "$@"; do
    env $v python bench.py $args --no-cpu-baseline 2>/dev/null | python scripts/ab_print.py "$v"
  done
done
