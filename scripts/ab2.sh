#!/bin/bash
# A/B of (library variant, environment) pairs inside ONE gpurun call: ab2.sh "<bench args>" "variant[:VAR=x]" ...
args="$1"; shift
for rep in 1 2; do
  for v in "$@"; do
    lib=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs="A=1"
    env HG_LIB_PATH=$PWD/scripts/libhg_$lib.so $envs python bench.py $args --no-cpu-baseline 2>/dev/null | python scripts/ab_print.py "$v"
  done
done
