#!/bin/bash
# A/B of the headline inside ONE gpurun call. Usage: bash scripts/r05_ab_head.sh variant...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in "$@"; do
    if [ $v = cur ]; then lib=$PWD/hectorgrapher_amd/libhg_mi355x.so; else lib=$PWD/scripts/libhg_$v.so; fi
    HG_LIB_PATH=$lib python bench.py --no-secondary --cpu-scans 3 2>/dev/null | python scripts/ab_print.py "headline/$v"
    HG_LIB_PATH=$lib python bench.py --no-secondary --no-cpu-baseline --steps 120 2>/dev/null | python scripts/ab_print.py "traj120/$v"
  done
done
