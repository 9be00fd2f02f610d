#!/bin/bash
# Round-5 A/B inside ONE gpurun call: the insert parity gates on the current build, then headline / 120-step
# trajectory / scan streams with every library variant given (scripts/libhg_<variant>.so; "cur" = the tree's build).
# Usage: bash scripts/r05_ab.sh [--no-tests] variant...
cd $GRAFT_REPO_ROOT
if [ "$1" != "--no-tests" ]; then
  python -m pytest tests/test_gpu_insert.py tests/test_gpu_headline.py -x -q 2>&1 | tail -3
else shift; fi
run() {  # tag, bench args
  tag=$1; shift
  for rep in 1 2; do
    for v in $VARIANTS; do
      if [ $v = cur ]; then lib=$PWD/hectorgrapher_amd/libhg_mi355x.so; else lib=$PWD/scripts/libhg_$v.so; fi
      HG_LIB_PATH=$lib python bench.py "$@" --no-cpu-baseline --no-secondary 2>/dev/null | python scripts/ab_print.py "$tag/$v"
    done
  done
}
VARIANTS="$@"
run headline
run traj120 --steps 120
run stream32 --workload insert_stream --stream-scans 32 --steps 4 --warmup 1
run stream64hbm --workload insert_stream --stream-scans 64 --stream-tiles 64 --steps 3 --warmup 1
