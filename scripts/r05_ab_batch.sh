#!/bin/bash
# Round-5 A/B of the batched workloads inside ONE gpurun call (same box, interleaved): the tree's build against
# library variants under scripts/libhg_<variant>.so. Usage: bash scripts/r05_ab_batch.sh variant...
cd $GRAFT_REPO_ROOT
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
run match16 --workload match_batch --batch 16 --steps 10
run match64 --workload match_batch --batch 64 --steps 10
run regbatch8 --workload register_batch --batch-submaps 8 --steps 40
run regbatch16 --workload register_batch --batch-submaps 16 --steps 40
run offline8 --total-submaps 8 --scans-per-submap 500
run filtered --workload register_filtered
run headline
