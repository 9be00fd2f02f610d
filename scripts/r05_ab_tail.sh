#!/bin/bash
# Round-5 A/B of the single-pose LM tail (k_tsdf_residuals_single / k_lm_single_batch) inside ONE gpurun call.
# Usage: bash scripts/r05_ab_tail.sh variant...   (scripts/libhg_<variant>.so; "cur" = the tree's build)
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  for rep in 1 2; do
    for v in $VARIANTS; do
      if [ $v = cur ]; then lib=$PWD/hectorgrapher_amd/libhg_mi355x.so; else lib=$PWD/scripts/libhg_$v.so; fi
      HG_LIB_PATH=$lib python bench.py "$@" --no-secondary 2>/dev/null | python scripts/ab_print.py "$tag/$v"
    done
  done
}
VARIANTS="$@"
run headline --cpu-scans 3
run traj120 --steps 120 --no-cpu-baseline
run match64 --workload match_batch --batch 64 --steps 10 --cpu-scans 2
run regbatch8 --workload register_batch --batch-submaps 8 --steps 40 --no-cpu-baseline
