#!/bin/bash
# Round-5 diagnostic: GPU busy / idle per step of the batched and window workloads (rocprofv3 kernel traces).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05gaps; mkdir -p $O
t() {  # tag, steps, marker, bench args
  tag=$1; steps=$2; marker=$3; shift 3
  timeout -s KILL 600 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o t -- python3 bench.py "$@" --no-cpu-baseline --no-secondary > $O/$tag.json 2> $O/$tag.err
  echo "== $tag: $(python3 scripts/ab_print.py $tag < $O/$tag.json)"
  python3 scripts/trace_gaps.py $(find $O/$tag -name "t_kernel_trace.csv" | head -1) $steps $marker | head -${LINES_PER:-14}
}
t regbatch8 30 k_lm_prepare_batch --workload register_batch --batch-submaps 8 --steps 40
t regbatch16 30 k_lm_prepare_batch --workload register_batch --batch-submaps 16 --steps 40
t offline8 100 k_lm_prepare_batch --total-submaps 8 --scans-per-submap 200
t window 10 k_bin_count --workload window
t window_batch8 6 k_bin_count_jobs --workload window_batch --batch-submaps 8 --steps 8 --warmup 2
