#!/bin/bash
# Round 6: what a level partition could buy the window pass -- k_window_residuals with EVERY lookup stopped at the
# finest level (scripts/libhg_forcel1.so = bash scripts/build_variant.sh forcel1 -DHG_FORCE_L1: wrong results, timing
# only) against the shipped library; per-launch durations by HIP events (prof-every 1).
cd $GRAFT_REPO_ROOT
R=$(pwd)
for rep in 1 2; do
for lib in "" $R/scripts/libhg_forcel1.so; do
  for w in "window --steps 20 --warmup 3" "window_batch --batch-submaps 8 --steps 8 --warmup 2"; do
    HG_LIB_PATH=$lib timeout 300 python3 bench.py --workload $w --no-cpu-baseline --no-secondary --prof-every 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('${lib:-shipped}'.split('/')[-1], '$w'.split()[0], round(d['value'],1), 'residual launch ms', round(r['avg_launch_ms'],5), 'lm', round(r.get('lm_avg_launch_ms',0),5), 'iterations', d['config'].get('mean_lm_iterations'))
"
  done
done; done
