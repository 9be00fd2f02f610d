#!/bin/bash
# Round 6: level partition of the window pass (window_partition) on / off and the iteration it starts at, one box,
# interleaved; no oracle leg (the parity gate of these workloads runs in the default line and in tests/).
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "1 2" "0 2" "1 3"; do
  set -- $v
  for w in "window --steps 20 --warmup 3" "window_batch --batch-submaps 8 --steps 8 --warmup 2"; do
    HG_WINDOW_PARTITION=$1 HG_PARTITION_AT=$2 timeout 600 python3 bench.py --workload $w --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('partition $1 at $2', '$w'.split()[0], round(d['value'],1), 'residual launch ms', round(r['avg_launch_ms'],5), 'lm', round(r.get('lm_avg_launch_ms',0),5), 'iterations', d['config'].get('mean_lm_iterations'))
"
  done
done; done
