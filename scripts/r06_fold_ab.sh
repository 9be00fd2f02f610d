#!/bin/bash
# Round 6: the level partition's classification folded into the residual launch in front (partition_fold) against the
# lookup pass of its own, at partition_at 2 and 3; one box, interleaved. Tests of the batched matcher first.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06fold; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_match.py -x -q -m gpu -k "batch" 2>&1 | tail -3
for rep in 1 2 3; do
for v in "1 2" "0 2" "1 3" "0 3" "1 1"; do
  set -- $v
  HG_PARTITION_FOLD=$1 HG_PARTITION_AT=$2 timeout 300 python3 bench.py --workload match_batch --batch 64 --steps 10 --no-cpu-baseline > $O/fold$1_at$2_$rep.json 2>/dev/null
done; done
for f in $O/*.json; do python3 -c "
import json
try:
  d=json.load(open('$f')); print('$f', round(d['value']), round(d['ms_per_step'],4))
except Exception as e: print('$f','ERR',e)
"; done
