cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for b in 32 16; do
for pm in 48 8; do
  HG_PARTITION_MIN=$pm timeout 300 python3 bench.py --workload match_batch --batch $b --steps 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b partition_min $pm', round(d['value']), round(d['ms_per_step'],4))"
done; done; done
