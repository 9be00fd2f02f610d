#!/bin/bash
# Round 6: the single-pose solve as one persistent launch against a launch per evaluation, same box, interleaved.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06persist; mkdir -p $O
timeout 300 python3 -m pytest tests/test_gpu_match.py -x -q -m gpu -k "solve_single_pose" 2>&1 | tail -2
for i in 1 2 3; do
timeout 300 python3 bench.py --steps 40 --warmup 3 --no-secondary --host-steps 0 --cpu-scans 2 > $O/on_$i.json 2> $O/on.err; echo rc=$?
timeout 300 python3 bench.py --steps 40 --warmup 3 --no-secondary --host-steps 0 --no-cpu-baseline --no-persistent-solve > $O/off_$i.json 2> $O/off.err; echo rc=$?
done
for f in $O/on_*.json $O/off_*.json; do python3 -c "
import json
d=json.load(open('$f'))
p=d.get('parity') or {}
print('$f', round(d['value'],1), round(d['ms_per_step'],5), p.get('max_dt_m'), p.get('same_iterations_and_termination'), d['roofline'].get('avg_launch_ms'))
"; done; tail -2 $O/on.err
