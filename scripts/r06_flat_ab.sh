#!/bin/bash
# Round 6: the job-table kernels with their pointers passed through the device-memory address space (global instead of
# FLAT operations) -- rates of the workloads that use them.
cd $GRAFT_REPO_ROOT
run() { timeout 600 python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*'[:70], round(d['value'],1), round(d['ms_per_step'],4))"; }
for rep in 1 2; do
run --workload insert_stream --stream-scans 32
run --workload insert_stream --stream-scans 64 --stream-tiles 64 --steps 3 --warmup 1 --prof-every 1
run --workload register_batch --batch-submaps 8 --steps 12 --warmup 2
run --workload window_batch --batch-submaps 8 --steps 8 --warmup 2
run --workload match_batch --batch 64 --steps 10
run --workload insert_stream --stream-scans 32 --insert-mode fast
done
