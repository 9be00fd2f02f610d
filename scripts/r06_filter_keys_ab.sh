#!/bin/bash
# Round 6: keys per thread in flight in the slice filter of the apply pass (HG_FILTER_KEYS: 8 shipped until now, 16, 24).
cd $GRAFT_REPO_ROOT
R=$(pwd)
run() { lib=$1; shift; HG_LIB_PATH=$lib timeout 600 python3 bench.py "$@" --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$(basename $lib)', '$*'[:60], round(d['value'],1), {k:round(v,3) for k,v in (r.get('per_kernel_ms_total') or {}).items() if k in ('apply',)}, r.get('per_kernel_launches',{}).get('apply'))"; }
for rep in 1 2; do
for lib in "" $R/scripts/libhg_fk16.so $R/scripts/libhg_fk24.so; do
  run "$lib" --host-steps 0
  run "$lib" --workload insert_stream --stream-scans 32
  run "$lib" --workload insert_stream --stream-scans 64 --stream-tiles 64 --steps 3 --warmup 1 --prof-every 1
  run "$lib" --workload register_batch --batch-submaps 8 --steps 12 --warmup 2
done; done
