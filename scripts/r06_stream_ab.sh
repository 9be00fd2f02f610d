#!/bin/bash
# Round 6: merged apply of a scan stream's group (stream_merge) against one apply launch per scan, same box.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06stream; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_insert.py -x -q -m gpu 2>&1 | tail -3
run() { # tag, env, args
  env $2 timeout 600 python3 bench.py --workload insert_stream $3 > $O/$1.json 2> $O/$1.err; echo "$1 rc=$?"
}
run auto32 "HG_STREAM_MERGE=1" "--stream-scans 32 --cpu-scans 2"
run perscan32 "HG_STREAM_MERGE=0" "--stream-scans 32 --no-cpu-baseline"
run forced32 "HG_STREAM_MERGE=2" "--stream-scans 32 --cpu-scans 2"
run merged32hbm "HG_STREAM_MERGE=1" "--stream-scans 32 --stream-tiles 32 --cpu-scans 2"
run perscan32hbm "HG_STREAM_MERGE=0" "--stream-scans 32 --stream-tiles 32 --no-cpu-baseline"
run merged64hbm "HG_STREAM_MERGE=1" "--stream-scans 64 --stream-tiles 64 --steps 3 --warmup 1 --prof-every 1 --cpu-scans 2"
run perscan64hbm "HG_STREAM_MERGE=0" "--stream-scans 64 --stream-tiles 64 --steps 3 --warmup 1 --prof-every 1 --no-cpu-baseline"
run merged64warm "HG_STREAM_MERGE=2" "--stream-scans 64 --steps 3 --warmup 1 --prof-every 1 --cpu-scans 2"
run perscan64warm "HG_STREAM_MERGE=0" "--stream-scans 64 --steps 3 --warmup 1 --prof-every 1 --no-cpu-baseline"
run merged500 "HG_STREAM_MERGE=1" "--stream-scans 500 --stream-tiles 400 --max-blocks 1048576 --steps 2 --warmup 1 --prof-every 1 --cpu-scans 2"
run perscan500 "HG_STREAM_MERGE=0" "--stream-scans 500 --stream-tiles 400 --max-blocks 1048576 --steps 2 --warmup 1 --prof-every 1 --no-cpu-baseline"
for f in $O/*.json; do python3 -c "
import json
try:
  d=json.load(open('$f'))
  r=d['roofline']
  print('$f', round(d['value']), (d.get('parity') or {}).get('bit_exact'), {k:round(v,3) for k,v in r['per_kernel_ms_total'].items() if v}, round(r['frac'],4))
except Exception as e: print('$f', 'ERR', e)
"; done
tail -3 $O/auto32.err
