#!/bin/bash
# Round 6: slice partition of the heavy bins in front of the merged stream apply (stream_partition), same box, interleaved.
# (needs docs/r06_stream_partition.patch applied: the pass was measured and not shipped, docs/EXPERIMENTS.md)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06part; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_insert.py tests/test_gpu_headline.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for v in 1 0; do
  HG_STREAM_PARTITION=$v timeout 300 python3 bench.py --workload insert_stream --stream-scans 32 --cpu-scans 2 > $O/room_p${v}_$rep.json 2>/dev/null
  HG_STREAM_PARTITION=$v timeout 300 python3 bench.py --workload insert_stream --stream-scans 64 --stream-tiles 64 --steps 3 --warmup 1 --prof-every 1 --cpu-scans 2 > $O/hbm64_p${v}_$rep.json 2>/dev/null
  HG_STREAM_PARTITION=$v timeout 300 python3 bench.py --workload insert_stream --stream-scans 64 --steps 3 --warmup 1 --prof-every 1 --cpu-scans 2 > $O/warm64_p${v}_$rep.json 2>/dev/null
done; done
for f in $O/*.json; do python3 -c "
import json
try:
  d=json.load(open('$f')); r=d['roofline']
  print('$f', round(d['value']), (d.get('parity') or {}).get('bit_exact'), {k:round(v,3) for k,v in r['per_kernel_ms_total'].items() if v})
except Exception as e: print('$f','ERR',e)
"; done
