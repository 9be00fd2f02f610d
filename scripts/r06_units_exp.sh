#!/bin/bash
# Round 6: k_stream_units after the owners' items went to their wavefronts -- kernel duration (rocprofv3) and the stream
# rates, one room / 64 rooms.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for w in "--stream-scans 32" "--stream-scans 64 --stream-tiles 64 --prof-every 1"; do
  rm -rf /tmp/up
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/up -o s -- python3 bench.py --workload insert_stream $w --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  echo "== $w"; grep -h "k_stream_units\|k_stream_union\|k_bin_apply_stream" $(find /tmp/up -name "*kernel_stats.csv") | cut -c1-40,100-190
done
for rep in 1 2; do
python3 bench.py --workload insert_stream --stream-scans 32 --cpu-scans 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one room', round(d['value']), (d.get('parity') or {}).get('bit_exact'))"
python3 bench.py --workload insert_stream --stream-scans 64 --stream-tiles 64 --steps 3 --warmup 1 --prof-every 1 --cpu-scans 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('64 rooms', round(d['value']), (d.get('parity') or {}).get('bit_exact'))"
done
