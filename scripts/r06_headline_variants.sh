#!/bin/bash
# Round 6: the persistent solve's pieces one at a time (diagnostic builds: scripts/build_variant.sh), interleaved.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06var; mkdir -p $O
A="--steps 60 --warmup 3 --no-secondary --host-steps 0 --no-cpu-baseline"
for rep in 1 2 3; do
for v in main nohoist late libsincos r6a; do
  lib=$GRAFT_REPO_ROOT/scripts/libhg_$v.so; [ $v = main ] && lib=$GRAFT_REPO_ROOT/hectorgrapher_amd/libhg_mi355x.so
  HG_LIB_PATH=$lib timeout 300 python3 bench.py $A > $O/${v}_on_$rep.json 2>/dev/null
  HG_LIB_PATH=$lib timeout 300 python3 bench.py $A --no-persistent-solve > $O/${v}_off_$rep.json 2>/dev/null
done; done
python3 - <<'PY'
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r06var/*.json')):
    try: d=json.load(open(f))
    except Exception: continue
    k='_'.join(f.split('/')[-1].split('_')[:2])
    r[k].append(d['value'])
for k,v in sorted(r.items()): print(k, [round(x) for x in v], round(sum(v)/len(v)))
PY
