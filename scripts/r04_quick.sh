#!/bin/bash
# Quick look at one workload: bench line + rocprofv3 kernel stats + SQ counters. Usage: bash scripts/r04_quick.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/r04q_$tag
mkdir -p $O
cd $R
SQ="SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"
python3 bench.py "$@" --no-secondary > $O/line.json 2> $O/line.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py "$@" --no-cpu-baseline --no-secondary > /dev/null 2> $O/stats.err
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq -o p -- python3 bench.py "$@" --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > /dev/null 2> $O/sq.err
python3 scripts/pmc_sq.py $(find $O/sq -name "*counter_collection.csv" | head -1) $O/pmc_sq.json "bench.py $*"
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
cut -c1-700 $O/line.json; echo; head -8 $O/kernel_stats.csv | cut -c1-200; python3 -c "
import json; d=json.load(open('$O/pmc_sq.json'))
for k,v in d['kernels'].items():
    if 'residual' in k or 'lm' in k: print(k[:70], {a: (round(b,1) if isinstance(b,float) else b) for a,b in v.items()})
"
