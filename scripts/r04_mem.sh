#!/bin/bash
# Memory-pipeline counters of one workload (texture addresser / L1): bash scripts/r04_mem.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/r04m_$tag
mkdir -p $O
cd $R
i=0
# (a set the hardware cannot collect aborts rocprofv3, which then hangs in its signal handler: every pass under `timeout`)
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python3 bench.py "$@" --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > /dev/null 2> $O/p$i.err
  python3 scripts/pmc_sq.py $(find $O/p$i -name "*counter_collection.csv" | head -1) $O/mem$i.json "bench.py $*" 2>> $O/agg.err
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/mem*.json')):
    d=json.load(open(f))
    for k,v in d['kernels'].items():
        if 'residual' in k or 'bin_' in k:
            print(k[:60], {a:(round(b) if isinstance(b,float) else b) for a,b in v.items() if a not in ('per_wave',)})
PY
tail -2 $O/p1.err
