#!/bin/bash
# Round 6: tolerance insert mode on the bins. (The round-2 atomics form it was measured against -- HG_FAST_ATOMICS=1 --
# lived until commit 56d0d05 and was then removed: `old` below needs that commit's library.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06fast
mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_insert_fast.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1 2; do
python3 bench.py --workload insert_stream --stream-scans 32 --insert-mode fast --cpu-scans 2 > $O/fast_new_$i.json 2> $O/fast_new.err
done
if [ "$1" = "old" ]; then HG_FAST_ATOMICS=1 python3 bench.py --workload insert_stream --stream-scans 32 --insert-mode fast --no-cpu-baseline > $O/fast_old_1.json 2> $O/fast_old.err; fi
python3 bench.py --workload insert_stream --stream-scans 64 --stream-tiles 64 --insert-mode fast --steps 3 --warmup 1 --prof-every 1 --cpu-scans 2 > $O/fast_hbm64.json 2>> $O/fast_new.err
for f in $O/*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
d=json.load(open('$f'))
r=d['roofline']
print(round(d['value']), 'scans/s', (d.get('parity') or {}).get('tolerance_ok'), (d.get('parity') or {}).get('max_dtsd_over_tau'), {k:round(v,3) for k,v in r['per_kernel_ms_total'].items() if v}, r['per_kernel_launches']['apply'], round(r['frac'],4))
")"; done
tail -3 $O/fast_new.err
SQ="SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"
A="--workload insert_stream --stream-scans 32 --insert-mode fast --no-cpu-baseline --no-secondary"
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py $A --steps 8 --warmup 2 > $O/stats_line.json 2> $O/stats.err
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o p -- python3 bench.py $A --steps 2 --warmup 1 > /dev/null 2> $O/f.err
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o p -- python3 bench.py $A --steps 2 --warmup 1 > /dev/null 2> $O/w.err
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq -o p -- python3 bench.py $A --steps 2 --warmup 1 > /dev/null 2> $O/sq.err
python3 scripts/pmc_traffic.py $(find $O/f -name "*counter_collection.csv" | head -1) $(find $O/w -name "*counter_collection.csv" | head -1) $O/r06_insert_stream_fast_pmc_traffic.json "bench.py $A --steps 2 --warmup 1"
python3 scripts/pmc_sq.py $(find $O/sq -name "*counter_collection.csv" | head -1) $O/r06_insert_stream_fast_pmc_sq.json "bench.py $A --steps 2 --warmup 1"
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r06_insert_stream_fast_kernel_stats.csv
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
cut -c1-150 $O/r06_insert_stream_fast_kernel_stats.csv | head -8
cat $O/r06_insert_stream_fast_pmc_traffic.json | head -60
cat $O/r06_insert_stream_fast_pmc_sq.json | head -80
