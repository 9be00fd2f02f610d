#!/bin/bash
# rocprofv3 passes of the default bench command (kernel stats; FETCH_SIZE; WRITE_SIZE; SQ counters), each its own run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02prof
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_stats.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_w.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_sq -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_sq.err
find $O -name "*.csv" | head -20
python3 scripts/pmc_traffic.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/r02_pmc_traffic.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r02_bench_kernel_stats.csv
cat $O/r02_bench_kernel_stats.csv | cut -c1-200
# keep the merge small: drop the raw traces
find $O -name "*kernel_trace.csv" -size +2M -delete
