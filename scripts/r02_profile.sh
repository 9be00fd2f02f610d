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
python3 scripts/pmc_sq.py $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) $O/r02_pmc_sq.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r02_bench_kernel_stats.csv
python3 scripts/trace_gaps.py $(find $O/stats -name "*kernel_trace.csv" | head -1) 20 > $O/r02_trace_gaps.txt
cat $O/r02_bench_kernel_stats.csv | cut -c1-200
# keep the merge small: drop the raw traces
find $O -name "*kernel_trace.csv" -size +2M -delete
# bench lines of the other workloads (plain runs, no profiler)
python3 bench.py > $O/r02_bench_default.json 2> $O/bench_default.err
python3 bench.py --steps 120 --no-cpu-baseline > $O/r02_bench_steps120.json 2>> $O/bench_default.err
python3 bench.py --workload match_batch --batch 64 --steps 10 > $O/r02_bench_match_batch64.json 2>> $O/bench_default.err
python3 bench.py --workload match_batch --batch 16 --steps 10 --no-cpu-baseline > $O/r02_bench_match_batch16.json 2>> $O/bench_default.err
python3 bench.py --workload register_batch --batch-submaps 8 --steps 40 --no-cpu-baseline > $O/r02_bench_register_batch8.json 2>> $O/bench_default.err
python3 bench.py --workload register_batch --batch-submaps 16 --steps 40 --no-cpu-baseline > $O/r02_bench_register_batch16.json 2>> $O/bench_default.err
python3 bench.py --workload insert_stream --stream-scans 32 > $O/r02_bench_insert_stream.json 2>> $O/bench_default.err
python3 bench.py --workload insert_stream --stream-scans 32 --insert-mode fast --no-cpu-baseline > $O/r02_bench_insert_stream_fast.json 2>> $O/bench_default.err
python3 bench.py --workload window --no-cpu-baseline > $O/r02_bench_window.json 2>> $O/bench_default.err
python3 bench.py --workload register_filtered --no-cpu-baseline > $O/r02_bench_register_filtered.json 2>> $O/bench_default.err
python3 bench.py --submaps 4 --steps 80 --no-cpu-baseline > $O/r02_bench_submaps4.json 2>> $O/bench_default.err
tail -n 3 $O/bench_default.err
cat $O/r02_bench_*.json | cut -c1-400
