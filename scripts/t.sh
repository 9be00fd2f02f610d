python -m pytest tests -q -m gpu -x 2>&1 | tail -3
./hectorgrapher_amd/cpp/example_threads 60
python scripts/diag_threads.py 2>&1 | tail -6
python bench.py --no-cpu-baseline | python scripts/ab_print.py "single"
python bench.py --no-cpu-baseline --workload insert_stream --stream-scans 32 --prof-every 0 | python scripts/ab_print.py "stream"
python bench.py --no-cpu-baseline --submaps 4 --steps 80 | python scripts/ab_print.py "submaps4"
