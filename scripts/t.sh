python -m pytest tests/test_gpu_insert.py tests/test_gpu_headline.py tests/test_gpu_match.py tests/test_gpu_longrun.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | python scripts/ab_print.py single
python bench.py --no-cpu-baseline --steps 120 2>/dev/null | python scripts/ab_print.py steps120
python bench.py --workload insert_stream --stream-scans 32 --no-cpu-baseline --prof-every 0 2>/dev/null | python scripts/ab_print.py stream
done
python bench.py --no-cpu-baseline --workload register_batch --batch-submaps 8 --steps 40 2>/dev/null | python scripts/ab_print.py batch8
python bench.py --no-cpu-baseline --workload register_batch --batch-submaps 2 --steps 40 2>/dev/null | python scripts/ab_print.py batch2
python bench.py --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['roofline']['per_kernel_ms_total'])"
