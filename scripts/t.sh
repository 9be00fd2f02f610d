python -m pytest tests/test_gpu_match.py tests/test_gpu_headline.py tests/test_gpu_cpp_adapter.py tests/test_gpu_longrun.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2; do
HG_PREPARE_KERNEL=1 python bench.py --no-cpu-baseline 2>/dev/null | python scripts/ab_print.py "prepare kernel"
python bench.py --no-cpu-baseline 2>/dev/null | python scripts/ab_print.py "first-launch upload"
done
python bench.py --cpu-scans 3 | cut -c1-1500
