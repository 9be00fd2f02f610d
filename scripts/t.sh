python -m pytest tests/test_gpu_match.py tests/test_gpu_headline.py tests/test_golden.py tests/test_gpu_longrun.py tests/test_gpu_cpp_adapter.py -q -m gpu -x 2>&1 | tail -3
bash scripts/ab.sh "--steps 20" base lds base lds
bash scripts/ab.sh "--workload match_batch --batch 64 --steps 10" base lds
