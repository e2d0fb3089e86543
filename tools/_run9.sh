python -m pytest tests/test_bench_contract_gpu.py tests/test_ctc_gpu.py tests/test_dropout_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -5
