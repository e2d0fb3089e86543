python -m pytest tests/test_frontend_gpu.py tests/test_train_gpu.py tests/test_golden.py tests/test_blstm_gpu.py tests/test_bench_contract_gpu.py -x -q 2>&1 | tail -4
