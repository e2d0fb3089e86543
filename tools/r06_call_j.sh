#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dp_gpu.py tests/test_bench_contract_gpu.py tests/test_train_gpu.py -q > gpurun_out/tests_j1.txt 2>&1; tail -5 gpurun_out/tests_j1.txt
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py > gpurun_out/bench_default_1.json 2> gpurun_out/bench_default_1.err; tail -c 600 gpurun_out/bench_default_1.err; python tools/bench_line.py default < gpurun_out/bench_default_1.json
