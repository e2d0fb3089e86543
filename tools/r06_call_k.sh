#!/bin/bash
# round 6, GPU call K: 16-byte stores of the wide GEMM tile -- tests, same-box A/B (AVSI_GEMM_DIAG=4: dword stores), headline
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gemm_gpu.py tests/test_blstm_gpu.py -q -x > gpurun_out/tests_k1.txt 2>&1; tail -4 gpurun_out/tests_k1.txt
python tools/gemm_layer0_gap.py 2>&1 | grep GEMM_DIAG | sed "s/^/x4 stores: /"
AVSI_GEMM_DIAG=4 python tools/gemm_layer0_gap.py 2>&1 | grep GEMM_DIAG | sed "s/^/dword stores: /"
AVSI_GEMM_DIAG=1 python tools/gemm_layer0_gap.py 2>&1 | grep GEMM_DIAG | sed "s/^/no stores: /"
python bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline 2> gpurun_out/bench_k.err | python tools/bench_line.py x4
AVSI_GEMM_DIAG=4 python bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline 2> gpurun_out/bench_k4.err | python tools/bench_line.py dword
