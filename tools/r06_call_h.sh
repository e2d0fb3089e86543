#!/bin/bash
# round 6, GPU call H: persistent wide GEMM tiles -- tests, same-box A/B on the layer shapes and the headline
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gemm_gpu.py -q -x > gpurun_out/tests_h1.txt 2>&1; tail -5 gpurun_out/tests_h1.txt
python tools/gemm_layer0_gap.py 2>&1 | grep GEMM_DIAG | sed "s/^/PERSIST=1 /"
AVSI_GEMM_PERSIST=0 python tools/gemm_layer0_gap.py 2>&1 | grep GEMM_DIAG | sed "s/^/PERSIST=0 /"
python bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline 2> gpurun_out/bench_h.err | python tools/bench_line.py persist
AVSI_GEMM_PERSIST=0 python bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline 2> gpurun_out/bench_h0.err | python tools/bench_line.py legacy
