#!/bin/bash
# round 6, GPU call B: GEMM fold tests, host-side tests, headline A/B of the folded projection, infer() stamps at batch 32
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gemm_gpu.py tests/test_train_gpu.py tests/test_step_guard_gpu.py -q > gpurun_out/tests_b1.txt 2>&1; tail -4 gpurun_out/tests_b1.txt
python bench.py --steps 5 --warmup 2 --no-also --no-cpu-baseline 2> gpurun_out/bench_fold.err | python tools/bench_line.py fold
AVSI_GEMM_FOLD_TAIL=0 python bench.py --steps 5 --warmup 2 --no-also --no-cpu-baseline 2> gpurun_out/bench_nofold.err | python tools/bench_line.py nofold
AVSI_E2E_PLAIN=1 python tools/e2e_infer_profile.py 4096 32 > gpurun_out/e2e_plain_b32_v2.txt 2>&1; grep -v WARNING gpurun_out/e2e_plain_b32_v2.txt | grep "infer" 
python -m pytest tests/test_dp_gpu.py tests/test_drivers_gpu.py -q > gpurun_out/tests_b2.txt 2>&1; tail -4 gpurun_out/tests_b2.txt
