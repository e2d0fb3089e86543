#!/bin/bash
# round 6, GPU call F: BPTT ping-pong kernel with the circular fragment ring (timing), recurrent / model parity, infer(), the bench line
cd $GRAFT_REPO_ROOT
python tools/rec_bwd_time.py 8192 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_recurrent_kernels_vs_oracle_gpu.py tests/test_blstm_gpu.py tests/test_train_gpu.py tests/test_fullsize_properties_gpu.py tests/test_drivers_gpu.py -q > gpurun_out/tests_f1.txt 2>&1; tail -6 gpurun_out/tests_f1.txt
AVSI_E2E_PLAIN=1 python tools/e2e_infer_profile.py 4096 32 > gpurun_out/e2e_plain_b32_v6.txt 2>&1; grep -v WARNING gpurun_out/e2e_plain_b32_v6.txt | grep "infer(" 
python bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline 2> gpurun_out/bench_f.err | python tools/bench_line.py headline
python bench.py --mode train --steps 4 --warmup 2 --no-also --no-cpu-baseline 2> gpurun_out/bench_f_train.err | tail -c 1500
