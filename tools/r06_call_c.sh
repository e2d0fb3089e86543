#!/bin/bash
# round 6, GPU call C: the quarter-block forward kernel (parity vs the ping-pong kernel, timing), infer() with coalesced model steps
cd $GRAFT_REPO_ROOT
python tools/rec_q_check.py > gpurun_out/rec_q_check.txt 2>&1; cat gpurun_out/rec_q_check.txt | grep -v amdgpu.ids
python -m pytest tests/test_drivers_gpu.py -q -k "infer" > gpurun_out/tests_c1.txt 2>&1; tail -15 gpurun_out/tests_c1.txt
AVSI_E2E_PLAIN=1 python tools/e2e_infer_profile.py 4096 32 > gpurun_out/e2e_plain_b32_v3.txt 2>&1; grep -v WARNING gpurun_out/e2e_plain_b32_v3.txt | grep "infer"
