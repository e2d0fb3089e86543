#!/bin/bash
# round 6, GPU call E: ping-pong kernel with its first fragment group resident, driver tests, infer() with deferred delivery
cd $GRAFT_REPO_ROOT
python tools/rec_q_check.py 2>&1 | grep "rows_per_wg=64"
python -m pytest tests/test_drivers_gpu.py -q > gpurun_out/tests_e1.txt 2>&1; tail -8 gpurun_out/tests_e1.txt
AVSI_E2E_PLAIN=1 python tools/e2e_infer_profile.py 4096 32 > gpurun_out/e2e_plain_b32_v5.txt 2>&1; grep -v WARNING gpurun_out/e2e_plain_b32_v5.txt | grep "infer"
AVSI_E2E_PLAIN=1 python tools/e2e_infer_profile.py 16384 1024 > gpurun_out/e2e_plain_b1024_v5.txt 2>&1; grep -v WARNING gpurun_out/e2e_plain_b1024_v5.txt | grep "infer"
