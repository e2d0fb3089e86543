#!/bin/bash
# round 6, GPU call D: counters of the quarter-block kernel, driver tests, infer() host-time laps
cd $GRAFT_REPO_ROOT
RPWS=66 OUTNAME=r06_rec_fwd_q_pmc.txt bash tools/collect_r06_dual.sh $1 > gpurun_out/collect_r06_q.log 2>&1; grep -A5 "pass 1\|pass 2" gpurun_out/profiles/r06_rec_fwd_q_pmc.txt | grep -v "^--$" | head -40
python -m pytest tests/test_drivers_gpu.py -q > gpurun_out/tests_d1.txt 2>&1; tail -8 gpurun_out/tests_d1.txt
AVSI_E2E_PLAIN=1 python tools/e2e_infer_profile.py 4096 32 > gpurun_out/e2e_plain_b32_v4.txt 2>&1; grep -v WARNING gpurun_out/e2e_plain_b32_v4.txt | grep "infer"
