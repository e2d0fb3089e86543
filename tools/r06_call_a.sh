#!/bin/bash
# round 6, GPU call A: CU-mask probe, infer() stage stamps at batch 32, the tests of this round's host-side changes, dual-tile counters
cd $GRAFT_REPO_ROOT
timeout -k 5 60 tools/cu_mask_probe > gpurun_out/cu_mask_probe.txt 2>&1; echo probe rc $?; cat gpurun_out/cu_mask_probe.txt
AVSI_E2E_PLAIN=1 python tools/e2e_infer_profile.py 4096 32 > gpurun_out/e2e_plain_b32.txt 2>&1; grep -v WARNING gpurun_out/e2e_plain_b32.txt | tail -20
python -m pytest tests/test_train_gpu.py tests/test_step_guard_gpu.py tests/test_dp_gpu.py tests/test_drivers_gpu.py -x -q > gpurun_out/tests_a.txt 2>&1; tail -5 gpurun_out/tests_a.txt
bash tools/collect_r06_dual.sh $1 > gpurun_out/collect_r06_dual.log 2>&1; tail -3 gpurun_out/collect_r06_dual.log
