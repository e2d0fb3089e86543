#!/bin/bash
# round 6, GPU call G: same-box A/B of the resident first fragment group (forward) and of the circular ring (BPTT); layer-0 GEMM gap
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in 1 0; do
    AVSI_PP_RESIDENT0=$v python tools/rec_fwd_time.py 8192 250 64 2>&1 | grep "rows_per_wg" | sed "s/^/RESIDENT0=$v /"
  done
done
for rep in 1 2; do
  for v in 1 0; do
    AVSI_BWD_PP_CIRC=$v python tools/rec_bwd_time.py 8192 2>&1 | grep "TFLOP" | sed "s/^/CIRC=$v /"
  done
done
python tools/gemm_layer0_gap.py 2>&1 | grep GEMM_DIAG
AVSI_GEMM_DIAG=1 python tools/gemm_layer0_gap.py 2>&1 | grep GEMM_DIAG
