#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/gemm_layer0_gap.py 2>&1 | grep GEMM_DIAG
AVSI_GEMM_DIAG=2 python tools/gemm_layer0_gap.py 2>&1 | grep GEMM_DIAG
AVSI_GEMM_DIAG=1 python tools/gemm_layer0_gap.py 2>&1 | grep GEMM_DIAG
