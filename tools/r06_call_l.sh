#!/bin/bash
# round 6, GPU call L: persistent wide GEMM with the tile queue -- tests, same-box A/B, HBM fetch of the K = 512 launch in both forms
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gemm_gpu.py -q -x > gpurun_out/tests_l1.txt 2>&1; tail -3 gpurun_out/tests_l1.txt
python tools/gemm_layer0_gap.py 2>&1 | grep GEMM_DIAG | sed "s/^/queue: /"
AVSI_GEMM_PERSIST=0 python tools/gemm_layer0_gap.py 2>&1 | grep GEMM_DIAG | sed "s/^/one workgroup per tile: /"
R=$PWD; cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/fq_$v
  AVSI_GEMM_PERSIST=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/fq_$v -o p -- python3 $R/tools/gemm_one.py 2048000 2048 512 > /tmp/fq_$v.txt 2> /tmp/fq_$v.err
  echo "PERSIST=$v FETCH_SIZE (KB, uncorrected: x2 for bytes/1024):"; python3 $R/tools/pmc_db.py $(ls /tmp/fq_$v/*/*.db /tmp/fq_$v/*.db 2>/dev/null | head -1) gemm_dma
done
cd $R
python bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline 2> gpurun_out/bench_l.err | python tools/bench_line.py queue
AVSI_GEMM_PERSIST=0 python bench.py --steps 10 --warmup 3 --no-also --no-cpu-baseline 2> gpurun_out/bench_l0.err | python tools/bench_line.py legacy
