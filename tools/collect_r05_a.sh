#!/bin/bash
# Round-5 profile collection, part A (run on the GPU box from the repository root): bash tools/collect_r05_a.sh [commit]
#   - the counter list of this box (rocprofv3 -L), for the TA / TCP / TCC passes of part B
#   - U-Net training at 512 clips: kernel statistics (tools/unet_train_one.py 512)
#   - SQ counters of the dominant GEMM and of the ping-pong forward recurrence
commit=${1:-unknown}
R=$PWD
out=$R/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1 || true
rm -rf /tmp/prof_ut
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ut -- python3 $R/tools/unet_train_one.py 512 > /tmp/prof_ut.txt 2> /tmp/prof_ut.err
cp $(ls /tmp/prof_ut/*/*kernel_stats.csv | head -1) $out/r05_bench_unet_train_b512_kernel_stats.csv
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
rm -rf /tmp/pmc_gemm
rocprofv3 --pmc $SQ --kernel-trace -d /tmp/pmc_gemm -o gemm -- python3 $R/tools/gemm_one.py 2048000 2048 512 > /tmp/pmc_gemm.txt 2> /tmp/pmc_gemm.err
echo "== rocprofv3 --pmc $SQ --kernel-trace -- python3 tools/gemm_one.py 2048000 2048 512  (layer input projection of layers 1-2 at 8192 utterances), commit $commit" > $out/r05_gemm_pmc.txt
python3 $R/tools/pmc_db.py $(ls /tmp/pmc_gemm/*/*.db /tmp/pmc_gemm/*.db 2>/dev/null | head -1) gemm_dma >> $out/r05_gemm_pmc.txt
rm -rf /tmp/pmc_fwd
rocprofv3 --pmc $SQ --kernel-trace -d /tmp/pmc_fwd -o fwd -- python3 $R/tools/rec_fwd_time.py 8192 > /tmp/pmc_fwd.txt 2> /tmp/pmc_fwd.err
echo "== rocprofv3 --pmc $SQ --kernel-trace -- python3 tools/rec_fwd_time.py 8192  (blstm_rec_fwd_pp_kernel<false> then <true>), commit $commit" > $out/r05_fwd_pmc.txt
grep "Bp=" /tmp/pmc_fwd.txt >> $out/r05_fwd_pmc.txt
python3 $R/tools/pmc_db.py $(ls /tmp/pmc_fwd/*/*.db /tmp/pmc_fwd/*.db 2>/dev/null | head -1) blstm_rec_fwd >> $out/r05_fwd_pmc.txt
cat $out/r05_gemm_pmc.txt $out/r05_fwd_pmc.txt
tail -3 /tmp/pmc_gemm.err /tmp/pmc_fwd.err /tmp/prof_ut.err
