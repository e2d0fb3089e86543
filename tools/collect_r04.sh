#!/bin/bash
# Round-4 profile collection (run on the GPU box from the repository root): bash tools/collect_r04.sh [tag] [commit]
set -e
tag=${1:-r04}
commit=${2:-unknown}
R=$PWD
out=$R/gpurun_out/profiles
mkdir -p $out
bash tools/collect_profiles.sh $tag "b8192 train_b8192 b32 train_b32 b1024 unet_b512"
cd /tmp && export TMPDIR=/tmp
# HBM traffic of the headline step: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --batch 8192 --steps 1 --warmup 1 --no-cpu-baseline --no-also > /tmp/pmc_$c.json 2> /tmp/pmc_$c.err
done
python3 $R/tools/make_traffic_profile.py $(ls /tmp/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls /tmp/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) $out/${tag}_traffic_b8192.json $commit $(date +%F)
# enhanced_sources (tools/istft_time.py 4096: mode 3 from the waveform, mode 2 from a stored STFT): kernel statistics + traffic
rm -rf /tmp/prof_istft
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_istft -- python3 $R/tools/istft_time.py 4096 > /tmp/prof_istft.txt 2> /tmp/prof_istft.err
cp $(ls /tmp/prof_istft/*/*kernel_stats.csv | head -1) $out/${tag}_istft_b4096_kernel_stats.csv
grep "B=" /tmp/prof_istft.txt > $out/${tag}_istft_b4096_under_rocprof.txt
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_istft_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_istft_$c -- python3 $R/tools/istft_time.py 4096 > /tmp/pmc_istft_$c.txt 2> /tmp/pmc_istft_$c.err
done
python3 $R/tools/make_traffic_profile.py $(ls /tmp/pmc_istft_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls /tmp/pmc_istft_WRITE_SIZE/*/*counter_collection.csv | head -1) $out/${tag}_istft_traffic.json $commit $(date +%F) "istft_kernel<3, true>=1;istft_kernel<2, true>=1" "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/istft_time.py 4096 (two separate passes; last launch of each mode: 3 = phase from the waveform, 2 = from a stored STFT)" || true
# BPTT ping-pong kernel: matrix-pipe occupancy (tools/rec_bwd_time.py 8192)
rm -rf /tmp/pmc_bwd
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_bwd -o bwd -- python3 $R/tools/rec_bwd_time.py 8192 > /tmp/pmc_bwd.txt 2> /tmp/pmc_bwd.err || true
echo "== tools/rec_bwd_time.py 8192 (blstm_rec_bwd_pp_kernel), commit $commit" > $out/${tag}_bwd_pmc.txt
grep "Bp=" /tmp/pmc_bwd.txt >> $out/${tag}_bwd_pmc.txt || true
python3 $R/tools/pmc_db.py $(ls /tmp/pmc_bwd/*/*.db /tmp/pmc_bwd/*.db 2>/dev/null | head -1) blstm_rec_bwd >> $out/${tag}_bwd_pmc.txt || true
cat $out/${tag}_bwd_pmc.txt
ls $out
