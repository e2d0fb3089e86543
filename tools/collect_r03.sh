#!/bin/bash
# Round-3 profile collection (run on the GPU box from the repository root): bash tools/collect_r03.sh [tag] [commit]
set -e
tag=${1:-r03}
commit=${2:-unknown}
R=$PWD
out=$R/gpurun_out/profiles
mkdir -p $out
bash tools/collect_profiles.sh $tag "b8192 train_b8192 b32 train_b32 b1024 unet_b512 lws_b32 lws_b1024"
cd /tmp && export TMPDIR=/tmp
# HBM traffic of the headline step: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --batch 8192 --steps 1 --warmup 1 --no-cpu-baseline --no-also > /tmp/pmc_$c.json 2> /tmp/pmc_$c.err
done
python3 $R/tools/make_traffic_profile.py $(ls /tmp/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls /tmp/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) $out/${tag}_traffic_b8192.json $commit $(date +%F)
# LWS sweeps: lanes that do work per VALU instruction, both kernels (the raster kernel runs its recurrence on 1 - 4 lanes)
for k in skew raster; do
    rm -rf /tmp/pmc_lws_$k
    AVSI_LWS_KERNEL=$k rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_lws_$k -o lws -- python3 $R/tools/lws_time.py 256 > /tmp/pmc_lws_$k.txt 2> /tmp/pmc_lws_$k.err || true
    echo "== AVSI_LWS_KERNEL=$k, tools/lws_time.py 256" >> $out/${tag}_lws_pmc.txt
    grep "B=" /tmp/pmc_lws_$k.txt >> $out/${tag}_lws_pmc.txt || true
    python3 $R/tools/pmc_db.py $(ls /tmp/pmc_lws_$k/*/*.db /tmp/pmc_lws_$k/*.db 2>/dev/null | head -1) lws_s >> $out/${tag}_lws_pmc.txt || true
done
cat $out/${tag}_lws_pmc.txt
