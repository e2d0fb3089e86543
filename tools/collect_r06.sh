#!/bin/bash
# Round-6 final profile collection (run on the GPU box from the repository root): bash tools/collect_r06.sh [commit]
#   kernel statistics of the bench workloads (tools/collect_profiles.sh), HBM traffic of the headline step, U-Net training
#   table, the default bench line.  The counters of the alternative forward kernels are tools/collect_r06_dual.sh.
set -e
commit=${1:-unknown}
R=$PWD
out=$R/gpurun_out/profiles
mkdir -p $out
bash tools/collect_profiles.sh r06 "b8192 train_b8192 b32 train_b32 b1024 unet_b512"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --batch 8192 --steps 1 --warmup 1 --no-cpu-baseline --no-also > /tmp/pmc_$c.json 2> /tmp/pmc_$c.err
done
python3 $R/tools/make_traffic_profile.py $(ls /tmp/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls /tmp/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) $out/r06_traffic_b8192.json $commit $(date +%F)
rm -rf /tmp/prof_ut
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ut -- python3 $R/tools/unet_train_one.py 512 > /tmp/prof_ut.txt 2> /tmp/prof_ut.err
cp $(ls /tmp/prof_ut/*/*kernel_stats.csv | head -1) $out/r06_bench_unet_train_b512_kernel_stats.csv
grep "B=" /tmp/prof_ut.txt > $out/r06_bench_unet_train_b512_under_rocprof.txt
rm -rf /tmp/prof_istft
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_istft -- python3 $R/tools/istft_time.py 4096 > /tmp/prof_istft.txt 2> /tmp/prof_istft.err
cp $(ls /tmp/prof_istft/*/*kernel_stats.csv | head -1) $out/r06_istft_b4096_kernel_stats.csv
grep "B=" /tmp/prof_istft.txt > $out/r06_istft_b4096_under_rocprof.txt
cd $R
python3 bench.py --steps 20 --warmup 5 > $out/r06_bench_default_line.json 2> /tmp/bench_default.err
tail -c 400 /tmp/bench_default.err
ls $out
