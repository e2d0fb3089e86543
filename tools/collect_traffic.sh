#!/bin/bash
# HBM traffic of the headline step alone (the PMC part of tools/collect_r03.sh): bash tools/collect_traffic.sh [tag] [commit]
set -e
tag=${1:-r03}
commit=${2:-unknown}
R=$PWD
out=$R/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --batch 8192 --steps 1 --warmup 1 --no-cpu-baseline --no-also > /tmp/pmc_$c.json 2> /tmp/pmc_$c.err
done
python3 $R/tools/make_traffic_profile.py $(ls /tmp/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls /tmp/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) $out/${tag}_traffic_b8192.json $commit $(date +%F)
