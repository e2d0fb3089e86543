#!/bin/bash
# Round-4 LWS profiles (run on the GPU box from the repository root): bash tools/collect_r04_lws.sh [commit]
commit=${1:-unknown}
R=$PWD
out=$R/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_lws
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lws -- python3 $R/tools/lws_time.py 1024 > /tmp/prof_lws.txt 2> /tmp/prof_lws.err
cp $(ls /tmp/prof_lws/*/*kernel_stats.csv | head -1) $out/r04_lws_b1024_kernel_stats.csv
{ echo "== tools/lws_time.py 1024 (kernel 'auto' = lws_duo_kernel), commit $commit, under rocprofv3 --kernel-trace --stats"; grep "B=" /tmp/prof_lws.txt; } > $out/r04_lws_pmc.txt
cd $R
bash tools/lws_pmc.sh 1024 "duo skew" >> $out/r04_lws_pmc.txt 2>&1
cat $out/r04_lws_pmc.txt
