#!/bin/bash
# Re-collect the rocprofv3 kernel-trace summaries kept under profiles/ (run on the GPU box from the repository root):
#   bash tools/collect_profiles.sh r02 "b32 train_b32 b1024 unet_b512"
# Each entry: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py <flags>; the JSON line the bench printed
# under the profiler goes next to the kernel statistics.
set -e
tag=${1:-r02}
what=${2:-"b8192 train_b8192 b32 train_b32 b1024 unet_b512"}
R=$PWD
out=$R/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for w in $what; do
    case $w in
        b8192) flags="--batch 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-also" ;;
        train_b8192) flags="--mode train --batch 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-also" ;;
        b32) flags="--batch 32 --steps 20 --warmup 3 --no-cpu-baseline --no-also" ;;
        train_b32) flags="--mode train --batch 32 --steps 20 --warmup 3 --no-cpu-baseline --no-also" ;;
        b1024) flags="--batch 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-also" ;;
        unet_b512) flags="--mode unet --batch 512 --steps 5 --warmup 2 --no-cpu-baseline --no-also" ;;
        lws_b*)      # LWS phase reconstruction alone (tools/lws_time.py <utterances>): kernel statistics + the tool's line
            n=${w#lws_b}
            rm -rf /tmp/prof_$w
            rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -- python3 $R/tools/lws_time.py $n > /tmp/prof_$w.txt 2> /tmp/prof_$w.err
            cp $(ls /tmp/prof_$w/*/*kernel_stats.csv | head -1) $out/${tag}_${w}_kernel_stats.csv
            grep "B=" /tmp/prof_$w.txt > $out/${tag}_${w}_under_rocprof.txt
            echo "$w: $(cat $out/${tag}_${w}_under_rocprof.txt)"
            continue ;;
        *) echo "unknown $w"; exit 1 ;;
    esac
    rm -rf /tmp/prof_$w
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -- python3 $R/bench.py $flags > /tmp/prof_$w.json 2> /tmp/prof_$w.err
    f=$(ls /tmp/prof_$w/*/*kernel_stats.csv | head -1)
    cp $f $out/${tag}_bench_${w}_kernel_stats.csv
    tail -1 /tmp/prof_$w.json > $out/${tag}_bench_${w}_under_rocprof.json
    echo "$w: $(python3 -c "import json,sys; d=json.load(open('$out/${tag}_bench_${w}_under_rocprof.json')); print(round(d['value']), d['unit'], round(d['ms_per_step'],2), 'ms')")"
done
