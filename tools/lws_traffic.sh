# HBM-side traffic of the LWS sweep kernels (FETCH_SIZE / WRITE_SIZE in separate passes): bash tools/lws_traffic.sh [B]
B=${1:-256}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for k in skew duo; do
    export AVSI_LWS_KERNEL=$k
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_lws_${k}_$c
        timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_lws_${k}_$c -- python3 $R/tools/lws_time.py $B > /tmp/pmc_lws_${k}_$c.txt 2> /tmp/pmc_lws_${k}_$c.err || exit 1
        echo "== $k $c (B=$B)"; grep "B=" /tmp/pmc_lws_${k}_$c.txt
        python3 $R/tools/pmc_summary.py $(ls /tmp/pmc_lws_${k}_$c/*/*counter_collection.csv | head -1) lws_
    done
done
