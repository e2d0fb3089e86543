# SQ counters of the LWS sweep kernels: bash tools/lws_pmc.sh [B] ["kernels"]
B=${1:-512}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for k in ${2:-skew duo}; do
    export AVSI_LWS_KERNEL=$k
    i=0
    for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
               "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS" \
               "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
        i=$((i+1))
        rm -rf /tmp/pmc_lws_${k}_$i
        timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_lws_${k}_$i -- python3 $R/tools/lws_time.py $B > /tmp/pmc_lws_${k}_$i.txt 2> /tmp/pmc_lws_${k}_$i.err
        echo "== $k set $i (B=$B) rc=$?"; grep "B=" /tmp/pmc_lws_${k}_$i.txt
        f=$(ls /tmp/pmc_lws_${k}_$i/*/*counter_collection.csv 2>/dev/null | head -1)
        if [ -n "$f" ]; then python3 $R/tools/pmc_summary.py $f "lws_${k}_kernel"; else tail -3 /tmp/pmc_lws_${k}_$i.err; fi
    done
done
