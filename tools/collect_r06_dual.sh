#!/bin/bash
# Round 6, VERDICT r5 item 2 (stop rule): the forward recurrence with BOTH 32-row tiles of a workgroup in ONE MFMA phase on
# the same Wh fragments (blstm_rec_fwd_kernel<2>, rows_per_wg = 65: Wh streamed ONCE per step and workgroup) beside the
# ping-pong kernel (rows_per_wg = 64: twice), 8192 utterances, one layer: launch time, clock, matrix-pipe busy cycles and the
# L1 / L2 counters DESIGN 4.3 tabulates.  One counter block per pass.
#   bash tools/collect_r06_dual.sh [commit]
commit=${1:-unknown}
R=$PWD
out=$R/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
passes=(
 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"
 "TA_TA_BUSY_sum TA_BUSY_avr"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum"
)
f=$out/${OUTNAME:-r06_rec_fwd_dual_pmc.txt}
echo "== rocprofv3 --pmc <one line per pass> --kernel-trace -- python3 tools/rec_fwd_time.py 8192 250 <rows_per_wg>, commit $commit; values are sums over the counter's instances, median over the dispatches" > $f
for rpw in ${RPWS:-64 65}; do
    echo "==== rows_per_wg = $rpw ($([ $rpw = 64 ] && echo 'blstm_rec_fwd_pp_kernel: two tiles ping-pong, Wh streamed twice per step' || ([ $rpw = 66 ] && echo 'blstm_rec_fwd_q_kernel: quarter products, paired waves on the same fragments' || echo 'blstm_rec_fwd_kernel<2>: both tiles in one MFMA phase, Wh streamed once per step')))" >> $f
    python3 $R/tools/rec_fwd_time.py 8192 250 $rpw >> $f 2>&1
    i=0
    for p in "${passes[@]}"; do
        i=$((i+1))
        rm -rf /tmp/pmcd_$i
        if rocprofv3 --pmc $p --kernel-trace -d /tmp/pmcd_$i -o p -- python3 $R/tools/rec_fwd_time.py 8192 250 $rpw > /tmp/pmcd_$i.txt 2> /tmp/pmcd_$i.err; then
            echo "-- pass $i: $p" >> $f
            python3 $R/tools/pmc_db.py $(ls /tmp/pmcd_$i/*/*.db /tmp/pmcd_$i/*.db 2>/dev/null | head -1) blstm_rec >> $f 2>&1
        else
            echo "-- pass $i FAILED: $p :: $(tail -2 /tmp/pmcd_$i.err | tr '\n' ' ')" >> $f
        fi
        echo "rows_per_wg $rpw pass $i done"
    done
done
cat $f
