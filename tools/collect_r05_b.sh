#!/bin/bash
# Round-5 profile collection, part B: the memory-path counters of the two recurrent kernels of training at 8192 utterances
# (DESIGN 4.3's open question: which queue do the cell's input bytes fill beside the Wh^T stream?).  One block per pass.
#   bash tools/collect_r05_b.sh [commit]
commit=${1:-unknown}
R=$PWD
out=$R/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
passes=(
 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM"
 "TA_TA_BUSY_sum TA_BUSY_avr"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum"
 "TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"
 "TCP_GATE_EN1_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum"
 "TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
)
for tool in rec_bwd_time rec_fwd_time; do
    f=$out/r05_${tool%_time}_mem_pmc.txt
    echo "== rocprofv3 --pmc <one line per pass> --kernel-trace -- python3 tools/$tool.py 8192, commit $commit; values are sums over the counter's instances, median over the dispatches" > $f
    i=0
    for p in "${passes[@]}"; do
        i=$((i+1))
        rm -rf /tmp/pmcb_$i
        if rocprofv3 --pmc $p --kernel-trace -d /tmp/pmcb_$i -o p -- python3 $R/tools/$tool.py 8192 > /tmp/pmcb_$i.txt 2> /tmp/pmcb_$i.err; then
            echo "-- pass $i: $p" >> $f
            python3 $R/tools/pmc_db.py $(ls /tmp/pmcb_$i/*/*.db /tmp/pmcb_$i/*.db 2>/dev/null | head -1) blstm_rec >> $f 2>&1
        else
            echo "-- pass $i FAILED: $p :: $(tail -2 /tmp/pmcb_$i.err | tr '\n' ' ')" >> $f
        fi
        echo "pass $i of $tool done"
    done
done
cat $out/r05_rec_bwd_mem_pmc.txt $out/r05_rec_fwd_mem_pmc.txt
