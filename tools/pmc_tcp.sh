#!/bin/bash
# Vector-memory-pipe counters (TA / TCP / TD) of the kernels of two 1-slice frames: pmc_tcp.sh <tag> [kernel substring]
out=gpurun_out/${1:-pmctcp}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 rocprofv3 --pmc "${@:2}" --output-format csv -d $out/$1 -- python3 bench.py --pmc-child > /dev/null 2> $out/err_$1.txt; echo "$1 $?"; }
run t1 TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum GRBM_GUI_ACTIVE
run t2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum
run t3 TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum
run t4 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run t5 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
run t6 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum SQ_INSTS_VMEM_RD
run t7 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM
python3 tools/pmc_summary.py $out "${2:-_kernel}" | tee $out/summary.txt
