#!/bin/bash
# SQ counters of one kernel family on the GPU box: pmc_kernel.sh <tag> <kernel substring>
out=gpurun_out/${1:-pmck}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out/p1 -- python3 bench.py --pmc-child > /dev/null 2> $out/err1.txt; echo "p1 $?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY --output-format csv -d $out/p2 -- python3 bench.py --pmc-child > /dev/null 2> $out/err2.txt; echo "p2 $?"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/p3 -- python3 bench.py --pmc-child > /dev/null 2> $out/err3.txt; echo "p3 $?"
python3 tools/pmc_summary.py $out "${2:-confirm}" | tee $out/summary.txt
